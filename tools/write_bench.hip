// write_bench.hip -- how fast can a write-only stream go?  (ceiling for the tip-tip kernels)
// hipcc --offload-arch=gfx950 -O3 tools/write_bench.hip -o tools/write_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

template <bool NT, int ROUNDS>
__global__ __launch_bounds__(256) void k_write(v2d * out, size_t n, double a)
{
  // every wave writes ROUNDS x 8 consecutive KiB, like a round of the tip-tip kernel
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const unsigned int lane = threadIdx.x & 63u;
  for (int r = 0; r < ROUNDS; ++r)
  {
    const size_t base = (wave * ROUNDS + r) * 512;
#pragma unroll
    for (int j = 0; j < 8; ++j)
    {
      const size_t g = base + j * 64 + lane;
      const v2d v = {a + (double)j, a * (double)lane};
      if (g < n)
      {
        if (NT) __builtin_nontemporal_store(v, out + g);
        else out[g] = v;
      }
    }
  }
}

template <bool NT, int ROUNDS>
static void run(v2d * d, size_t n, const char * name)
{
  const size_t waves = (n + 512 * ROUNDS - 1) / (512 * ROUNDS);
  const unsigned int grid = (unsigned int)((waves + 3) / 4);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) k_write<NT, ROUNDS><<<grid, 256>>>(d, n, 1.5);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 20; ++i) k_write<NT, ROUNDS><<<grid, 256>>>(d, n, 1.5);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-28s %7.2f us  %6.2f TB/s\n", name, ms / 20 * 1e3, n * 16.0 / (ms / 20 * 1e-3) / 1e12);
}

int main()
{
  const size_t n = (size_t)1000000 * 8; // 16-byte granules of a 1 M-site 4x4 CLV (128 MB)
  v2d * d;
  CK(hipMalloc((void **)&d, n * 16 * 8));
  run<false, 1>(d, n, "plain stores, 1 round/wave");
  run<true, 1>(d, n, "nontemporal, 1 round/wave");
  run<false, 4>(d, n, "plain stores, 4 rounds/wave");
  run<true, 4>(d, n, "nontemporal, 4 rounds/wave");
  run<false, 1>(d, n * 8, "plain, 1 GB, 1 round/wave");
  run<true, 1>(d, n * 8, "nontemporal, 1 GB");
  CK(hipMemset(d, 0, n * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < 20; ++i) CK(hipMemsetAsync(d, 0, n * 16, 0));
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-28s %7.2f us  %6.2f TB/s\n", "hipMemsetAsync 128 MB", ms / 20 * 1e3, n * 16.0 / (ms / 20 * 1e-3) / 1e12);
  return 0;
}
