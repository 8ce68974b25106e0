// lookahead_bench.hip -- how much does the distance between a load's issue and its use matter for a kernel that
// also streams stores?  Vector-memory operations retire in ONE in-order queue per wave (vmcnt), so consuming a load
// forces every store issued before it to be acknowledged.  The whole-list kernel's pattern: per op two 1 KB stores to
// stream k, and a small L2-resident load (its P-matrix block / pair-table entry) that was requested D ops earlier.
// D = 1, 2, 3, 4; 12 waves per CU; 62 and 126 streams of 1 M sites.
//   hipcc --offload-arch=gfx950 -O3 tools/lookahead_bench.hip -o tools/lookahead_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

template <int D>
__global__ __launch_bounds__(256, 3) void k_pattern(v2d * base, size_t stride_g, unsigned int K, size_t tiles, const v2d * table)
{
  const unsigned int lane = threadIdx.x & 63u;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
  for (size_t t = wave; t < tiles; t += nwaves)
  {
    v2d acc = {(double)t, (double)lane};
    v2d q[D];
#pragma unroll
    for (int d = 0; d < D; ++d) q[d] = table[(((t + d) * 37 + lane) & 4095u) * 8 + (lane & 7u)];
    for (unsigned int k = 0; k < K; ++k)
    {
      // request the entry of op k + D, use the one requested D ops ago
      const v2d fresh = table[(((t + k + D) * 37 + lane) & 4095u) * 8 + (lane & 7u)];
      acc = acc * 1.0000001 + q[0];
#pragma unroll
      for (int d = 0; d + 1 < D; ++d) q[d] = q[d + 1];
      q[D - 1] = fresh;
      v2d * out = base + (size_t)k * stride_g + t * 128;
      __builtin_nontemporal_store(acc, out + lane);
      __builtin_nontemporal_store(acc, out + 64 + lane);
    }
  }
}

template <int D>
static void run(v2d * d, size_t stride_g, unsigned int K, size_t sites, const v2d * table)
{
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 6; ++rep)
  {
    CK(hipEventRecord(e0));
    k_pattern<D><<<768, 256>>>(d, stride_g, K, sites / 16, table);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep >= 2 && ms < best) best = ms;
  }
  printf("K %3u, load used %d op(s) after its request: %8.1f us  %5.2f TB/s\n", K, D, best * 1e3,
         (double)K * (sites / 16) * 2048 / (best * 1e-3) / 1e12);
}

int main()
{
  v2d * table; CK(hipMalloc((void **)&table, 4096 * 8 * 16 + 4096)); CK(hipMemset(table, 0, 4096 * 8 * 16 + 4096));
  for (unsigned int K : {62u, 126u})
  {
    const size_t sites = 1000000, stride_g = (sites + 64) * 8, total = stride_g * K * 16;
    v2d * d; CK(hipMalloc((void **)&d, total)); CK(hipMemset(d, 0, total));
    run<1>(d, stride_g, K, sites, table);
    run<2>(d, stride_g, K, sites, table);
    run<3>(d, stride_g, K, sites, table);
    run<4>(d, stride_g, K, sites, table);
    run<6>(d, stride_g, K, sites, table);
    CK(hipFree(d));
  }
  return 0;
}
