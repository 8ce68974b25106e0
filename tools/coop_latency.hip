// coop_latency.hip -- is one cooperative launch with grid-wide barriers between tree
// levels cheaper than one plain launch per level on a small partition?
//   (a) L plain launches + stream sync     (b) one hipLaunchCooperativeKernel with L-1 grid.sync()
// hipcc --offload-arch=gfx950 -O2 tools/coop_latency.hip -o tools/coop_latency.bin
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace cg = cooperative_groups;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Args { double * p[24]; unsigned int n; int levels; };

__device__ void level_body(const Args & a, unsigned int nblocks)
{
  for (unsigned int op = 0; op < 8; ++op)
  {
    double * p = a.p[op];
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += nblocks * blockDim.x)
      p[i] = p[i] * 1.0000001 + 1e-9;
  }
}

__global__ void k_level(Args a) { level_body(a, gridDim.x); }

__global__ void k_all(Args a)
{
  cg::grid_group g = cg::this_grid();
  for (int l = 0; l < a.levels; ++l)
  {
    level_body(a, gridDim.x);
    if (l + 1 < a.levels) g.sync();
  }
}

int main(int argc, char ** argv)
{
  const unsigned int n = argc > 1 ? atoi(argv[1]) : 16000;
  const int levels = argc > 2 ? atoi(argv[2]) : 6;
  const int reps = 2000;
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  Args a;
  a.n = n;
  a.levels = levels;
  for (int i = 0; i < 24; ++i) { CK(hipMalloc((void **)&a.p[i], n * sizeof(double))); CK(hipMemset(a.p[i], 0, n * sizeof(double))); }
  int dev = 0, cus = 0, per_cu = 0;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_all, 256, 0));
  unsigned int blocks = (n + 255) / 256;
  if (blocks > (unsigned int)(cus * per_cu)) blocks = cus * per_cu;
  const dim3 grid(blocks), block(256);

  for (int i = 0; i < 50; ++i) for (int l = 0; l < levels; ++l) hipLaunchKernelGGL(k_level, grid, block, 0, s, a);
  CK(hipStreamSynchronize(s));
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r) { for (int l = 0; l < levels; ++l) hipLaunchKernelGGL(k_level, grid, block, 0, s, a); CK(hipStreamSynchronize(s)); }
  const double plain = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;

  void * params[] = {&a};
  for (int i = 0; i < 50; ++i) CK(hipLaunchCooperativeKernel((const void *)k_all, grid, block, params, 0, s));
  CK(hipStreamSynchronize(s));
  t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r) { CK(hipLaunchCooperativeKernel((const void *)k_all, grid, block, params, 0, s)); CK(hipStreamSynchronize(s)); }
  const double coop = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
  printf("n=%u levels=%d blocks=%u (max %d/CU): plain %.1f us, cooperative %.1f us\n", n, levels, blocks, per_cu, plain, coop);
  return 0;
}
