// graph_latency.hip -- does a captured hipGraph beat plain launches for a short
// chain of small dependent kernels (one tree level each)?  Measures wall time of
//   (a) N launches + stream sync            (b) one hipGraphLaunch of the same N nodes + sync
// hipcc --offload-arch=gfx950 -O2 tools/graph_latency.hip -o tools/graph_latency.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Args { double * p[24]; unsigned int n; };

__global__ void k_level(Args a)
{
  double * p = a.p[blockIdx.y];
  for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += gridDim.x * blockDim.x)
    p[i] = p[i] * 1.0000001 + 1e-9;
}

int main(int argc, char ** argv)
{
  const unsigned int n = argc > 1 ? atoi(argv[1]) : 160000; // doubles per buffer (10 k sites x 16)
  const int levels = argc > 2 ? atoi(argv[2]) : 6;
  const int reps = 2000;
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  Args a;
  a.n = n;
  for (int i = 0; i < 24; ++i) { CK(hipMalloc((void **)&a.p[i], n * sizeof(double))); CK(hipMemset(a.p[i], 0, n * sizeof(double))); }
  const dim3 grid((n + 255) / 256, 8), block(256);

  auto run_plain = [&]() { for (int l = 0; l < levels; ++l) hipLaunchKernelGGL(k_level, grid, block, 0, s, a); };
  for (int i = 0; i < 50; ++i) run_plain();
  CK(hipStreamSynchronize(s));
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r) { run_plain(); CK(hipStreamSynchronize(s)); }
  double plain = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;

  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  run_plain();
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 50; ++i) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r) { CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
  double graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;

  // one launch alone, for the floor
  t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r) { hipLaunchKernelGGL(k_level, grid, block, 0, s, a); CK(hipStreamSynchronize(s)); }
  double one = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
  printf("n=%u levels=%d : plain %.1f us, graph %.1f us, single launch+sync %.1f us\n", n, levels, plain, graph, one);
  return 0;
}
