// microbench.hip -- developer tool: what does a 2-reads + 1-write 16-B-per-lane
// stream reach on this GPU, as a function of grid size, unroll, traversal order
// and cache policy?  Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o /tmp/mb
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ntload(const double2 * p) { v2d v = __builtin_nontemporal_load((const v2d *)p); return make_double2(v.x, v.y); }
__device__ __forceinline__ void ntstore(double2 r, double2 * p) { v2d v = {r.x, r.y}; __builtin_nontemporal_store(v, (v2d *)p); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int U, bool NT, bool CHUNK, int NREAD>
__global__ __launch_bounds__(256) void k_rw(const double2 * __restrict__ a, const double2 * __restrict__ b,
                                            double2 * __restrict__ out, size_t total)
{
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  size_t g, step, end;
  if (CHUNK)
  {
    // each block owns one contiguous chunk
    const size_t per = (total + gridDim.x - 1) / gridDim.x;
    g = blockIdx.x * per + threadIdx.x; step = blockDim.x; end = min(total, (blockIdx.x + 1) * per);
  }
  else { g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; step = nthreads; end = total; }
  for (; g + (U - 1) * step < end; g += U * step)
  {
    double2 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      if (NREAD >= 1) x[u] = NT ? ntload(a + g + u * step) : a[g + u * step]; else x[u] = make_double2(1.0, 2.0);
      if (NREAD >= 2) y[u] = NT ? ntload(b + g + u * step) : b[g + u * step]; else y[u] = make_double2(3.0, 4.0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      double2 r = make_double2(x[u].x * y[u].x + x[u].y, x[u].y * y[u].y + y[u].x);
      if (NT) ntstore(r, out + g + u * step); else out[g + u * step] = r;
    }
  }
  for (; g < end; g += step) out[g] = make_double2(a[g].x + b[g].x, a[g].y);
}

template <int NREAD>
__global__ __launch_bounds__(256) void k_readonly(const double2 * __restrict__ a, const double2 * __restrict__ b,
                                                  double * __restrict__ out, size_t total)
{
  double acc = 0;
  for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x)
  {
    double2 x = a[g]; acc += x.x * x.y;
    if (NREAD == 2) { double2 y = b[g]; acc += y.x * y.y; }
  }
  if (acc == 12345.678) out[0] = acc;
}

template <typename F> float timeit(F f, int reps)
{
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps * 1000.f;
}

int main(int argc, char ** argv)
{
  const size_t sites = argc > 1 ? atol(argv[1]) : 1000000;
  const size_t total = sites * 8; // double2 granules per CLV
  double2 *a, *b, *o; double * sink;
  // allocate several CLVs so consecutive launches rotate buffers like a traversal does
  const int NBUF = 12;
  std::vector<double2 *> bufs(NBUF);
  for (auto & p : bufs) { CK(hipMalloc(&p, total * sizeof(double2))); CK(hipMemset(p, 0, total * sizeof(double2))); }
  CK(hipMalloc(&sink, 64));
  a = bufs[0]; b = bufs[1]; o = bufs[2];
  const double mb3 = total * 16.0 * 3 / 1e6, mb1 = total * 16.0 / 1e6;
  int rot = 0;
  auto next = [&]() { a = bufs[rot % NBUF]; b = bufs[(rot + 1) % NBUF]; o = bufs[(rot + 2) % NBUF]; rot += 3; };
#define RUN(U, NT, CH, NR, G, label) { float us = timeit([&]() { next(); k_rw<U, NT, CH, NR><<<G, 256>>>(a, b, o, total); }, 30); \
    double mb = (NR + 1) * mb1; printf("%-34s grid %6d: %7.1f us  %7.1f GB/s\n", label, (int)(G), us, mb / us * 1e6 / 1e9 * 1e-3 * 1e3); }
  const int full = (int)((total + 255) / 256);
  int grids[] = {1024, 2048, 4096, 8192, 16384, full};
  for (int G : grids) RUN(1, false, false, 2, G, "2r1w stride U1");
  for (int G : grids) RUN(2, false, false, 2, G, "2r1w stride U2");
  for (int G : grids) RUN(4, false, false, 2, G, "2r1w stride U4");
  for (int G : {1024, 2048, 4096, 8192}) RUN(1, false, true, 2, G, "2r1w chunk U1");
  for (int G : {1024, 2048, 4096, 8192}) RUN(4, false, true, 2, G, "2r1w chunk U4");
  for (int G : {2048, 8192, full}) RUN(1, true, false, 2, G, "2r1w stride U1 nontemporal");
  for (int G : {2048, 8192, full}) RUN(4, true, false, 2, G, "2r1w stride U4 nontemporal");
  for (int G : {2048, 8192, full}) RUN(1, false, false, 1, G, "1r1w copy U1");
  for (int G : {2048, 8192, full}) RUN(4, false, false, 1, G, "1r1w copy U4");
  for (int G : {2048, 8192, full}) RUN(1, false, false, 0, G, "0r1w write U1");
  for (int G : {2048, 8192, full}) RUN(4, false, false, 0, G, "0r1w write U4");
  for (int G : {2048, 8192, full}) RUN(4, true, false, 0, G, "0r1w write U4 nontemporal");
  for (int G : {2048, 8192, full}) { float us = timeit([&]() { next(); k_readonly<2><<<G, 256>>>(a, b, sink, total); }, 30);
    printf("%-34s grid %6d: %7.1f us  %7.1f GB/s\n", "2r0w read", G, us, 2 * mb1 / us * 1e3); }
  (void)mb3;
  return 0;
}
