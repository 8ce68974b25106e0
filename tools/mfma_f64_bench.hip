// developer tool: issue rate of v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64 on this GPU
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC, bool SMALL>
__global__ __launch_bounds__(256) void k(double * out, int iters, long long * cyc)
{
  v4d acc[NACC];
  double sacc[NACC];
  for (int i = 0; i < NACC; ++i) { acc[i] = (v4d){0, 0, 0, 0}; sacc[i] = 0; }
  double a = threadIdx.x * 0.001, b = threadIdx.x * 0.002 + 1;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < NACC; ++i)
    {
      if (SMALL) sacc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, sacc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += SMALL ? sacc[i] : (acc[i].x + acc[i].y + acc[i].z + acc[i].w);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main()
{
  double * out; long long * cyc, h;
  hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
#define RUN(NACC, SMALL, BLOCKS, label) { k<NACC, SMALL><<<BLOCKS, 256>>>(out, 10, cyc); hipDeviceSynchronize(); hipEventRecord(e0); \
  k<NACC, SMALL><<<BLOCKS, 256>>>(out, iters, cyc); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); \
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); double n = (double)iters * NACC; \
  printf("%-28s blocks %4d: %.1f memtime-ticks/MFMA, %.1f ns/MFMA/wave, %.2f TFLOP/s chip\n", label, BLOCKS, h / n, ms * 1e6 / n, \
         (SMALL ? 512.0 : 2048.0) * n * 4 * BLOCKS / (ms * 1e-3) / 1e12); }
  RUN(8, false, 256, "16x16x4 8 acc, 1 wave/SIMD");
  RUN(16, false, 256, "16x16x4 16 acc, 1 wave/SIMD");
  RUN(8, false, 512, "16x16x4 8 acc, 2 waves/SIMD");
  RUN(8, true, 256, "4x4x4_4b 8 acc, 1 wave/SIMD");
  RUN(8, true, 512, "4x4x4_4b 8 acc, 2 waves/SIMD");
  return 0;
}
