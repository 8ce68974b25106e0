// developer tool (round 6): does the mat-vec of the 20-state whole-list kernel gain from more waves?
//
// k_aa_fused holds two sub-tiles per wave (J = 2: 8 sites) in 128 + 110 registers: two waves per SIMD.  The one way to
// three waves per SIMD is one sub-tile per wave (J = 1: 4 sites, 55 slot registers).  Before rebuilding a 1,900-line
// kernel for it, this probe measures the part that such a rebuild is meant to speed up -- the mat-vec, af_matvec of
// partials_aa_fused.hip restated here statement for statement (A operands of a row group from LDS in operand order,
// eight / four MFMAs, the chains' fifth step on the vector unit, the pairwise tree) -- ALONE, for J = 2 and J = 1 at
// one, two and three waves per SIMD (a workgroup of four waves; workgroups per CU limited by their LDS), matrix
// block resident in LDS, operands in registers, nothing else in the loop:
//   site-updates per microsecond and CU = waves per CU x 4 J sites x mat-vecs per wave / time.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/aa_matvec_occupancy_bench.hip -o tools/aa_matvec_occupancy_bench.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int MAT_B = 13 * 1024; // a matrix block in operand order: 25 blocks of 512 bytes, padded

template <int J>
struct Aops
{
  double a1[4];
  double2 a5[2];
};

template <int J>
__device__ __forceinline__ void matvec(const char * mat_lane, const char * mat_cls, unsigned int lane, const double (&b)[J][5],
                                       double (&x)[J][5])
{
  double c5[J][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
  {
    const unsigned int src = ((lane & 15u) + 16u * m) * 4u;
#pragma unroll
    for (int j = 0; j < J; ++j)
    {
      const int lo = __builtin_amdgcn_ds_bpermute((int)src, __double2loint(b[j][4]));
      const int hi = __builtin_amdgcn_ds_bpermute((int)src, __double2hiint(b[j][4]));
      c5[j][m] = __hiloint2double(hi, lo);
    }
  }
#pragma unroll
  for (int t = 0; t < 5; ++t)
  {
    double a1[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) a1[m] = *reinterpret_cast<const double *>(mat_lane + (t * 5 + m) * 512);
    const double2 a50 = *reinterpret_cast<const double2 *>(mat_cls + (t * 5 + 4) * 512);
    const double2 a51 = *reinterpret_cast<const double2 *>(mat_cls + (t * 5 + 4) * 512 + 16);
    const double a5[4] = {a50.x, a50.y, a51.x, a51.y};
    double acc[J][4];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[j][m] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1[m], b[j][m], 0.0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[j][m] = fma(a5[m], c5[j][m], acc[j][m]);
#pragma unroll
    for (int j = 0; j < J; ++j)
    {
      const double sum = (acc[j][0] + acc[j][1]) + (acc[j][2] + acc[j][3]);
      x[j][t] = x[j][t] * sum;
      asm volatile("" : "+v"(x[j][t]));
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// PAD_B: extra LDS per workgroup, to limit the workgroups a CU takes (160 KB per CU)
template <int J>
__global__ __launch_bounds__(256) void k_matvec(const double * __restrict__ mat, double * out, int iters)
{
  extern __shared__ double2 lds2[];
  char * lds = reinterpret_cast<char *>(lds2);
  for (unsigned int i = threadIdx.x; i < MAT_B / 8; i += 256) reinterpret_cast<double *>(lds)[i] = mat[i];
  __syncthreads();
  const unsigned int lane = threadIdx.x & 63u;
  const char * mat_lane = lds + lane * 8u;
  const char * mat_cls = lds + ((lane >> 4) * 4u + ((lane >> 2) & 3u)) * 32u;
  double b[J][5], x[J][5];
#pragma unroll
  for (int j = 0; j < J; ++j)
#pragma unroll
    for (int t = 0; t < 5; ++t)
    {
      b[j][t] = 0.01 + 0.001 * (lane + 7 * t + 3 * j);
      x[j][t] = 1.0;
    }
  for (int it = 0; it < iters; ++it)
  {
    matvec<J>(mat_lane, mat_cls, lane, b, x);
    // (the next mat-vec's operand depends on this one's result, as a parent's does on its children's products:
    // nothing for the compiler to hoist, and the chain a wave really has)
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int t = 0; t < 5; ++t)
      {
        b[j][t] = b[j][t] + 1e-300 * x[j][t];
        x[j][t] = 1.0;
      }
  }
  double s = 0.0;
#pragma unroll
  for (int j = 0; j < J; ++j)
#pragma unroll
    for (int t = 0; t < 5; ++t) s += b[j][t];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int J>
static void run(const double * mat, double * out, int wgs_per_cu, int cus, int iters)
{
  // LDS per workgroup so that exactly wgs_per_cu fit a CU's 160 KB
  const size_t lds = (size_t)(160 * 1024 / wgs_per_cu) - 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_matvec<J>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int grid = cus * wgs_per_cu;
  k_matvec<J><<<grid, 256, lds>>>(mat, out, 10);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  k_matvec<J><<<grid, 256, lds>>>(mat, out, iters);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double per_matvec_ns = ms * 1e6 / iters;
  const double sites_per_us_cu = (double)wgs_per_cu * 4 * (4 * J) * iters / (ms * 1e3);
  printf("J = %d (%d sites per wave), %d workgroup(s) of four waves per CU = %d wave(s) per SIMD: %7.1f ns per mat-vec and wave "
         "(%5.0f cycles at 2.4 GHz), %6.2f site-mat-vecs per us and CU\n",
         J, 4 * J, wgs_per_cu, wgs_per_cu, per_matvec_ns, per_matvec_ns * 2.4, sites_per_us_cu);
}

int main()
{
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  double * mat, * out;
  CK(hipMalloc(&mat, MAT_B));
  CK(hipMalloc(&out, (size_t)cus * 4 * 256 * sizeof(double)));
  double * h = (double *)malloc(MAT_B);
  for (int i = 0; i < MAT_B / 8; ++i) h[i] = 0.05 + 1e-4 * (i % 97);
  CK(hipMemcpy(mat, h, MAT_B, hipMemcpyHostToDevice));
  printf("the mat-vec of k_aa_fused alone (matrix block in LDS, operands in registers), %d CUs\n", cus);
  for (int rep = 0; rep < 2; ++rep)
  {
    for (int w = 1; w <= 3; ++w) run<2>(mat, out, w, cus, 4000);
    for (int w = 1; w <= 4; ++w) run<1>(mat, out, w, cus, 4000);
  }
  return 0;
}
