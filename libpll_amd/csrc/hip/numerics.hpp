// numerics.hpp -- device helpers whose ONLY job is to pin floating-point
// operation order.  All .hip files are compiled with -ffp-contract=off, so a*b+c
// written with operators is two roundings; a fused multiply-add happens only
// where fma() is spelled out.  That is what lets CLVs and scaler counts match
// the reference's AVX (mul,add) and AVX2 (fmadd) kernels bit for bit.
#pragma once
#ifndef PLLHIP_NUMERICS_HOST_BUILD /* oracle/check_expm1.cpp builds this header for the host */
#include <hip/hip_runtime.h>
#endif
#include <math.h>
#include <stdint.h>

#define PLLHIP_SCALE_FACTOR 0x1p+256
#define PLLHIP_SCALE_THRESHOLD 0x1p-256
#define PLLHIP_SCALE_RATE_MAXDIFF 4

// (x0 + x1) + (x2 + x3): the horizontal-add tree every 4-wide reference
// kernel ends with (e.g. core_partials_avx.c:459-471)
__device__ __forceinline__ double pairsum4(double x0, double x1, double x2, double x3)
{
  return (x0 + x1) + (x2 + x3);
}

// 4-state row . vector, products rounded separately
__device__ __forceinline__ double dot4(const double * __restrict__ m, double v0, double v1,
                                       double v2, double v3)
{
  return pairsum4(m[0] * v0, m[1] * v1, m[2] * v2, m[3] * v3);
}

// 4-state masked row sum: entries whose bit is clear contribute +0.0
// (the masked loads of core_partials_avx.c:303-356)
__device__ __forceinline__ double masksum4(const double * __restrict__ m, unsigned int mask)
{
  return pairsum4((mask & 1u) ? m[0] : 0.0, (mask & 2u) ? m[1] : 0.0,
                  (mask & 4u) ? m[2] : 0.0, (mask & 8u) ? m[3] : 0.0);
}

// Row . vector for S % 4 == 0 in the AVX2 order: four accumulators strided by
// j mod 4, fused multiply-add, then the pairwise tree
// (core_partials_avx2.c:671-750).  FUSED=false gives the AVX order (mul, add)
// of core_partials_avx.c:1237-1262.
template <bool FUSED, typename VEC>
__device__ __forceinline__ double dot_strided4(const double * __restrict__ m, const VEC & v,
                                               unsigned int S)
{
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  for (unsigned int j = 0; j < S; j += 4)
  {
    if (FUSED)
    {
      a0 = fma(m[j + 0], v[j + 0], a0);
      a1 = fma(m[j + 1], v[j + 1], a1);
      a2 = fma(m[j + 2], v[j + 2], a2);
      a3 = fma(m[j + 3], v[j + 3], a3);
    }
    else
    {
      a0 = a0 + m[j + 0] * v[j + 0];
      a1 = a1 + m[j + 1] * v[j + 1];
      a2 = a2 + m[j + 2] * v[j + 2];
      a3 = a3 + m[j + 3] * v[j + 3];
    }
  }
  return pairsum4(a0, a1, a2, a3);
}

// plain left-to-right dot product (generic C kernels, core_partials.c:612-623)
template <typename VEC>
__device__ __forceinline__ double dot_seq(const double * __restrict__ m, const VEC & v,
                                          unsigned int S)
{
  double a = 0.0;
  for (unsigned int j = 0; j < S; ++j) a += m[j] * v[j];
  return a;
}

// sum of the row entries selected by a state bitmask, ascending
// (core_partials_avx.c:1149-1166; a single set bit returns the entry itself)
__device__ __forceinline__ double masksum_seq(const double * __restrict__ m, unsigned int mask,
                                              unsigned int S)
{
  double a = 0.0;
  for (unsigned int j = 0; j < S; ++j)
    if ((mask >> j) & 1u) a += m[j];
  return a;
}

// exp(x) - 1 with exactly the operation sequence of the C library the
// reference calls (glibc 2.35 expm1, the fdlibm algorithm with the degree-5
// polynomial split into three independent pairs).  Verified bit-identical to
// glibc on 2e7 random arguments (oracle/check_expm1.c); keeping it identical
// makes device P-matrices equal the reference's to the last bit, which in turn
// keeps scaler counts bit-exact.
__device__ __forceinline__ double pll_expm1(double x)
{
  const double o_threshold = 7.09782712893383973096e+02;
  const double ln2_hi = 6.93147180369123816490e-01;
  const double ln2_lo = 1.90821492927058770002e-10;
  const double invln2 = 1.44269504088896338700e+00;
  const double Q1 = -3.33333333333331316428e-02;
  const double Q2 = 1.58730158725481460165e-03;
  const double Q3 = -7.93650757867487942473e-05;
  const double Q4 = 4.00821782732936239552e-06;
  const double Q5 = -2.01099218183624371326e-07;
  const double huge = 1e300, tiny = 1e-300;

  const uint64_t bits = (uint64_t)__double_as_longlong(x);
  uint32_t hx = (uint32_t)(bits >> 32);
  const uint32_t lx = (uint32_t)bits;
  const uint32_t sign = hx & 0x80000000u;
  hx &= 0x7fffffffu;

  double hi, lo, c = 0.0, t, e, y;
  int k;

  if (hx >= 0x4043687Au) // |x| >= 56 ln2
  {
    if (hx >= 0x40862E42u) // |x| >= 709.78
    {
      if (hx >= 0x7ff00000u)
      {
        if (((hx & 0xfffffu) | lx) != 0) return x + x; // NaN
        return sign ? -1.0 : x;                        // exp(+-inf) - 1
      }
      if (x > o_threshold) return huge * huge;
    }
    if (sign && x + tiny < 0.0) return tiny - 1.0;
  }

  if (hx > 0x3fd62e42u) // |x| > 0.5 ln2: reduce
  {
    if (hx < 0x3FF0A2B2u) // |x| < 1.5 ln2
    {
      if (!sign) { hi = x - ln2_hi; lo = ln2_lo; k = 1; }
      else       { hi = x + ln2_hi; lo = -ln2_lo; k = -1; }
    }
    else
    {
      k = (int)(invln2 * x + (sign ? -0.5 : 0.5));
      t = (double)k;
      hi = x - t * ln2_hi;
      lo = t * ln2_lo;
    }
    x = hi - lo;
    c = (hi - x) - lo;
  }
  else if (hx < 0x3c900000u) // |x| < 2^-54
  {
    t = huge + x;
    return x - (t - (huge + x));
  }
  else
    k = 0;

  const double hfx = 0.5 * x;
  const double hxs = x * hfx;
  const double R1 = 1.0 + hxs * Q1;
  const double h2 = hxs * hxs;
  const double R2 = Q2 + hxs * Q3;
  const double h4 = h2 * h2;
  const double R3 = Q4 + hxs * Q5;
  const double r1 = R1 + h2 * R2 + h4 * R3;
  t = 3.0 - r1 * hfx;
  e = hxs * ((r1 - t) / (6.0 - x * t));
  if (k == 0) return x - (x * e - hxs);

  e = (x * (e - c) - c);
  e -= hxs;
  if (k == -1) return 0.5 * (x - e) - 0.5;
  if (k == 1)
  {
    if (x < -0.25) return -2.0 * (e - (x + 0.5));
    return 1.0 + 2.0 * (x - e);
  }
  const uint64_t kexp = ((uint64_t)(uint32_t)(k << 20)) << 32;
  if (k <= -2 || k > 56)
  {
    y = 1.0 - (e - x);
    y = __longlong_as_double((long long)((uint64_t)__double_as_longlong(y) + kexp));
    return y - 1.0;
  }
  if (k < 20)
  {
    t = __longlong_as_double((long long)(((uint64_t)(0x3ff00000u - (0x200000u >> k))) << 32));
    y = t - (e - x);
  }
  else
  {
    t = __longlong_as_double((long long)(((uint64_t)((uint32_t)(0x3ff - k) << 20)) << 32));
    y = x - (e + t);
    y += 1.0;
  }
  return __longlong_as_double((long long)((uint64_t)__double_as_longlong(y) + kexp));
}

// array view usable where the dot helpers expect operator[]
struct dview
{
  const double * p;
  __device__ __forceinline__ double operator[](unsigned int j) const { return p[j]; }
};

#ifndef PLLHIP_NUMERICS_HOST_BUILD
// true iff `lane_flag` holds on all W lanes of this lane's aligned group (W = 2..64)
template <int W>
__device__ __forceinline__ bool group_all(bool lane_flag)
{
  const unsigned long long b = __ballot(lane_flag);
  const unsigned int lane = threadIdx.x & 63u;
  const unsigned long long grp = b >> (lane & ~(unsigned int)(W - 1));
  const unsigned long long full = (W >= 64) ? ~0ull : ((1ull << W) - 1ull);
  return (grp & full) == full;
}

// 16-byte global accesses with a compile-time cache policy
typedef double pll_v2d __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ double2 ld16(const double2 * p)
{
  if (NT)
  {
    const pll_v2d v = __builtin_nontemporal_load(reinterpret_cast<const pll_v2d *>(p));
    return make_double2(v.x, v.y);
  }
  return *p;
}
template <bool NT>
__device__ __forceinline__ void st16(double2 * p, double a, double b)
{
  if (NT)
  {
    const pll_v2d v = {a, b};
    __builtin_nontemporal_store(v, reinterpret_cast<pll_v2d *>(p));
  }
  else
    *p = make_double2(a, b);
}

// ---- lane-pair helpers of the 4-state kernels (one lane per 16 bytes) ----
__device__ __forceinline__ double dpp_pair_swap(double v)
{
  // quad_perm [1,0,3,2]: every lane reads its xor-1 neighbour
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// DPP lane exchanges (VALU, no LDS crossbar): CTRL 0x4E = quad_perm [2,3,0,1] (xor 2),
// 0x141 = row_half_mirror (lane i <-> 7 - i of each 8), 0x140 = row_mirror (i <-> 15 - i)
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// Sum of v over the W = 2, 4, 8 or 16 consecutive lanes of a group whose lane pairs
// (2m, 2m+1) already hold equal values: pairwise tree (v0+v1)+(v2+v3) ..., the same
// association as xor-2, xor-4, xor-8 butterflies, every lane of the group gets the total.
template <int W>
__device__ __forceinline__ double dpp_group_sum_pairs(double v)
{
  if (W >= 4) v += dpp_move<0x4E>(v);  // the other pair of the quad
  if (W >= 8) v += dpp_move<0x141>(v); // the other quad of the 8 (all 4 lanes of a quad are equal)
  if (W >= 16) v += dpp_move<0x140>(v); // the other 8 of the 16
  return v;
}

// matrix rows 2h, 2h+1 of category k, columns split into the lane's own pair
// (2h, 2h+1) and its partner's: m[r][0..1] own, m[r][2..3] partner
struct half_rows
{
  double m[2][4];
  __device__ __forceinline__ void load(const double * __restrict__ mat, unsigned int k,
                                       unsigned int h)
  {
#pragma unroll
    for (int r = 0; r < 2; ++r)
    {
      const double * row = mat + k * 16 + (2 * h + r) * 4;
      m[r][0] = row[2 * h];
      m[r][1] = row[2 * h + 1];
      m[r][2] = row[2 - 2 * h];
      m[r][3] = row[3 - 2 * h];
    }
  }
  // row r of the matrix times the 4-vector (own.x, own.y | par.x, par.y)
  __device__ __forceinline__ double dot(int r, double2 own, double2 par) const
  {
    return (m[r][0] * own.x + m[r][1] * own.y) + (m[r][2] * par.x + m[r][3] * par.y);
  }
};

#endif
