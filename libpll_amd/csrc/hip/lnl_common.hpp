// lnl_common.hpp -- internal: argument block and per-site tail shared by the
// log-likelihood kernels (likelihood.hip, likelihood_aa_mfma.hip).
#pragma once
#include "ctx.hpp"
#include "numerics.hpp"

struct LnlArgs
{
  const double * __restrict__ parent;   // CLV carrying the frequencies side
  const double * __restrict__ child;    // inner child CLV (ii)
  const unsigned char * __restrict__ tip; // tip child codes (ti)
  const unsigned int * __restrict__ pscaler;
  const unsigned int * __restrict__ cscaler;
  const double * __restrict__ pmat;     // [R][S][S]
  const double * __restrict__ freqs;    // [rate_matrices][S]
  const double * __restrict__ prop_invar; // [rate_matrices]
  const double * __restrict__ rate_weights;
  const unsigned int * __restrict__ pattern_weights;
  const int * __restrict__ invariant;   // nullable
  const unsigned int * __restrict__ tipmap;
  const unsigned int * zero;            // device word holding 0
  double * __restrict__ persite;        // nullable
  double * __restrict__ block_partials; // [gridDim.x]
  unsigned int sites, rate_cats, states, maxstates;
  int rate_scalers;
  unsigned int freqs_indices[PLLHIP_MAX_RATE_CATS];
};

enum { EDGE_II = 0, EDGE_TI = 1, ROOT = 2 };

// (2^-256)^d for d = 1..4, exact powers of two (core_likelihood_avx.c:1119-1128)
__device__ __forceinline__ double scale_minlh(unsigned int d)
{
  return d == 1 ? 0x1p-256 : d == 2 ? 0x1p-512 : d == 3 ? 0x1p-768 : 0x1p-1024;
}

__device__ __forceinline__ double block_sum_to_partials(double v, double * __restrict__ out)
{
  __shared__ double s_wave[16];
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const unsigned int wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  if (lane == 0) s_wave[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    double t = 0.0;
    for (unsigned int w = 0; w < (blockDim.x >> 6); ++w) t += s_wave[w];
    out[blockIdx.x] = t;
  }
  return v;
}

// category term -> weighted contribution (core_likelihood_avx.c:1219-1240)
template <bool GUARD_POSITIVE>
__device__ __forceinline__ double category_term(const LnlArgs & a, double terma_r,
                                                unsigned int k, size_t n, unsigned int rel_scale)
{
  if (rel_scale > 0) terma_r *= scale_minlh(rel_scale);
  if (GUARD_POSITIVE && !(terma_r > 0.0)) return 0.0;
  const unsigned int fi = a.freqs_indices[k];
  const double pinv = a.prop_invar[fi];
  const double w = a.rate_weights[k];
  if (pinv > 0.0)
  {
    const int inv = a.invariant ? a.invariant[n] : -1;
    const double inv_lk = (inv == -1) ? 0.0 : a.freqs[(size_t)fi * a.states + inv];
    return w * (terma_r * (1.0 - pinv) + inv_lk * pinv);
  }
  return terma_r * w;
}

__device__ __forceinline__ double site_loglk(const LnlArgs & a, double terma, size_t n,
                                             unsigned int site_scalings)
{
  double lk = log(terma);
  if (site_scalings) lk += (double)site_scalings * log(PLLHIP_SCALE_THRESHOLD);
  lk *= (double)a.pattern_weights[n];
  if (a.persite) a.persite[n] = lk;
  return lk;
}


// 20-state kernels on the matrix cores; returns 1 if the case is not covered
int pllhip_launch_lnl_aa_mfma(pllhip_ctx * c, const LnlArgs & a, int kind, unsigned int * grid_out);
