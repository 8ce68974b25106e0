// asc_bias.hip -- ascertainment-bias correction terms.
//
// Replaces root_loglikelihood_asc_bias (likelihood.c:50-119),
// edge_loglikelihood_asc_bias_ti (likelihood.c:170-247), _ii (likelihood.c:321-414),
// compute_asc_bias_correction (likelihood.c:24-48) and the correction block of
// pll_core_likelihood_derivatives (core_derivatives.c:654-727).
//
// A partition created with an ascertainment-bias attribute carries `states`
// extra sites behind its alignment sites: site sites+n is "every tip shows
// state n".  The CLV and sumtable kernels simply run over sites+states sites
// (partials.c:33-35, derivatives.c:55-57).  What is computed here is the scalar
// that the reference's plain-C epilogues add to the site sum: one lane per
// extra site evaluates that site's likelihood (and, for derivatives, its first
// and second derivative) with the reference's loop nests and operation order,
// lane 0 adds the `states` values in order and applies the Lewis / Felsenstein
// / Stamatakis formula.  The result lands in ctx->d_asc; the final-sum step of
// the reducing kernel that follows on the same stream adds it to the total
// (ReduceOut::extra), so a corrected lnL still costs one host synchronisation.
//
// Kept from the reference, because results must match it: the epilogues index
// scale buffers per site (`scaler[sites + n]`) even when the partition uses
// per-rate scalers, and the Stamatakis form does not multiply the scaler term by
// the state weight.
#include "ctx.hpp"
#include "numerics.hpp"

#include "lnl_common.hpp"

struct AscLnlArgs
{
  const double * parent;              // CLV that carries the frequencies, at its first extra site
  const double * child;               // inner child CLV at its first extra site (EDGE_II)
  const unsigned int * pscaler;       // at the first extra site, or nullptr
  const unsigned int * cscaler;
  const double * pmat;                // [R][S][S]
  const double * freqs;               // [rate_matrices][S]
  const double * rate_weights;
  const unsigned int * state_weights; // pattern weights of the extra sites
  double * out;
  unsigned int states, rate_cats, weight_sum;
  int kind, asc_type;
  unsigned int freqs_indices[PLLHIP_MAX_RATE_CATS];
};

__device__ __forceinline__ double scale_pow(unsigned int k)
{
  // pow(2^-256, k): exact, underflows to zero like the libm call it stands for
  return ldexp(1.0, -256 * (int)(k > 8u ? 8u : k));
}

__global__ __launch_bounds__(256) void k_asc_lnl(AscLnlArgs a)
{
  extern __shared__ double s_site[];
  const unsigned int S = a.states, R = a.rate_cats;
  for (unsigned int n = threadIdx.x; n < S; n += blockDim.x)
  {
    const double * clvp = a.parent + (size_t)n * R * S;
    const double * clvc = a.child ? a.child + (size_t)n * R * S : nullptr;
    double terma = 0.0;
    for (unsigned int i = 0; i < R; ++i)
    {
      const double * fr = a.freqs + (size_t)a.freqs_indices[i] * S;
      const double * pm = a.pmat ? a.pmat + (size_t)i * S * S : nullptr;
      double terma_r = 0.0;
      for (unsigned int j = 0; j < S; ++j)
      {
        if (a.kind == ROOT) terma_r += clvp[j] * fr[j];
        else if (a.kind == EDGE_TI) terma_r += clvp[j] * fr[j] * pm[(size_t)j * S + n];
        else
        {
          double termb = 0.0;
          for (unsigned int k = 0; k < S; ++k) termb += pm[(size_t)j * S + k] * clvc[k];
          terma_r += clvp[j] * fr[j] * termb;
        }
      }
      terma += terma_r * a.rate_weights[i];
      clvp += S;
      if (clvc) clvc += S;
    }
    unsigned int scale_factors = a.pscaler ? a.pscaler[n] : 0u;
    if (a.kind == EDGE_II && a.cscaler) scale_factors += a.cscaler[n];
    double site_lk;
    if (a.asc_type == PLLHIP_AB_STAMATAKIS)
    {
      site_lk = log(terma) * (double)a.state_weights[n];
      if (scale_factors) site_lk += (double)scale_factors * log(PLLHIP_SCALE_THRESHOLD);
    }
    else
      site_lk = terma * scale_pow(scale_factors);
    s_site[n] = site_lk;
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    double base = 0.0;
    unsigned int sum_w_inv = 0;
    for (unsigned int n = 0; n < S; ++n)
    {
      base += s_site[n];
      sum_w_inv += a.state_weights[n];
    }
    double corr;
    if (a.asc_type == PLLHIP_AB_LEWIS) corr = -((double)a.weight_sum * log(1 - base));
    else if (a.asc_type == PLLHIP_AB_FELSENSTEIN) corr = (double)sum_w_inv * log(base);
    else corr = base;
    a.out[0] = corr;
  }
}

struct AscDerivArgs
{
  const double * sumtable;            // at the first extra site
  const double * diagp;               // [R][S][4]
  const double * rate_weights;
  const unsigned int * pscaler;       // at the first extra site, or nullptr
  const unsigned int * cscaler;
  const unsigned int * state_weights;
  double * out;                       // [2]: what to add to d_f, dd_f
  unsigned int states, rate_cats, weight_sum;
  int asc_type;
};

__global__ __launch_bounds__(256) void k_asc_derivatives(AscDerivArgs a)
{
  extern __shared__ double s_lk[]; // [3][states]
  const unsigned int S = a.states, R = a.rate_cats;
  for (unsigned int n = threadIdx.x; n < S; n += blockDim.x)
  {
    const double * sum = a.sumtable + (size_t)n * R * S;
    const double * diagp = a.diagp;
    double lk0 = 0.0, lk1 = 0.0, lk2 = 0.0;
    for (unsigned int i = 0; i < R; ++i)
    {
      double c0 = 0.0, c1 = 0.0, c2 = 0.0;
      for (unsigned int j = 0; j < S; ++j)
      {
        c0 += sum[j] * diagp[0];
        c1 += sum[j] * diagp[1];
        c2 += sum[j] * diagp[2];
        diagp += 4;
      }
      lk0 += c0 * a.rate_weights[i];
      lk1 += c1 * a.rate_weights[i];
      lk2 += c2 * a.rate_weights[i];
      sum += S;
    }
    unsigned int scale_factors = a.pscaler ? a.pscaler[n] : 0u;
    if (a.cscaler) scale_factors += a.cscaler[n];
    const double sc = scale_pow(scale_factors);
    s_lk[n] = lk0 * sc;
    s_lk[S + n] = lk1 * sc;
    s_lk[2 * S + n] = lk2 * sc;
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    double L0 = 0.0, L1 = 0.0, L2 = 0.0;
    unsigned int sum_w_inv = 0;
    for (unsigned int n = 0; n < S; ++n)
    {
      L0 += s_lk[n];
      L1 += s_lk[S + n];
      L2 += s_lk[2 * S + n];
      sum_w_inv += a.state_weights[n];
    }
    // derivatives of -lnL: the signs are those of core_derivatives.c:700-720
    if (a.asc_type == PLLHIP_AB_LEWIS)
    {
      const double w = (double)a.weight_sum;
      a.out[0] = w * (L1 / (L0 - 1.0));
      a.out[1] = w * (((L0 - 1.0) * L2 - L1 * L1) / ((L0 - 1.0) * (L0 - 1.0)));
    }
    else
    {
      const double w = (double)sum_w_inv;
      a.out[0] = -(w * (L1 / L0));
      a.out[1] = -(w * (((L2 * L0) - L1 * L1) / (L0 * L0)));
    }
  }
}

static int asc_buffer(pllhip_ctx * c)
{
  if (!c->d_asc) HIP_TRY(hipMalloc((void **)&c->d_asc, 4 * sizeof(double)));
  return 0;
}

// Launches the lnL correction kernel when a correction type is set; `a` is the
// argument block of the site kernel about to run (sites = ordinary sites only).
// On return *extra is what that kernel's ReduceOut::extra must point to.
int pllhip_asc_lnl(pllhip_ctx * c, const LnlArgs & a, int kind, const double ** extra)
{
  *extra = nullptr;
  if (!c->sh.asc_states || !(c->asc_type & PLLHIP_AB_MASK)) return 0;
  if (asc_buffer(c)) return -1;
  const unsigned int S = c->sh.states, R = c->sh.rate_cats;
  const size_t first = (size_t)a.sites; // ordinary sites come first
  AscLnlArgs x;
  memset(&x, 0, sizeof(x));
  x.parent = a.parent + first * R * S;
  x.child = (kind == EDGE_II) ? a.child + first * R * S : nullptr;
  x.pscaler = a.pscaler ? a.pscaler + first : nullptr; // per-site indexing, see the header
  x.cscaler = (kind == EDGE_II && a.cscaler) ? a.cscaler + first : nullptr;
  x.pmat = (kind == ROOT) ? nullptr : a.pmat;
  x.freqs = a.freqs;
  x.rate_weights = a.rate_weights;
  x.state_weights = a.pattern_weights + first;
  x.out = c->d_asc;
  x.states = S;
  x.rate_cats = R;
  x.weight_sum = c->asc_weight_sum;
  x.kind = kind;
  x.asc_type = c->asc_type & PLLHIP_AB_MASK;
  for (unsigned int k = 0; k < R; ++k) x.freqs_indices[k] = a.freqs_indices[k];
  k_asc_lnl<<<1, 256, (size_t)S * sizeof(double), c->stream>>>(x);
  HIP_TRY(hipGetLastError());
  *extra = c->d_asc;
  return 0;
}

// Same for the derivatives (Lewis / Felsenstein; the Stamatakis form needs no
// epilogue: its extra sites are ordinary weighted sites of the site kernel).
int pllhip_asc_derivatives(pllhip_ctx * c, const double * sumtable, const double * d_diagp,
                           size_t ordinary_sites, int parent_scaler, int child_scaler,
                           const double ** extra)
{
  *extra = nullptr;
  const int type = c->asc_type & PLLHIP_AB_MASK;
  if (!c->sh.asc_states || !type || type == PLLHIP_AB_STAMATAKIS) return 0;
  if (asc_buffer(c)) return -1;
  const unsigned int S = c->sh.states, R = c->sh.rate_cats;
  AscDerivArgs x;
  memset(&x, 0, sizeof(x));
  x.sumtable = sumtable + ordinary_sites * R * S;
  x.diagp = d_diagp;
  x.rate_weights = c->rate_weights;
  const unsigned int * ps = pllhip_scaler_ptr(c, parent_scaler);
  const unsigned int * cs = pllhip_scaler_ptr(c, child_scaler);
  x.pscaler = ps ? ps + ordinary_sites : nullptr;
  x.cscaler = cs ? cs + ordinary_sites : nullptr;
  x.state_weights = c->pattern_weights + ordinary_sites;
  x.out = c->d_asc;
  x.states = S;
  x.rate_cats = R;
  x.weight_sum = c->asc_weight_sum;
  x.asc_type = type;
  k_asc_derivatives<<<1, 256, (size_t)3 * S * sizeof(double), c->stream>>>(x);
  HIP_TRY(hipGetLastError());
  *extra = c->d_asc;
  return 0;
}

extern "C" int pllhip_set_asc(pllhip_ctx_t * c, int asc_type, unsigned int pattern_weight_sum)
{
  if (!c->shards.empty())
  {
    // the per-state sites live on the last shard; the correction is added to that shard's sum
    c->asc_type = asc_type;
    c->asc_weight_sum = pattern_weight_sum;
    return pllhip_set_asc(c->shards.back(), asc_type, pattern_weight_sum);
  }
  if (asc_type & ~PLLHIP_AB_MASK)
  {
    pllhip_set_error("pllhip_set_asc: not an ascertainment-bias type (%d)", asc_type);
    return -1;
  }
  if (asc_type && !c->sh.asc_states)
  {
    pllhip_set_error("pllhip_set_asc: the context holds no ascertainment-bias sites");
    return -1;
  }
  c->asc_type = asc_type;
  c->asc_weight_sum = pattern_weight_sum;
  return 0;
}
