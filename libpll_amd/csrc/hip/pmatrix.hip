// pmatrix.hip -- transition-probability matrices on the device.
//
//   P_k(t) = I + Vinv . diag(expm1(lambda_j * r_k * t / (1 - pinv))) . V
//
// Replaces pll_core_update_pmatrix (core_pmatrix.c:24) and its AVX2-flag
// kernels (4 states: core_pmatrix_avx.c:42; 20 states: core_pmatrix_avx2.c:37).
// One workgroup per branch; a thread owns one P entry.  The three summation
// orders of the reference are reproduced exactly, and expm1 is the C library's
// own sequence (numerics.hpp), so P is bit-identical to the reference's.
// Traffic is a few KB per branch: launch-latency bound, not HBM bound.
#include "ctx.hpp"
#include "numerics.hpp"

#define PMAT_INLINE 256

struct PmatArgs
{
  double * pmatrix;              // [prob_matrices][R][S][S]
  const double * eigenvals;      // [rate_matrices][S]
  const double * eigenvecs;      // [rate_matrices][S*S]
  const double * inv_eigenvecs;  // [rate_matrices][S*S]
  const double * prop_invar;     // [rate_matrices]
  const double * rates;          // [R]
  const unsigned int * matrix_indices; // [count]
  const double * branch_lengths;       // [count]
  unsigned int states, rate_cats;
  unsigned int split;            // workgroups per branch: 1, or rate_cats (one per category; round 5)
  unsigned int params_indices[PLLHIP_MAX_RATE_CATS];
  // up to PMAT_INLINE branches travel as kernel arguments (used when matrix_indices is
  // null): no staging copy and, above all, no draining of the stream to reuse the
  // staging buffer -- a branch-length update in the middle of queued work costs a launch
  unsigned int mi_inline[PMAT_INLINE];
  double bl_inline[PMAT_INLINE];
};

__global__ __launch_bounds__(256) void k_update_pmatrix(PmatArgs a)
{
  extern __shared__ double s_expd[]; // [R][S]
  const unsigned int S = a.states, R = a.rate_cats;
  // Round 5: 20 states -- a workgroup per (branch, rate category) instead of per branch.  One branch's 1600 entries
  // are twenty-term chains over strided eigenvector columns: 6 of them per thread took 9.0 us, the longest kernel
  // of a three-op step at a few thousand sites (pll_update_prob_matrices for ONE branch + three ops + edge lnL,
  // tools/step_floor.c); with the categories side by side every thread forms one or two.  Same arithmetic per entry.
  const unsigned int b = blockIdx.x / a.split;
  const unsigned int e_lo = a.split > 1 ? (blockIdx.x % a.split) : 0u, e_n = a.split > 1 ? 1u : R; // categories [e_lo, e_lo + e_n)
  const bool inl = a.matrix_indices == nullptr;
  const double t = inl ? a.bl_inline[b] : a.branch_lengths[b];
  double * const pm = a.pmatrix + (size_t)(inl ? a.mi_inline[b] : a.matrix_indices[b]) * R * S * S;

  if (t == 0.0)
  {
    // zero-length branch: exact identity (core_pmatrix.c:174-179)
    for (unsigned int e = e_lo * S * S + threadIdx.x; e < (e_lo + e_n) * S * S; e += blockDim.x)
    {
      const unsigned int j = (e / S) % S, k = e % S;
      pm[e] = (j == k) ? 1.0 : 0.0;
    }
    return;
  }

  for (unsigned int e = e_lo * S + threadIdx.x; e < (e_lo + e_n) * S; e += blockDim.x)
  {
    const unsigned int n = e / S, m = e % S;
    const unsigned int pi = a.params_indices[n];
    const double pinv = a.prop_invar[pi];
    // ((lambda * rate) * t) [/ (1 - pinv)]: core_pmatrix_avx.c:117-130
    double arg = (a.eigenvals[pi * S + m] * a.rates[n]) * t;
    if (pinv > 1e-8) arg = arg / (1.0 - pinv);
    s_expd[e] = pll_expm1(arg);
  }
  __syncthreads();

  for (unsigned int e = e_lo * S * S + threadIdx.x; e < (e_lo + e_n) * S * S; e += blockDim.x)
  {
    const unsigned int n = e / (S * S), j = (e / S) % S, k = e % S;
    const unsigned int pi = a.params_indices[n];
    const double * __restrict__ inv = a.inv_eigenvecs + (size_t)pi * S * S + j * S;
    const double * __restrict__ vec = a.eigenvecs + (size_t)pi * S * S + k;
    const double * __restrict__ ex = s_expd + n * S;
    double p;
    if (S == 4)
    {
      // core_pmatrix_avx.c:200-222: four products, pairwise tree, then + I
      p = pairsum4((inv[0] * ex[0]) * vec[0], (inv[1] * ex[1]) * vec[4],
                   (inv[2] * ex[2]) * vec[8], (inv[3] * ex[3]) * vec[12]);
      p = p + ((j == k) ? 1.0 : 0.0);
    }
    else if (S == 20)
    {
      // core_pmatrix_avx2.c:24-37 (ONESTEP) and :236-271: accumulators strided
      // by m mod 4, first step a product, the rest fused, pairwise tree, + 1 on
      // the diagonal
      double acc[4];
      for (unsigned int l = 0; l < 4; ++l) acc[l] = (inv[l] * ex[l]) * vec[l * S];
      for (unsigned int m = 4; m < 20; m += 4)
        for (unsigned int l = 0; l < 4; ++l)
          acc[l] = fma(inv[m + l] * ex[m + l], vec[(m + l) * S], acc[l]);
      p = pairsum4(acc[0], acc[1], acc[2], acc[3]);
      if (j == k) p += 1.0;
    }
    else
    {
      // core_pmatrix.c:226-237: start from I, accumulate left to right
      p = (j == k) ? 1.0 : 0.0;
      for (unsigned int m = 0; m < S; ++m) p += (inv[m] * ex[m]) * vec[m * S];
    }
    pm[e] = p;
  }
}

extern "C" int pllhip_update_pmatrices(pllhip_ctx_t * c, const unsigned int * h_params_indices,
                                       const unsigned int * h_matrix_indices,
                                       const double * h_branch_lengths, unsigned int count)
{
  PLLHIP_ALL_SHARDS_PAR(c, pllhip_update_pmatrices(s, h_params_indices, h_matrix_indices, h_branch_lengths, count));
  if (!count) return 0;
  HIP_TRY(hipSetDevice(c->sh.device));
  PLLHIP_CERT_FIRST(c); // (a list that may have to run again must find the matrices it ran with)
  for (unsigned int i = 0; i < count; ++i)
  {
    if (h_matrix_indices[i] >= c->sh.prob_matrices)
    {
      pllhip_set_error("pllhip_update_pmatrices: matrix index %u out of range", h_matrix_indices[i]);
      return -1;
    }
    if (!(h_branch_lengths[i] >= 0.0))
    {
      pllhip_set_error("pllhip_update_pmatrices: negative branch length");
      return -1;
    }
  }
  PmatArgs a;
  a.pmatrix = c->pmatrix;
  a.eigenvals = c->eigenvals;
  a.eigenvecs = c->eigenvecs;
  a.inv_eigenvecs = c->inv_eigenvecs;
  a.prop_invar = c->prop_invar;
  a.rates = c->rates;
  a.states = c->sh.states;
  a.rate_cats = c->sh.rate_cats;
  // (a workgroup per category from 16 states on, while the branches are few: a whole tree's matrices fill the chip
  // either way)
  a.split = (c->sh.states >= 16 && c->sh.rate_cats > 1 && count <= 64) ? c->sh.rate_cats : 1u;
  for (unsigned int n = 0; n < c->sh.rate_cats; ++n)
  {
    if (h_params_indices[n] >= c->sh.rate_matrices)
    {
      pllhip_set_error("pllhip_update_pmatrices: params index %u out of range", h_params_indices[n]);
      return -1;
    }
    a.params_indices[n] = h_params_indices[n];
  }

  const size_t lds_small = (size_t)c->sh.rate_cats * c->sh.states * sizeof(double);
  if (count <= 4 * PMAT_INLINE)
  {
    a.matrix_indices = nullptr;
    a.branch_lengths = nullptr;
    for (unsigned int done = 0; done < count;)
    {
      const unsigned int n = (count - done < PMAT_INLINE) ? count - done : PMAT_INLINE;
      memcpy(a.mi_inline, h_matrix_indices + done, n * sizeof(unsigned int));
      memcpy(a.bl_inline, h_branch_lengths + done, n * sizeof(double));
      pllhip_prof_scope prof(c, PLLHIP_PROF_PMATRIX);
      k_update_pmatrix<<<n * a.split, 256, lds_small, c->stream>>>(a);
      HIP_TRY(hipGetLastError());
      done += n;
    }
    return 0;
  }
  // Long lists go through the staging buffer.  It is reused by later calls, and the
  // copy below reads it asynchronously: chunk so one chunk fits, and drain the stream
  // before the host overwrites it again.
  const size_t per = sizeof(double) + sizeof(unsigned int);
  const unsigned int chunk_max = (unsigned int)(c->stage_bytes / (2 * per));
  for (unsigned int done = 0; done < count;)
  {
    const unsigned int n = (count - done < chunk_max) ? count - done : chunk_max;
    HIP_TRY(hipStreamSynchronize(c->stream));
    double * hb = (double *)c->h_stage;
    unsigned int * hm = (unsigned int *)(hb + n);
    memcpy(hb, h_branch_lengths + done, n * sizeof(double));
    memcpy(hm, h_matrix_indices + done, n * sizeof(unsigned int));
    HIP_TRY(hipMemcpyAsync(c->d_stage, c->h_stage, n * per, hipMemcpyHostToDevice, c->stream));
    a.branch_lengths = (const double *)c->d_stage;
    a.matrix_indices = (const unsigned int *)((const double *)c->d_stage + n);
    const size_t lds = (size_t)c->sh.rate_cats * c->sh.states * sizeof(double);
    k_update_pmatrix<<<n * a.split, 256, lds, c->stream>>>(a);
    HIP_TRY(hipGetLastError());
    done += n;
  }
  return 0;
}
