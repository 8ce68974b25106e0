/* models.c -- substitution-model parameters on the host: setters, the
 * eigendecomposition of the rate matrix, invariant-site bookkeeping, and the
 * entry point that turns branch lengths into device P-matrices.
 *
 * Replaces the reference's models.c: setters :366-400, pll_update_eigen :251
 * (symmetrised Q -> Householder tridiagonalisation -> implicit QL, the
 * classic tred2/tqli pair, :24-180), pll_update_prob_matrices :333,
 * invariant sites :402-647.  The eigen system is 4x4 or 20x20 and changes only
 * when the model changes, so it stays scalar host C; the operation order of
 * the published algorithm is kept so that eigenvectors -- and through them the
 * device P-matrices -- agree with the reference to the last bit.
 */
#include <stdio.h>

#include "internal.h"

void pll_set_frequencies(pll_partition_t * p, unsigned int index, const double * f)
{
  memcpy(p->frequencies[index], f, p->states * sizeof(double));
  p->eigen_decomp_valid[index] = 0;
  pll_amd_priv(p)->model_dirty[index] = 1;
}

void pll_set_subst_params(pll_partition_t * p, unsigned int index, const double * params)
{
  memcpy(p->subst_params[index], params,
         (size_t)p->states * (p->states - 1) / 2 * sizeof(double));
  p->eigen_decomp_valid[index] = 0;
  pll_amd_priv(p)->model_dirty[index] = 1;
}

void pll_set_category_rates(pll_partition_t * p, const double * rates)
{
  memcpy(p->rates, rates, p->rate_cats * sizeof(double));
  pll_amd_priv(p)->rates_dirty = 1;
}

void pll_set_category_weights(pll_partition_t * p, const double * w)
{
  memcpy(p->rate_weights, w, p->rate_cats * sizeof(double));
  pll_amd_priv(p)->rates_dirty = 1;
}

/* ---- symmetric eigen solver ---------------------------------------------- */

#define A(r, c) a[(size_t)(r) * n + (c)]

/* Householder reduction of the symmetric matrix `a` (n x n, row-major) to
 * tridiagonal form, accumulating the transformation in `a`; d = diagonal,
 * e = sub-diagonal (e[0] unused).  Column-oriented variant, as models.c:100. */
static void tridiagonalise(double * a, unsigned int n, double * d, double * e)
{
  unsigned int I, j, k;
  for (I = n - 1; I >= 1; --I)
  {
    const unsigned int cnt = I;
    double h = 0.0, scale = 0.0;
    if (cnt > 1)
    {
      for (k = 0; k < cnt; ++k) scale += fabs(A(k, I));
      if (scale == 0.0)
        e[I] = A(cnt - 1, I);
      else
      {
        double f, g, hh;
        for (k = 0; k < cnt; ++k)
        {
          A(k, I) /= scale;
          h += A(k, I) * A(k, I);
        }
        f = A(cnt - 1, I);
        g = (f > 0) ? -sqrt(h) : sqrt(h);
        e[I] = scale * g;
        h -= f * g;
        A(cnt - 1, I) = f - g;
        f = 0.0;
        for (j = 0; j < cnt; ++j)
        {
          A(I, j) = A(j, I) / h;
          g = 0.0;
          for (k = 0; k <= j; ++k) g += A(k, j) * A(k, I);
          for (k = j + 1; k < cnt; ++k) g += A(j, k) * A(k, I);
          e[j] = g / h;
          f += e[j] * A(j, I);
        }
        hh = f / (h + h);
        for (j = 0; j < cnt; ++j)
        {
          f = A(j, I);
          g = e[j] - hh * f;
          e[j] = g;
          for (k = 0; k <= j; ++k) A(k, j) -= (f * e[k] + g * A(k, I));
        }
      }
    }
    else
      e[I] = A(cnt - 1, I);
    d[I] = h;
  }
  d[0] = 0.0;
  e[0] = 0.0;
  for (I = 0; I < n; ++I)
  {
    const unsigned int cnt = I;
    if (d[I] != 0.0)
      for (j = 0; j < cnt; ++j)
      {
        double g = 0.0;
        for (k = 0; k < cnt; ++k) g += A(k, I) * A(j, k);
        for (k = 0; k < cnt; ++k) A(j, k) -= g * A(I, k);
      }
    d[I] = A(I, I);
    A(I, I) = 1.0;
    for (j = 0; j < cnt; ++j) A(I, j) = A(j, I) = 0.0;
  }
}

/* implicit-shift QL on the tridiagonal (d, e); rows of `a` are rotated along
 * (models.c:24).  Returns 0 if an eigenvalue needs more than 60 sweeps. */
static int ql_implicit(double * d, double * e, unsigned int n, double * a)
{
  unsigned int L, M, k;
  int I;
  for (k = 1; k < n; ++k) e[k - 1] = e[k];
  e[n - 1] = 0.0;

  for (L = 0; L < n; ++L)
  {
    unsigned int sweeps = 0;
    for (;;)
    {
      double g, r, s, c, p, f, b, dd;
      for (M = L; M + 1 < n; ++M)
      {
        dd = fabs(d[M]) + fabs(d[M + 1]);
        if (fabs(e[M]) + dd == dd) break;
      }
      if (M == L) break;
      if (++sweeps > 60) return 0;

      g = (d[L + 1] - d[L]) / (2.0 * e[L]);
      r = sqrt((g * g) + 1.0);
      g = d[M] - d[L] + e[L] / (g + ((g < 0) ? -fabs(r) : fabs(r)));
      s = c = 1.0;
      p = 0.0;
      for (I = (int)M - 1; I >= (int)L; --I)
      {
        f = s * e[I];
        b = c * e[I];
        if (fabs(f) >= fabs(g))
        {
          c = g / f;
          r = sqrt((c * c) + 1.0);
          e[I + 1] = f * r;
          c *= (s = 1.0 / r);
        }
        else
        {
          s = f / g;
          r = sqrt((s * s) + 1.0);
          e[I + 1] = g * r;
          s *= (c = 1.0 / r);
        }
        g = d[I + 1] - p;
        r = (d[I] - g) * s + 2.0 * c * b;
        p = s * r;
        d[I + 1] = g + p;
        g = c * r - b;
        for (k = 0; k < n; ++k)
        {
          f = A(I + 1, k);
          A(I + 1, k) = s * A(I, k) + c * f;
          A(I, k) = c * A(I, k) - s * f;
        }
      }
      d[L] = d[L] - p;
      e[L] = g;
      e[M] = 0.0;
    }
  }
  return 1;
}

/* sqrt(pi) Q sqrt(pi)^-1, normalised to one expected substitution per unit
 * time (create_ratematrix, models.c:182-249) */
static double * symmetric_ratematrix(const double * params, const double * freqs, unsigned int n)
{
  unsigned int i, j, k = 0;
  const unsigned int np = n * (n - 1) / 2;
  double mean = 0.0;
  double * a = (double *)calloc((size_t)n * n, sizeof(double));
  double * pn = (double *)malloc(np * sizeof(double));
  if (!a || !pn)
  {
    free(a);
    free(pn);
    return NULL;
  }
  memcpy(pn, params, np * sizeof(double));
  if (pn[np - 1] > 0.0)
    for (i = 0; i < np; ++i) pn[i] /= pn[np - 1];

  for (i = 0; i < n; ++i)
    for (j = i + 1; j < n; ++j)
    {
      const double factor = pn[k++];
      A(i, j) = A(j, i) = factor * sqrt(freqs[i] * freqs[j]);
      A(i, i) -= factor * freqs[j];
      A(j, j) -= factor * freqs[i];
    }
  for (i = 0; i < n; ++i) mean += freqs[i] * (-A(i, i));
  for (i = 0; i < n * n; ++i) a[i] /= mean;
  free(pn);
  return a;
}

int pll_amd_eigen_decompose(unsigned int n, const double * subst_params, const double * freqs,
                            double * eigenvals, double * evecs, double * inv)
{
  unsigned int i, j;
  double * a = symmetric_ratematrix(subst_params, freqs, n);
  double * d = (double *)malloc(n * sizeof(double));
  double * e = (double *)malloc(n * sizeof(double));
  if (!a || !d || !e)
  {
    free(a);
    free(d);
    free(e);
    pll_amd_set_error(PLL_ERROR_MEM_ALLOC, "Unable to allocate enough memory.");
    return PLL_FAILURE;
  }
  tridiagonalise(a, n, d, e);
  if (!ql_implicit(d, e, n, a))
  {
    free(a);
    free(d);
    free(e);
    pll_amd_set_error(PLL_ERROR_PARAM_INVALID, "Eigendecomposition did not converge.");
    return PLL_FAILURE;
  }
  memcpy(evecs, a, (size_t)n * n * sizeof(double));
  memcpy(eigenvals, d, n * sizeof(double));
  /* inverse = transpose, then undo the sqrt(pi) similarity (models.c:301-320) */
  for (i = 0; i < n; ++i)
    for (j = 0; j < n; ++j) inv[i * n + j] = evecs[j * n + i];
  for (i = 0; i < n; ++i)
    for (j = 0; j < n; ++j) inv[i * n + j] /= sqrt(freqs[i]);
  for (i = 0; i < n; ++i)
    for (j = 0; j < n; ++j) evecs[i * n + j] *= sqrt(freqs[j]);
  free(a);
  free(d);
  free(e);
  return PLL_SUCCESS;
}

int pll_update_eigen(pll_partition_t * p, unsigned int index)
{
  if (!pll_amd_eigen_decompose(p->states, p->subst_params[index], p->frequencies[index],
                               p->eigenvals[index], p->eigenvecs[index],
                               p->inv_eigenvecs[index]))
    return PLL_FAILURE;
  p->eigen_decomp_valid[index] = 1;
  pll_amd_priv(p)->model_dirty[index] = 1;
  return PLL_SUCCESS;
}
#undef A

int pll_amd_flush_model(pll_partition_t * p)
{
  pll_amd_partition_t * q = pll_amd_priv(p);
  unsigned int i;
  int rc;
  for (i = 0; i < p->rate_matrices; ++i)
  {
    if (!q->model_dirty[i]) continue;
    rc = pllhip_put_model(q->ctx, i, p->eigenvals[i], p->eigenvecs[i], p->inv_eigenvecs[i],
                          p->frequencies[i], p->prop_invar[i]);
    if (rc) return pll_amd_fail_hip(rc, "model upload");
    q->model_dirty[i] = 0;
  }
  if (q->rates_dirty)
  {
    if ((rc = pllhip_put_rates(q->ctx, p->rates, p->rate_weights)))
      return pll_amd_fail_hip(rc, "rates upload");
    q->rates_dirty = 0;
  }
  if (q->tipmap_dirty && p->tipmap)
  {
    if ((rc = pllhip_put_tipmap(q->ctx, p->tipmap, p->maxstates)))
      return pll_amd_fail_hip(rc, "tipmap upload");
    q->tipmap_dirty = 0;
  }
  return PLL_SUCCESS;
}

int pll_update_prob_matrices(pll_partition_t * p, const unsigned int * params_indices,
                             const unsigned int * matrix_indices, const double * branch_lengths,
                             unsigned int count)
{
  unsigned int n;
  int rc;
  /* lazy eigendecomposition, like models.c:342-349 */
  for (n = 0; n < p->rate_cats; ++n)
    if (!p->eigen_decomp_valid[params_indices[n]])
      if (!pll_update_eigen(p, params_indices[n])) return PLL_FAILURE;
  if (!pll_amd_flush_model(p)) return PLL_FAILURE;
  rc = pllhip_update_pmatrices(pll_amd_priv(p)->ctx, params_indices, matrix_indices,
                               branch_lengths, count);
  if (rc) return pll_amd_fail_hip(rc, "P-matrix update");
  if (PLL_AMD_MIRRORS(p) && count)
  {
    /* the mirrors of the matrices just computed: ONE copy of the range they span (the host block and the device
       arena are laid out alike, partition.c; a copy per matrix is a stream wait per matrix: 126 of them are 2-3 ms) */
    unsigned int lo = matrix_indices[0], hi = matrix_indices[0];
    for (n = 1; n < count; ++n)
    {
      if (matrix_indices[n] < lo) lo = matrix_indices[n];
      if (matrix_indices[n] > hi) hi = matrix_indices[n];
    }
    rc = pllhip_get_pmatrices(pll_amd_priv(p)->ctx, lo, hi - lo + 1, p->pmatrix[lo]);
    if (rc) return pll_amd_fail_hip(rc, "P-matrix download");
  }
  return PLL_SUCCESS;
}

/* ---- invariant sites ------------------------------------------------------ */

static unsigned int all_states_mask(unsigned int states)
{
  return states >= 32 ? 0xffffffffu : ((1u << states) - 1u);
}

/* per-site AND of all tips' state masks; needs the tip data on the host */
static int site_state_intersection(pll_partition_t * p, unsigned int * acc)
{
  unsigned int i, j, k;
  const unsigned int gap = all_states_mask(p->states);
  for (j = 0; j < p->sites; ++j) acc[j] = gap;
  if (p->attributes & PLL_ATTRIB_PATTERN_TIP)
  {
    if (!p->tipchars)
    {
      for (j = 0; j < p->sites; ++j) acc[j] = 0;
      return PLL_SUCCESS;
    }
    for (i = 0; i < p->tips; ++i)
      for (j = 0; j < p->sites; ++j)
      {
        const unsigned int c = p->tipchars[i][j];
        acc[j] &= (p->states == 4) ? c : p->tipmap[c];
      }
  }
  else
  {
    /* tip CLVs live on the device: fetch them one at a time */
    const size_t span = (size_t)p->rate_cats * p->states;
    for (i = 0; i < p->tips; ++i)
    {
      if (!pll_amd_sync_clv(p, i)) return PLL_FAILURE;
      for (j = 0; j < p->sites; ++j)
      {
        unsigned int s = 0;
        const double * v = p->clv[i] + j * span;
        for (k = 0; k < p->states; ++k) s |= ((unsigned int)v[k] << k);
        acc[j] &= s;
      }
    }
  }
  return PLL_SUCCESS;
}

int pll_update_invariant_sites(pll_partition_t * p)
{
  unsigned int j;
  int rc;
  unsigned int * acc = (unsigned int *)malloc((size_t)p->sites * sizeof(unsigned int));
  /* sized for the device upload, which covers the ascertainment sites too (they are never
     invariant-model sites: -1) */
  if (!p->invariant)
  {
    const size_t n = pll_amd_priv(p)->sites_alloc;
    p->invariant = (int *)malloc(n * sizeof(int));
    if (p->invariant) for (size_t t = p->sites; t < n; ++t) p->invariant[t] = -1;
  }
  if (!acc || !p->invariant)
  {
    free(acc);
    pll_amd_set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate invariant sites array.");
    return PLL_FAILURE;
  }
  if (!site_state_intersection(p, acc))
  {
    free(acc);
    return PLL_FAILURE;
  }
  /* a single surviving state -> its index, otherwise -1 (models.c:637-645) */
  for (j = 0; j < p->sites; ++j)
    p->invariant[j] = (__builtin_popcount(acc[j]) == 1) ? __builtin_ctz(acc[j]) : -1;
  free(acc);
  rc = pllhip_put_invariant(pll_amd_priv(p)->ctx, p->invariant);
  if (rc) return pll_amd_fail_hip(rc, "invariant upload");
  return PLL_SUCCESS;
}

unsigned int pll_count_invariant_sites(pll_partition_t * p, unsigned int * state_inv_count)
{
  unsigned int j, count = 0;
  if (state_inv_count) memset(state_inv_count, 0, p->states * sizeof(unsigned int));
  if (p->invariant)
  {
    for (j = 0; j < p->sites; ++j)
      if (p->invariant[j] > -1)
      {
        count += p->pattern_weights[j];
        if (state_inv_count) state_inv_count[p->invariant[j]]++;
      }
  }
  else
  {
    unsigned int * acc = (unsigned int *)malloc((size_t)p->sites * sizeof(unsigned int));
    if (!acc || !site_state_intersection(p, acc))
    {
      free(acc);
      return 0;
    }
    for (j = 0; j < p->sites; ++j)
      if (__builtin_popcount(acc[j]) == 1)
      {
        count += p->pattern_weights[j];
        if (state_inv_count) state_inv_count[__builtin_ctz(acc[j])]++;
      }
    free(acc);
  }
  return count;
}

int pll_update_invariant_sites_proportion(pll_partition_t * p, unsigned int index, double pinv)
{
  /* models.c:407-414 */
  if (pinv != 0.0 && (p->attributes & PLL_ATTRIB_AB_MASK))
  {
    pll_amd_set_error(PLL_ERROR_INVAR_INCOMPAT,
                      "Invariant sites are not compatible with asc bias correction");
    return PLL_FAILURE;
  }
  if (pinv < 0 || pinv >= 1)
  {
    pll_amd_set_error(PLL_ERROR_INVAR_PROPORTION, "Invalid proportion of invariant sites (%f)", pinv);
    return PLL_FAILURE;
  }
  if (index >= p->rate_matrices)
  {
    pll_amd_set_error(PLL_ERROR_INVAR_PARAMINDEX, "Invalid params index (%d)", index);
    return PLL_FAILURE;
  }
  if (pinv > 0.0 && !p->invariant)
    if (!pll_update_invariant_sites(p))
    {
      pll_amd_set_error(PLL_ERROR_INVAR_NONEFOUND, "No invariant sites found");
      return PLL_FAILURE;
    }
  p->prop_invar[index] = pinv;
  pll_amd_priv(p)->model_dirty[index] = 1;
  return PLL_SUCCESS;
}
