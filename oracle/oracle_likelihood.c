/* oracle_likelihood.c -- test infrastructure, see oracle.h */
#include <limits.h>
#include <math.h>

#include "oracle.h"
#include "oracle_sums.h"

static double minlh(unsigned int d)
{
  /* scale_minlh[d-1] = (2^-256)^d built by repeated multiplication
     (core_likelihood_avx.c:1119-1128); exact powers of two */
  double f = 1.0;
  while (d--) f *= ORC_SCALE_THRESHOLD;
  return f;
}

/* per-site scaler bookkeeping shared by the edge kernels
 * (core_likelihood_avx.c:1136-1161): returns the common per-site count and
 * fills rel[k] with the capped per-rate excess (all 0 in per-site mode) */
static unsigned int site_scalings(const unsigned int * ps, const unsigned int * cs, size_t n,
                                  unsigned int R, int per_rate, unsigned int * rel)
{
  unsigned int k, mn;
  if (!per_rate)
  {
    for (k = 0; k < R; ++k) rel[k] = 0;
    return (ps ? ps[n] : 0u) + (cs ? cs[n] : 0u);
  }
  mn = UINT_MAX;
  for (k = 0; k < R; ++k)
  {
    rel[k] = (ps ? ps[n * R + k] : 0u) + (cs ? cs[n * R + k] : 0u);
    if (rel[k] < mn) mn = rel[k];
  }
  for (k = 0; k < R; ++k)
  {
    rel[k] -= mn;
    if (rel[k] > ORC_RATE_MAXDIFF) rel[k] = ORC_RATE_MAXDIFF;
  }
  return mn;
}

/* category term -> weighted contribution; guard = the `terma_r > 0` test that
 * only the 4-state AVX kernels have (core_likelihood_avx.c:1225) */
static double weigh(double terma_r, unsigned int rel, double w, double pinv, const double * freqs,
                    const int * invariant, size_t n, int guard)
{
  if (rel > 0) terma_r *= minlh(rel);
  if (guard && !(terma_r > 0.)) return 0.0;
  if (pinv > 0)
  {
    const double inv_lk = (!invariant || invariant[n] == -1) ? 0 : freqs[invariant[n]];
    return w * (terma_r * (1 - pinv) + inv_lk * pinv);
  }
  return terma_r * w;
}

static double finish(double terma, unsigned int scalings, unsigned int weight, double * persite,
                     size_t n)
{
  double lk = log(terma);
  if (scalings) lk += scalings * log(ORC_SCALE_THRESHOLD);
  lk *= weight;
  if (persite) persite[n] = lk;
  return lk;
}

double orc_edge_loglikelihood_ii(unsigned int S, unsigned int sites, unsigned int R,
                                 const double * pclv, const unsigned int * ps,
                                 const double * cclv, const unsigned int * cs,
                                 const double * pmat, const double * const * freqs,
                                 const double * w, const unsigned int * pw, const double * pinv,
                                 const int * invariant, double * persite, int per_rate)
{
  double logl = 0;
  size_t n;
  unsigned int k, j, r, rel[64];
  for (n = 0; n < sites; ++n)
  {
    const unsigned int sc = site_scalings(ps, cs, n, R, per_rate, rel);
    double terma = 0;
    for (k = 0; k < R; ++k)
    {
      const double * p = pclv + (n * R + k) * S, * c = cclv + (n * R + k) * S;
      const double * m = pmat + (size_t)k * S * S, * f = freqs[k];
      double terma_r;
      if (S == 4)
      {
        /* row dot, x pi, x parent, pairwise: core_likelihood_avx.c:1175-1217 */
        double e[4];
        for (j = 0; j < 4; ++j) e[j] = (f[j] * orc_dot(m + j * 4, c, 4, 0)) * p[j];
        terma_r = orc_pair4(e[0], e[1], e[2], e[3]);
      }
      else if (S == 20)
      {
        /* chunks of 4 rows, chunk sums added in order: core_likelihood_avx2.c:432-502 */
        terma_r = 0;
        for (j = 0; j < 20; j += 4)
        {
          double e[4];
          for (r = 0; r < 4; ++r) e[r] = (orc_dot(m + (j + r) * 20, c, 20, 1) * f[j + r]) * p[j + r];
          terma_r += orc_pair4(e[0], e[1], e[2], e[3]);
        }
      }
      else
      {
        /* core_likelihood.c:946-957 */
        terma_r = 0;
        for (j = 0; j < S; ++j) terma_r += p[j] * f[j] * orc_dot(m + j * S, c, S, 0);
      }
      terma += weigh(terma_r, rel[k], w[k], pinv[k], f, invariant, n, S == 4);
    }
    logl += finish(terma, sc, pw[n], persite, n);
  }
  return logl;
}

double orc_edge_loglikelihood_ti(unsigned int S, unsigned int sites, unsigned int R,
                                 const double * pclv, const unsigned int * ps,
                                 const unsigned char * tipchars, const unsigned int * tipmap,
                                 const double * pmat, const double * const * freqs,
                                 const double * w, const unsigned int * pw, const double * pinv,
                                 const int * invariant, double * persite, int per_rate)
{
  double logl = 0;
  size_t n;
  unsigned int k, j, rel[64];
  for (n = 0; n < sites; ++n)
  {
    const unsigned int sc = site_scalings(ps, NULL, n, R, per_rate, rel);
    const unsigned int mask = orc_tipmask(S, tipmap, tipchars[n]);
    double terma = 0;
    for (k = 0; k < R; ++k)
    {
      const double * p = pclv + (n * R + k) * S;
      const double * m = pmat + (size_t)k * S * S, * f = freqs[k];
      double terma_r;
      if (S == 4)
      {
        /* lookup = pi * rowsum, x parent, pairwise: core_likelihood_avx.c:257-309,352-358 */
        double e[4];
        for (j = 0; j < 4; ++j) e[j] = (f[j] * orc_masksum(m + j * 4, mask, 4)) * p[j];
        terma_r = orc_pair4(e[0], e[1], e[2], e[3]);
      }
      else if (S == 20)
      {
        /* lookup = rowsum * pi, fused strided dot with the parent: core_likelihood_avx2.c:191-276 */
        double look[20];
        for (j = 0; j < 20; ++j) look[j] = orc_masksum(m + j * 20, mask, 20) * f[j];
        terma_r = orc_dot(look, p, 20, 1);
      }
      else
      {
        /* core_likelihood.c:664-679 */
        terma_r = 0;
        for (j = 0; j < S; ++j) terma_r += p[j] * f[j] * orc_masksum(m + j * S, mask, S);
      }
      terma += weigh(terma_r, rel[k], w[k], pinv[k], f, invariant, n, S == 4);
    }
    logl += finish(terma, sc, pw[n], persite, n);
  }
  return logl;
}
