/* oracle_derivatives.c -- test infrastructure, see oracle.h */
#include <limits.h>
#include <math.h>
#include <stdlib.h>

#include "oracle.h"
#include "oracle_sums.h"

static void rate_rescale(double * sum, unsigned int S, unsigned int R, const unsigned int * ps,
                         const unsigned int * cs, size_t n)
{
  /* core_derivatives.c:226-246,264-265 */
  unsigned int k, j, mn = UINT_MAX, v[64];
  for (k = 0; k < R; ++k)
  {
    v[k] = (ps ? ps[n * R + k] : 0u) + (cs ? cs[n * R + k] : 0u);
    if (v[k] < mn) mn = v[k];
  }
  for (k = 0; k < R; ++k)
  {
    unsigned int d = v[k] - mn;
    double f = 1.0;
    if (!d) continue;
    if (d > ORC_RATE_MAXDIFF) d = ORC_RATE_MAXDIFF;
    while (d--) f *= ORC_SCALE_THRESHOLD;
    for (j = 0; j < S; ++j) sum[k * S + j] *= f;
  }
}

void orc_update_sumtable_ii(unsigned int S, unsigned int sites, unsigned int R,
                            const double * pclv, const double * cclv, const unsigned int * ps,
                            const unsigned int * cs, const double * const * evecs,
                            const double * const * inv, const double * const * freqs,
                            double * sum, int per_rate)
{
  size_t n;
  unsigned int k, j, m;
  for (n = 0; n < sites; ++n)
  {
    for (k = 0; k < R; ++k)
    {
      const double * p = pclv + (n * R + k) * S, * c = cclv + (n * R + k) * S;
      for (j = 0; j < S; ++j)
      {
        /* core_derivatives.c:254-262 */
        double l = 0, r = 0;
        for (m = 0; m < S; ++m)
        {
          l += p[m] * freqs[k][m] * inv[k][m * S + j];
          r += evecs[k][j * S + m] * c[m];
        }
        sum[(n * R + k) * S + j] = l * r;
      }
    }
    if (per_rate) rate_rescale(sum + n * R * S, S, R, ps, cs, n);
  }
}

void orc_update_sumtable_ti(unsigned int S, unsigned int sites, unsigned int R,
                            const double * iclv, const unsigned char * tipchars,
                            const unsigned int * is, const double * const * evecs,
                            const double * const * inv, const double * const * freqs,
                            const unsigned int * tipmap, double * sum, int per_rate)
{
  size_t n;
  unsigned int k, j, m;
  for (n = 0; n < sites; ++n)
  {
    const unsigned int mask = orc_tipmask(S, tipmap, tipchars[n]);
    for (k = 0; k < R; ++k)
    {
      const double * c = iclv + (n * R + k) * S;
      for (j = 0; j < S; ++j)
      {
        /* core_derivatives.c:419-433 */
        double l = 0, r = 0;
        for (m = 0; m < S; ++m)
        {
          l += ((mask >> m) & 1u) * freqs[k][m] * inv[k][m * S + j];
          r += evecs[k][j * S + m] * c[m];
        }
        sum[(n * R + k) * S + j] = l * r;
      }
    }
    if (per_rate) rate_rescale(sum + n * R * S, S, R, is, NULL, n);
  }
}

void orc_likelihood_derivatives(unsigned int S, unsigned int sites, unsigned int R,
                                const double * w, const int * invariant,
                                const unsigned int * pw, double t, const double * pinv,
                                const double * const * freqs, const double * rates,
                                const double * const * evals, const double * sum, double * d_f,
                                double * dd_f)
{
  size_t n;
  unsigned int k, j;
  double * diag = (double *)malloc((size_t)R * S * 4 * sizeof(double));
  /* core_derivatives.c:560-575 */
  for (k = 0; k < R; ++k)
  {
    const double ki = rates[k] / (1.0 - pinv[k]);
    for (j = 0; j < S; ++j)
    {
      double * dp = diag + ((size_t)k * S + j) * 4;
      dp[0] = exp(evals[k][j] * ki * t);
      dp[1] = evals[k][j] * ki * dp[0];
      dp[2] = evals[k][j] * ki * evals[k][j] * ki * dp[0];
      dp[3] = 0;
    }
  }
  *d_f = 0.0;
  *dd_f = 0.0;
  for (n = 0; n < sites; ++n)
  {
    /* core_site_likelihood_derivatives, core_derivatives.c:448-497 */
    double lk[3] = {0, 0, 0}, d1, d2;
    for (k = 0; k < R; ++k)
    {
      const double * s = sum + (n * R + k) * S;
      double c[3] = {0, 0, 0};
      for (j = 0; j < S; ++j)
      {
        const double * dp = diag + ((size_t)k * S + j) * 4;
        c[0] += s[j] * dp[0];
        c[1] += s[j] * dp[1];
        c[2] += s[j] * dp[2];
      }
      if (pinv[k] > 0)
      {
        const double inv_lk = (!invariant || invariant[n] == -1) ? 0 : freqs[k][invariant[n]] * pinv[k];
        c[0] = c[0] * (1. - pinv[k]) + inv_lk;
        c[1] = c[1] * (1. - pinv[k]);
        c[2] = c[2] * (1. - pinv[k]);
      }
      lk[0] += c[0] * w[k];
      lk[1] += c[1] * w[k];
      lk[2] += c[2] * w[k];
    }
    d1 = (-lk[1] / lk[0]);
    d2 = (d1 * d1 - (lk[2] / lk[0]));
    *d_f += pw[n] * d1;
    *dd_f += pw[n] * d2;
  }
  free(diag);
}
