/* oracle_pmatrix.c -- test infrastructure, see oracle.h */
#include "oracle.h"
#include "oracle_sums.h"

void orc_update_pmatrix(unsigned int S, unsigned int R, double * pmat, const double * rates,
                        double t, const double * const * eigenvals,
                        const double * const * eigenvecs, const double * const * inv_eigenvecs,
                        const double * prop_invar)
{
  unsigned int n, j, k, m;
  double expd[64];
  for (n = 0; n < R; ++n, pmat += S * S)
  {
    const double * ev = eigenvecs[n], * iv = inv_eigenvecs[n];
    if (!t)
    {
      /* core_pmatrix.c:174-179 */
      for (j = 0; j < S; ++j)
        for (k = 0; k < S; ++k) pmat[j * S + k] = (j == k) ? 1.0 : 0.0;
      continue;
    }
    for (m = 0; m < S; ++m)
    {
      /* ((lambda*r)*t) [/(1-pinv)], then expm1: core_pmatrix_avx.c:117-146 */
      double arg = (eigenvals[n][m] * rates[n]) * t;
      if (prop_invar[n] > 1e-8) arg = arg / (1.0 - prop_invar[n]);
      expd[m] = expm1(arg);
    }
    for (j = 0; j < S; ++j)
      for (k = 0; k < S; ++k)
      {
        double p;
        if (S == 4)
        {
          /* core_pmatrix_avx.c:172-222 */
          p = orc_pair4((iv[j * 4 + 0] * expd[0]) * ev[0 * 4 + k], (iv[j * 4 + 1] * expd[1]) * ev[1 * 4 + k],
                        (iv[j * 4 + 2] * expd[2]) * ev[2 * 4 + k], (iv[j * 4 + 3] * expd[3]) * ev[3 * 4 + k]);
          p = p + ((j == k) ? 1.0 : 0.0);
        }
        else if (S == 20)
        {
          /* core_pmatrix_avx2.c:24-37,236-271 */
          double a[4];
          unsigned int l;
          for (l = 0; l < 4; ++l) a[l] = (iv[j * S + l] * expd[l]) * ev[l * S + k];
          for (m = 4; m < S; m += 4)
            for (l = 0; l < 4; ++l)
              a[l] = fma(iv[j * S + m + l] * expd[m + l], ev[(m + l) * S + k], a[l]);
          p = orc_pair4(a[0], a[1], a[2], a[3]);
          if (j == k) p += 1.0;
        }
        else
        {
          /* core_pmatrix.c:226-237 */
          p = (j == k) ? 1.0 : 0.0;
          for (m = 0; m < S; ++m) p += (iv[j * S + m] * expd[m]) * ev[m * S + k];
        }
        pmat[j * S + k] = p;
      }
  }
}
