/* oracle_partials.c -- test infrastructure, see oracle.h */
#include <stdlib.h>
#include <string.h>

#include "oracle.h"
#include "oracle_sums.h"

/* fill_parent_scaler (core_partials.c:24) fused with the scaling decision:
 * returns the count inherited from the children for entry i */
static unsigned int inherited(const unsigned int * l, const unsigned int * r, size_t i)
{
  return (l ? l[i] : 0u) + (r ? r[i] : 0u);
}

/* one site: x = left factor per (rate,state) already in `lx`, y likewise;
 * writes parent and applies the scaling rule of core_partials_avx.c:486-527 */
static void finish_site(unsigned int S, unsigned int R, double * parent, const double * lx,
                        const double * ry, unsigned int * pscaler, size_t n,
                        const unsigned int * ls, const unsigned int * rs, int per_rate)
{
  unsigned int k, i;
  int site_small = 1;
  for (k = 0; k < R; ++k)
  {
    int rate_small = 1;
    for (i = 0; i < S; ++i)
    {
      const double p = lx[k * S + i] * ry[k * S + i];
      parent[k * S + i] = p;
      rate_small = rate_small && (p < ORC_SCALE_THRESHOLD);
    }
    if (pscaler && per_rate)
    {
      if (rate_small)
        for (i = 0; i < S; ++i) parent[k * S + i] *= ORC_SCALE_FACTOR;
      pscaler[n * R + k] = inherited(ls, rs, n * R + k) + (rate_small ? 1u : 0u);
    }
    site_small = site_small && rate_small;
  }
  if (pscaler && !per_rate)
  {
    if (site_small)
      for (i = 0; i < S * R; ++i) parent[i] *= ORC_SCALE_FACTOR;
    pscaler[n] = inherited(ls, rs, n) + (site_small ? 1u : 0u);
  }
}

void orc_update_partial_ii(unsigned int S, unsigned int sites, unsigned int R, double * parent,
                           unsigned int * pscaler, const double * left, const double * right,
                           const double * lmat, const double * rmat, const unsigned int * ls,
                           const unsigned int * rs, int per_rate)
{
  size_t n;
  unsigned int k, i;
  double * x = (double *)malloc(2 * S * R * sizeof(double)), * y = x + S * R;
  for (n = 0; n < sites; ++n)
  {
    for (k = 0; k < R; ++k)
      for (i = 0; i < S; ++i)
      {
        x[k * S + i] = orc_dot(lmat + (k * S + i) * S, left + (n * R + k) * S, S, 1);
        y[k * S + i] = orc_dot(rmat + (k * S + i) * S, right + (n * R + k) * S, S, 1);
      }
    finish_site(S, R, parent + n * R * S, x, y, pscaler, n, ls, rs, per_rate);
  }
  free(x);
}

void orc_update_partial_ti(unsigned int S, unsigned int sites, unsigned int R, double * parent,
                           unsigned int * pscaler, const unsigned char * ltip,
                           const double * right, const double * lmat, const double * rmat,
                           const unsigned int * rs, const unsigned int * tipmap, int per_rate)
{
  size_t n;
  unsigned int k, i;
  double * x = (double *)malloc(2 * S * R * sizeof(double)), * y = x + S * R;
  for (n = 0; n < sites; ++n)
  {
    const unsigned int mask = orc_tipmask(S, tipmap, ltip[n]);
    for (k = 0; k < R; ++k)
      for (i = 0; i < S; ++i)
      {
        x[k * S + i] = orc_masksum(lmat + (k * S + i) * S, mask, S);
        /* the AVX2 flag dispatches the AVX (mul,add) kernel here: core_partials.c:428-443 */
        y[k * S + i] = orc_dot(rmat + (k * S + i) * S, right + (n * R + k) * S, S, 0);
      }
    finish_site(S, R, parent + n * R * S, x, y, pscaler, n, NULL, rs, per_rate);
  }
  free(x);
}

void orc_update_partial_tt(unsigned int S, unsigned int sites, unsigned int R, double * parent,
                           unsigned int * pscaler, const unsigned char * ltip,
                           const unsigned char * rtip, const double * lmat,
                           const double * rmat, const unsigned int * tipmap, int per_rate)
{
  size_t n;
  unsigned int k, i;
  for (n = 0; n < sites; ++n)
  {
    const unsigned int ml = orc_tipmask(S, tipmap, ltip[n]);
    const unsigned int mr = orc_tipmask(S, tipmap, rtip[n]);
    for (k = 0; k < R; ++k)
      for (i = 0; i < S; ++i)
        parent[(n * R + k) * S + i] = orc_masksum(lmat + (k * S + i) * S, ml, S) *
                                      orc_masksum(rmat + (k * S + i) * S, mr, S);
  }
  /* no scaling test; the scaler is cleared (core_partials_avx.c:598-599) */
  if (pscaler) memset(pscaler, 0, sizeof(unsigned int) * sites * (per_rate ? R : 1));
}

void orc_update_partials(unsigned int S, unsigned int sites, unsigned int R, unsigned int tips,
                         int pattern_tip, int per_rate, double * clv, unsigned int * scalers,
                         const unsigned char * tipchars, const double * pmatrix,
                         const unsigned int * tipmap, const orc_op_t * ops, unsigned int count)
{
  const size_t clv_len = (size_t)sites * R * S, sc_len = (size_t)sites * (per_rate ? R : 1);
  const size_t pm_len = (size_t)R * S * S;
  unsigned int i;
#define SC(idx) ((idx) < 0 ? NULL : scalers + (size_t)(idx) * sc_len)
  for (i = 0; i < count; ++i)
  {
    const orc_op_t * op = ops + i;
    const int t1 = pattern_tip && op->child1_clv < tips, t2 = pattern_tip && op->child2_clv < tips;
    double * par = clv + op->parent_clv * clv_len;
    const double * m1 = pmatrix + op->child1_matrix * pm_len, * m2 = pmatrix + op->child2_matrix * pm_len;
    if (t1 && t2)
      orc_update_partial_tt(S, sites, R, par, SC(op->parent_scaler),
                            tipchars + (size_t)op->child1_clv * sites,
                            tipchars + (size_t)op->child2_clv * sites, m1, m2, tipmap, per_rate);
    else if (t1)
      orc_update_partial_ti(S, sites, R, par, SC(op->parent_scaler),
                            tipchars + (size_t)op->child1_clv * sites,
                            clv + op->child2_clv * clv_len, m1, m2, SC(op->child2_scaler), tipmap,
                            per_rate);
    else if (t2) /* tip presented as the left child: partials.c:91-112 */
      orc_update_partial_ti(S, sites, R, par, SC(op->parent_scaler),
                            tipchars + (size_t)op->child2_clv * sites,
                            clv + op->child1_clv * clv_len, m2, m1, SC(op->child1_scaler), tipmap,
                            per_rate);
    else
      orc_update_partial_ii(S, sites, R, par, SC(op->parent_scaler),
                            clv + op->child1_clv * clv_len, clv + op->child2_clv * clv_len, m1, m2,
                            SC(op->child1_scaler), SC(op->child2_scaler), per_rate);
  }
#undef SC
}
