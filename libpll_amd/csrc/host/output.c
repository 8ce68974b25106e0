/* output.c -- debug printers.  Replaces pll_show_pmatrix / pll_show_clv
 * (output.c:26,56 of the reference): same text layout, but the data is first
 * fetched from the device into the host mirrors.  In per-site scaling mode
 * pll_show_clv undoes the scaling exactly like the reference (output.c:74-92).
 */
#include <stdio.h>

#include "internal.h"

void pll_show_pmatrix(const pll_partition_t * cp, unsigned int index, unsigned int prec)
{
  pll_partition_t * p = (pll_partition_t *)cp;
  unsigned int i, j, k;
  const unsigned int S = p->states;
  if (!pll_amd_sync_pmatrix(p, index)) return;
  for (k = 0; k < p->rate_cats; ++k)
  {
    const double * pm = p->pmatrix[index] + (size_t)k * S * S;
    for (i = 0; i < S; ++i)
    {
      for (j = 0; j < S; ++j) printf("%+2.*f   ", prec, pm[i * S + j]);
      printf("\n");
    }
    printf("\n");
  }
}

void pll_show_clv(const pll_partition_t * cp, unsigned int clv_index, int scaler_index,
                  unsigned int prec)
{
  pll_partition_t * p = (pll_partition_t *)cp;
  unsigned int i, j, k;
  const unsigned int S = p->states, R = p->rate_cats;
  const unsigned int * scaler = NULL;
  const int per_rate = (p->attributes & PLL_ATTRIB_RATE_SCALERS) ? 1 : 0;

  if ((p->attributes & PLL_ATTRIB_PATTERN_TIP) && clv_index < p->tips) return;
  if (!pll_amd_sync_clv(p, clv_index)) return;
  if (scaler_index != PLL_SCALE_BUFFER_NONE)
  {
    if (!pll_amd_sync_scaler(p, (unsigned int)scaler_index)) return;
    scaler = p->scale_buffer[scaler_index];
  }

  printf("[ ");
  for (i = 0; i < p->sites; ++i)
  {
    printf("{");
    for (j = 0; j < R; ++j)
    {
      printf("(");
      for (k = 0; k < S; ++k)
      {
        double v = p->clv[clv_index][((size_t)i * R + j) * S + k];
        if (scaler)
        {
          unsigned int s = per_rate ? scaler[(size_t)i * R + j] : scaler[i];
          /* undo the 2^256 factors by plain multiplication, as the reference does */
          for (; s; --s) v *= PLL_SCALE_THRESHOLD;
        }
        printf("%.*f", prec, v);
        if (k < S - 1) printf(",");
      }
      printf(")");
      if (j < R - 1) printf(",");
    }
    printf("} ");
  }
  printf("]\n");
}
