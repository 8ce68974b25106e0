/* internal.h -- private view of a partition: the public pll_partition_t
 * followed by the device handle and host-side bookkeeping.  The reference
 * allocates exactly sizeof(pll_partition_t) inside the library (pll.c:421), so
 * a larger allocation is invisible to callers. */
#ifndef PLL_AMD_INTERNAL_H_
#define PLL_AMD_INTERNAL_H_

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "pll_amd.h"
#include "pllhip.h"

#define PLL_AMD_MAGIC 0x504c4c414d443031ull /* "PLLAMD01" */

typedef struct pll_amd_partition
{
  pll_partition_t pub;          /* MUST be first */
  unsigned long long magic;
  pllhip_ctx_t * ctx;
  unsigned int sites_alloc;     /* sites (+ states when asc-bias sites are allocated) */
  int * model_dirty;            /* [rate_matrices] eigen/freqs/pinv need upload */
  int rates_dirty;
  int tipmap_dirty;
  /* device sumtable slots keyed by the caller's host pointer */
  const double * sumtable_key[PLLHIP_SUMTABLE_SLOTS];
  unsigned int sumtable_next;
} pll_amd_partition_t;

static inline pll_amd_partition_t * pll_amd_priv(const pll_partition_t * p)
{
  return (pll_amd_partition_t *)p;
}

/* sets pll_errno/pll_errmsg from the shim's last error; returns PLL_FAILURE */
int pll_amd_fail_hip(int rc, const char * what);
void pll_amd_set_error(int code, const char * fmt, ...);

/* push host-side model state that changed since the last kernel launch */
int pll_amd_flush_model(pll_partition_t * partition);

extern int pll_amd_mirror_mode;

#endif
