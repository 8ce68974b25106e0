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

#define PLL_AMD_MAX_RATE_CATS 64 /* = PLLHIP_MAX_RATE_CATS of the shim */
#define PLL_AMD_MAGIC 0x504c4c414d443031ull /* "PLLAMD01" */

/* site repeats (repeats.c): classes of one CLV slot */
typedef struct pll_amd_node_repeats
{
  unsigned int classes;     /* rows the CLV is stored in; 0 = one per site */
  unsigned int * site_id;   /* host copy of the site -> row map, fetched for mirrors only */
  int site_id_valid;
  unsigned int gen;         /* bumped whenever the classes (or a tip's characters) change */
  unsigned int sig[4];      /* (child1, its gen, child2, its gen) the classes were built from */
  int sig_valid;
  unsigned int touched;     /* rep_epoch of the last op of the current piece that used the slot */
} pll_amd_node_repeats_t;

typedef struct pll_amd_partition
{
  pll_partition_t pub;          /* MUST be first */
  unsigned long long magic;
  pllhip_ctx_t * ctx;
  unsigned int sites_alloc;     /* sites (+ states when asc-bias sites are allocated) */
  int * model_dirty;            /* [rate_matrices] eigen/freqs/pinv need upload */
  int rates_dirty;
  int tipmap_dirty;
  /* device sumtable slots keyed by the caller's host pointer: as many as are alive, up to
     sumtable_cap (pllhip_sumtable_budget); then the least recently used one is recycled
     and its key remembered, so that a later use of it fails instead of reading the
     caller's never-written host buffer */
  const double * sumtable_key[PLLHIP_SUMTABLE_MAX_SLOTS];
  unsigned long long sumtable_stamp[PLLHIP_SUMTABLE_MAX_SLOTS];
  unsigned int sumtable_used, sumtable_cap;
  unsigned long long sumtable_clock;
  /* every key whose device table was recycled and that has not been produced or forgotten since:
     a growable set (a ring of 64 let older keys drop out, and a later use of such a buffer
     uploaded host memory nobody had written) */
  const double ** sumtable_evicted;
  unsigned int sumtable_evicted_n, sumtable_evicted_cap;
  /* PLL_ATTRIB_SITE_REPEATS: per CLV slot, and which CLV each scale buffer belongs to */
  pll_amd_node_repeats_t * rep;
  int * scaler_owner;
  unsigned int rep_epoch;
  /* (round 6) host mirrors kept current by the library itself: partitions whose CLVs together stay below
     PLL_AMD_AUTO_MIRROR_MB (default 64 MB; 0: never) behave as under pll_amd_set_mirror_mode(1) -- every call that
     writes a CLV, a scale buffer, a P-matrix or a sumtable on the device copies it to partition->clv[i] / ... before it
     returns, so an unmodified client that reads those arrays finds what the reference would hold there, not NULL */
  int auto_mirror;
  /* which mirrors are pinned memory of the device layer's (pllhip_host_alloc: the batched copy writes them over the
     bus) rather than the C library's: [tips + clv_buffers] and [scale_buffers] flags, NULL until the first one */
  unsigned char * clv_pinned, * scaler_pinned;
} pll_amd_partition_t;
#define PLL_AMD_MIRRORS(p) (pll_amd_mirror_mode || pll_amd_priv(p)->auto_mirror)

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

/* repeats.c */
int pll_amd_repeats_alloc(pll_amd_partition_t * q);
void pll_amd_repeats_free(pll_amd_partition_t * q);
void pll_amd_repeats_tip_changed(pll_amd_partition_t * q, unsigned int tip);
int pll_amd_repeats_update(pll_partition_t * p, const pll_operation_t * ops, unsigned int count);
/* site -> row map of a CLV stored by class (NULL if it is stored per site or on error) */
int pll_amd_repeats_scaler_ok(pll_partition_t * p, unsigned int clv, int scaler);
const unsigned int * pll_amd_repeats_site_id(pll_partition_t * p, unsigned int clv_index);
int pll_amd_repeats_expand(void * buf, const unsigned int * site_id, unsigned int classes,
                           unsigned int sites, size_t row_bytes);

#endif
