/* core_api.c -- the reference's array-level entry points (pll.h:827-1013,1659): pll_core_*.
 *
 * These take every operand as a HOST array and return results in host arrays.  There is no
 * CPU compute path in this library, so each call stages its operands into a scratch device
 * context (kept per thread and reused while the geometry stays the same), runs the very
 * kernels the partition-level API runs, and copies the results back.  They are PCIe-bound by
 * construction; a client that cares about speed keeps its data in a partition.
 *
 * Layout: arrays are UNPADDED (states_padded == states, as everywhere in this library,
 * include/pll_amd.h).  An ISA bit in `attrib` is accepted where the reference's padding for it is
 * none (4, 8, 20 states ...) and refused with PLL_ERROR_PARAM_INVALID where the reference would
 * expect padded rows (5 or 7 states under AVX: core_layout_ok below); otherwise of `attrib` only
 * PLL_ATTRIB_RATE_SCALERS matters.  Arithmetic order is the AVX2-flag path's for 4 and 20
 * states and the plain C kernels' otherwise -- see numerics.hpp.
 *
 * Reference implementations replaced (src/): core_partials.c:82 (tt), :354 (ti), :510 (ii),
 * :725 (create_lookup); core_likelihood.c:25 (root), :412/:211 (edge ti), :726 (edge ii);
 * core_derivatives.c:125 (sumtable ii), :277 (sumtable ti), :501 (derivatives);
 * core_pmatrix.c:24.
 */
#include <pthread.h>
#include <stdio.h>

#include "internal.h"

/* ---- the scratch contexts of the calling thread ----------------------------------------
 * A traversal through pll_core_update_partial_tt / _ti / _ii changes shape on almost every call (tips,
 * CLVs, scale buffers differ), so ONE kept context would be torn down and rebuilt -- stream, arenas,
 * pinned buffers -- call after call.  A few are kept, least recently used out first.
 * They live until pll_amd_core_release(), or until the thread exits: a pthread key's destructor releases them
 * (six device contexts leaked per worker thread that never called the release function; the main
 * thread's are simply still there when the process ends). */
#define CORE_CTX_KEPT 6
static __thread struct
{
  pllhip_ctx_t * ctx;
  pllhip_shape_t shape;
  unsigned long long stamp;
} t_kept[CORE_CTX_KEPT];
static __thread unsigned long long t_clock;

static pthread_key_t t_kept_key;
static pthread_once_t t_kept_once = PTHREAD_ONCE_INIT;
static void kept_at_thread_exit(void * unused)
{
  (void)unused;
  pll_amd_core_release();
}
static void kept_make_key(void) { (void)pthread_key_create(&t_kept_key, kept_at_thread_exit); }

void pll_amd_core_release(void)
{
  int i;
  for (i = 0; i < CORE_CTX_KEPT; ++i)
  {
    if (t_kept[i].ctx) pllhip_ctx_destroy(t_kept[i].ctx);
    t_kept[i].ctx = NULL;
  }
}

extern int pll_amd_core_device(void); /* partition.c: the device a new partition would bind to */

/* The arrays of pll_core_* are UNPADDED here.  The reference pads every row of states to
 * states_padded under its SIMD flags (pll.c:437-451: even for SSE, a multiple of 4 for AVX / AVX2):
 * a client that passes such a flag with a state count the flag would pad has laid its arrays out
 * differently -- refused, rather than read with the wrong stride. */
static int core_layout_ok(unsigned int states, unsigned int attrib)
{
  unsigned int padded = states;
  if (attrib & PLL_ATTRIB_ARCH_SSE) padded = (states + 1u) & ~1u;
  if (attrib & (PLL_ATTRIB_ARCH_AVX | PLL_ATTRIB_ARCH_AVX2 | PLL_ATTRIB_ARCH_AVX512)) padded = (states + 3u) & ~3u;
  if (padded == states) return 1;
  pll_amd_set_error(PLL_ERROR_PARAM_INVALID,
                    "pll_core_*: %u states under a SIMD attribute mean arrays padded to %u per row in the "
                    "reference; this library takes unpadded arrays: pass PLL_ATTRIB_ARCH_CPU", states, padded);
  return 0;
}

static pllhip_ctx_t * scratch(unsigned int states, unsigned int sites, unsigned int rate_cats,
                              unsigned int tips, unsigned int clv_buffers, unsigned int rate_matrices,
                              unsigned int prob_matrices, unsigned int scale_buffers, int pattern_tip,
                              unsigned int attrib)
{
  pllhip_shape_t sh;
  int rc, i, victim = 0;
  if (!core_layout_ok(states, attrib)) return NULL;
  memset(&sh, 0, sizeof(sh));
  sh.device = pll_amd_core_device();
  sh.states = states;
  sh.rate_cats = rate_cats;
  sh.sites = sites;
  sh.tips = tips;
  sh.clv_buffers = clv_buffers;
  sh.rate_matrices = rate_matrices;
  sh.prob_matrices = prob_matrices;
  sh.scale_buffers = scale_buffers;
  sh.pattern_tip = pattern_tip;
  sh.rate_scalers = (attrib & PLL_ATTRIB_RATE_SCALERS) ? 1 : 0;
  for (i = 0; i < CORE_CTX_KEPT; ++i)
    if (t_kept[i].ctx && !memcmp(&sh, &t_kept[i].shape, sizeof(sh)))
    {
      t_kept[i].stamp = ++t_clock;
      return t_kept[i].ctx;
    }
  /* an empty place, else the least recently used one */
  for (i = 1; i < CORE_CTX_KEPT; ++i)
    if (t_kept[victim].ctx && (!t_kept[i].ctx || t_kept[i].stamp < t_kept[victim].stamp)) victim = i;
  if (t_kept[victim].ctx) pllhip_ctx_destroy(t_kept[victim].ctx);
  t_kept[victim].ctx = NULL;
  if ((rc = pllhip_ctx_create(&sh, &t_kept[victim].ctx)))
  {
    pll_amd_set_error(rc == -1 ? PLL_ERROR_HIP_UNSUPPORTED : PLL_ERROR_HIP_RUNTIME,
                      "pll_core_*: cannot create the scratch device context: %s", pllhip_last_error());
    t_kept[victim].ctx = NULL;
    return NULL;
  }
  t_kept[victim].shape = sh;
  t_kept[victim].stamp = ++t_clock;
  /* (a non-NULL value makes the key's destructor run when this thread exits) */
  (void)pthread_once(&t_kept_once, kept_make_key);
  (void)pthread_setspecific(t_kept_key, (void *)t_kept);
  return t_kept[victim].ctx;
}

#define TRY(call, what)                  \
  do {                                   \
    int rc_ = (call);                    \
    if (rc_) {                           \
      pll_amd_fail_hip(rc_, what);       \
      return FAILVAL;                    \
    }                                    \
  } while (0)

static unsigned int log2_ceil(unsigned int n)
{
  unsigned int b = 0;
  while ((1u << b) < n) ++b;
  return b;
}

/* ---- CLV updates ------------------------------------------------------------------------ */

#define FAILVAL
void pll_core_update_partial_ii(unsigned int states, unsigned int sites, unsigned int rate_cats,
                                double * parent_clv, unsigned int * parent_scaler,
                                const double * left_clv, const double * right_clv,
                                const double * left_matrix, const double * right_matrix,
                                const unsigned int * left_scaler, const unsigned int * right_scaler,
                                unsigned int attrib)
{
  /* CLVs 0 (left), 1 (right), 2 (parent); scale buffers with the same numbers; matrices 0, 1 */
  pllhip_ctx_t * c = scratch(states, sites, rate_cats, 0, 3, 1, 2, 3, 0, attrib);
  pllhip_op_t op = {2, parent_scaler ? 2 : -1, 0, 0, left_scaler ? 0 : -1, 1, 1, right_scaler ? 1 : -1};
  if (!c) return;
  TRY(pllhip_put_clv(c, 0, left_clv), "pll_core_update_partial_ii");
  TRY(pllhip_put_clv(c, 1, right_clv), "pll_core_update_partial_ii");
  TRY(pllhip_put_pmatrix(c, 0, left_matrix), "pll_core_update_partial_ii");
  TRY(pllhip_put_pmatrix(c, 1, right_matrix), "pll_core_update_partial_ii");
  if (left_scaler) TRY(pllhip_put_scaler(c, 0, left_scaler), "pll_core_update_partial_ii");
  if (right_scaler) TRY(pllhip_put_scaler(c, 1, right_scaler), "pll_core_update_partial_ii");
  TRY(pllhip_update_partials(c, &op, 1), "pll_core_update_partial_ii");
  TRY(pllhip_get_clv(c, 2, parent_clv), "pll_core_update_partial_ii");
  if (parent_scaler) TRY(pllhip_get_scaler(c, 2, parent_scaler), "pll_core_update_partial_ii");
}

void pll_core_update_partial_ti(unsigned int states, unsigned int sites, unsigned int rate_cats,
                                double * parent_clv, unsigned int * parent_scaler,
                                const unsigned char * left_tipchars, const double * right_clv,
                                const double * left_matrix, const double * right_matrix,
                                const unsigned int * right_scaler, const unsigned int * tipmap,
                                unsigned int tipmap_size, unsigned int attrib)
{
  /* tip 0; CLVs 1 (right), 2 (parent); scale buffers 0 (right), 1 (parent) */
  pllhip_ctx_t * c = scratch(states, sites, rate_cats, 1, 2, 1, 2, 2, 1, attrib);
  pllhip_op_t op = {2, parent_scaler ? 1 : -1, 0, 0, -1, 1, 1, right_scaler ? 0 : -1};
  if (!c) return;
  TRY(pllhip_put_tipchars(c, 0, left_tipchars), "pll_core_update_partial_ti");
  if (tipmap && tipmap_size) TRY(pllhip_put_tipmap(c, tipmap, tipmap_size), "pll_core_update_partial_ti");
  TRY(pllhip_put_clv(c, 1, right_clv), "pll_core_update_partial_ti");
  TRY(pllhip_put_pmatrix(c, 0, left_matrix), "pll_core_update_partial_ti");
  TRY(pllhip_put_pmatrix(c, 1, right_matrix), "pll_core_update_partial_ti");
  if (right_scaler) TRY(pllhip_put_scaler(c, 0, right_scaler), "pll_core_update_partial_ti");
  TRY(pllhip_update_partials(c, &op, 1), "pll_core_update_partial_ti");
  TRY(pllhip_get_clv(c, 2, parent_clv), "pll_core_update_partial_ti");
  if (parent_scaler) TRY(pllhip_get_scaler(c, 1, parent_scaler), "pll_core_update_partial_ti");
}

void pll_core_update_partial_ti_4x4(unsigned int sites, unsigned int rate_cats, double * parent_clv,
                                    unsigned int * parent_scaler, const unsigned char * left_tipchars,
                                    const double * right_clv, const double * left_matrix,
                                    const double * right_matrix, const unsigned int * right_scaler,
                                    unsigned int attrib)
{
  /* a 4-state tip character IS its state mask (pll.c:825-845): no tipmap */
  pll_core_update_partial_ti(4, sites, rate_cats, parent_clv, parent_scaler, left_tipchars, right_clv,
                             left_matrix, right_matrix, right_scaler, NULL, 0, attrib);
}

/* The table of a tip-tip node: entry ((j << ceil(log2 maxstates)) + k) is what a site with the
 * characters j, k gets (core_partials.c:790-862; 4 states: j, k = 1..15 are the masks
 * themselves and the shift is 4, core_partials.c:665-723).  Built by the tip-tip kernel itself,
 * run over one "site" per pair of characters. */
void pll_core_create_lookup(unsigned int states, unsigned int rate_cats, double * lookup,
                            const double * left_matrix, const double * right_matrix,
                            const unsigned int * tipmap, unsigned int tipmap_size, unsigned int attrib)
{
  const unsigned int codes = states == 4 ? 16 : tipmap_size;
  const unsigned int first = states == 4 ? 1 : 0; /* the 4-state table has no rows for code 0 */
  const unsigned int shift = states == 4 ? 4 : log2_ceil(tipmap_size);
  const unsigned int pairs = (codes - first) * (codes - first);
  const size_t span = (size_t)states * rate_cats;
  pllhip_ctx_t * c;
  pllhip_op_t op = {2, -1, 0, 0, -1, 1, 1, -1};
  unsigned char * c1, * c2;
  double * rows;
  unsigned int j, k, n = 0;
  if (!pairs) return;
  c = scratch(states, pairs, rate_cats, 2, 1, 1, 2, 0, 1, attrib & ~PLL_ATTRIB_RATE_SCALERS);
  if (!c) return;
  c1 = (unsigned char *)malloc(2 * (size_t)pairs);
  rows = (double *)malloc((size_t)pairs * span * sizeof(double));
  if (!c1 || !rows)
  {
    free(c1);
    free(rows);
    pll_amd_set_error(PLL_ERROR_MEM_ALLOC, "pll_core_create_lookup: out of memory");
    return;
  }
  c2 = c1 + pairs;
  for (j = first; j < codes; ++j)
    for (k = first; k < codes; ++k, ++n)
    {
      c1[n] = (unsigned char)j;
      c2[n] = (unsigned char)k;
    }
  if (pllhip_put_tipchars(c, 0, c1) || pllhip_put_tipchars(c, 1, c2) ||
      (states != 4 && pllhip_put_tipmap(c, tipmap, tipmap_size)) ||
      pllhip_put_pmatrix(c, 0, left_matrix) || pllhip_put_pmatrix(c, 1, right_matrix) ||
      pllhip_update_partials(c, &op, 1) || pllhip_get_clv(c, 2, rows))
    pll_amd_fail_hip(-1, "pll_core_create_lookup");
  else
  {
    /* (4 states: the rows of character 0 exist in the caller's 256-row table and are zero in the
       reference, core_partials.c:665-723; pll_core_update_partial_tt uploads all 256) */
    if (states == 4)
      for (j = 0; j < 16; ++j)
      {
        memset(lookup + ((size_t)j << 4) * span, 0, span * sizeof(double));
        memset(lookup + (size_t)j * span, 0, span * sizeof(double));
      }
    for (n = 0, j = first; j < codes; ++j)
      for (k = first; k < codes; ++k, ++n)
        memcpy(lookup + (((size_t)j << shift) + k) * span, rows + (size_t)n * span, span * sizeof(double));
  }
  free(c1);
  free(rows);
}

void pll_core_create_lookup_4x4(unsigned int rate_cats, double * lookup, const double * left_matrix,
                                const double * right_matrix)
{
  pll_core_create_lookup(4, rate_cats, lookup, left_matrix, right_matrix, NULL, 0, 0);
}

void pll_core_update_partial_tt(unsigned int states, unsigned int sites, unsigned int rate_cats,
                                double * parent_clv, unsigned int * parent_scaler,
                                const unsigned char * left_tipchars, const unsigned char * right_tipchars,
                                const unsigned int * tipmap, unsigned int tipmap_size,
                                const double * lookup, unsigned int attrib)
{
  /* tips 0, 1; CLV 2 (parent); the rows come from the caller's table, not from matrices */
  const unsigned int shift = states == 4 ? 4 : log2_ceil(tipmap_size);
  const size_t rows = states == 4 ? 256 : (((size_t)(tipmap_size - 1) << shift) + tipmap_size);
  pllhip_ctx_t * c = scratch(states, sites, rate_cats, 2, 1, 1, 1, 1, 1, attrib);
  (void)tipmap;
  if (!c) return;
  TRY(pllhip_put_tipchars(c, 0, left_tipchars), "pll_core_update_partial_tt");
  TRY(pllhip_put_tipchars(c, 1, right_tipchars), "pll_core_update_partial_tt");
  TRY(pllhip_partial_tt_from_lookup(c, 2, parent_scaler ? 0 : -1, 0, 1, lookup, rows, shift),
      "pll_core_update_partial_tt");
  TRY(pllhip_get_clv(c, 2, parent_clv), "pll_core_update_partial_tt");
  if (parent_scaler) TRY(pllhip_get_scaler(c, 0, parent_scaler), "pll_core_update_partial_tt");
}

void pll_core_update_partial_tt_4x4(unsigned int sites, unsigned int rate_cats, double * parent_clv,
                                    unsigned int * parent_scaler, const unsigned char * left_tipchars,
                                    const unsigned char * right_tipchars, const double * lookup,
                                    unsigned int attrib)
{
  pll_core_update_partial_tt(4, sites, rate_cats, parent_clv, parent_scaler, left_tipchars,
                             right_tipchars, NULL, 0, lookup, attrib);
}
#undef FAILVAL

/* ---- log-likelihood ---------------------------------------------------------------------- */

/* model slots: one per distinct entry of freqs_indices */
static unsigned int max_index(const unsigned int * idx, unsigned int n)
{
  unsigned int i, m = 0;
  for (i = 0; i < n; ++i)
    if (idx[i] > m) m = idx[i];
  return m;
}

static int put_lnl_model(pllhip_ctx_t * c, unsigned int rate_cats, double * const * frequencies,
                         const double * rate_weights, const unsigned int * pattern_weights,
                         const double * invar_proportion, const int * invar_indices,
                         const unsigned int * freqs_indices)
{
  unsigned int k;
  int rc;
  for (k = 0; k < rate_cats; ++k)
  {
    const unsigned int fi = freqs_indices[k];
    if ((rc = pllhip_put_model(c, fi, NULL, NULL, NULL, frequencies[fi],
                               invar_proportion ? invar_proportion[fi] : 0.0)))
      return rc;
  }
  if ((rc = pllhip_put_rates(c, NULL, rate_weights))) return rc;
  if ((rc = pllhip_put_pattern_weights(c, pattern_weights))) return rc;
  return pllhip_put_invariant(c, invar_indices);
}

#define FAILVAL (-INFINITY)
double pll_core_edge_loglikelihood_ii(unsigned int states, unsigned int sites, unsigned int rate_cats,
                                      const double * parent_clv, const unsigned int * parent_scaler,
                                      const double * child_clv, const unsigned int * child_scaler,
                                      const double * pmatrix, double * const * frequencies,
                                      const double * rate_weights, const unsigned int * pattern_weights,
                                      const double * invar_proportion, const int * invar_indices,
                                      const unsigned int * freqs_indices, double * persite_lnl,
                                      unsigned int attrib)
{
  double lnl = -INFINITY;
  pllhip_ctx_t * c = scratch(states, sites, rate_cats, 0, 2, max_index(freqs_indices, rate_cats) + 1, 1, 2, 0, attrib);
  if (!c) return -INFINITY;
  TRY(pllhip_put_clv(c, 0, parent_clv), "pll_core_edge_loglikelihood_ii");
  TRY(pllhip_put_clv(c, 1, child_clv), "pll_core_edge_loglikelihood_ii");
  if (parent_scaler) TRY(pllhip_put_scaler(c, 0, parent_scaler), "pll_core_edge_loglikelihood_ii");
  if (child_scaler) TRY(pllhip_put_scaler(c, 1, child_scaler), "pll_core_edge_loglikelihood_ii");
  TRY(pllhip_put_pmatrix(c, 0, pmatrix), "pll_core_edge_loglikelihood_ii");
  TRY(put_lnl_model(c, rate_cats, frequencies, rate_weights, pattern_weights, invar_proportion, invar_indices,
                    freqs_indices), "pll_core_edge_loglikelihood_ii");
  TRY(pllhip_edge_loglikelihood(c, 0, parent_scaler ? 0 : -1, 1, child_scaler ? 1 : -1, 0, freqs_indices,
                                persite_lnl, &lnl), "pll_core_edge_loglikelihood_ii");
  return lnl;
}

double pll_core_edge_loglikelihood_ti(unsigned int states, unsigned int sites, unsigned int rate_cats,
                                      const double * parent_clv, const unsigned int * parent_scaler,
                                      const unsigned char * tipchars, const unsigned int * tipmap,
                                      unsigned int tipmap_size, const double * pmatrix,
                                      double * const * frequencies, const double * rate_weights,
                                      const unsigned int * pattern_weights, const double * invar_proportion,
                                      const int * invar_indices, const unsigned int * freqs_indices,
                                      double * persite_lnl, unsigned int attrib)
{
  double lnl = -INFINITY;
  /* tip 0, CLV 1 (the inner node, "parent"), scale buffer 0 */
  pllhip_ctx_t * c = scratch(states, sites, rate_cats, 1, 1, max_index(freqs_indices, rate_cats) + 1, 1, 1, 1, attrib);
  if (!c) return -INFINITY;
  TRY(pllhip_put_tipchars(c, 0, tipchars), "pll_core_edge_loglikelihood_ti");
  if (tipmap && tipmap_size) TRY(pllhip_put_tipmap(c, tipmap, tipmap_size), "pll_core_edge_loglikelihood_ti");
  TRY(pllhip_put_clv(c, 1, parent_clv), "pll_core_edge_loglikelihood_ti");
  if (parent_scaler) TRY(pllhip_put_scaler(c, 0, parent_scaler), "pll_core_edge_loglikelihood_ti");
  TRY(pllhip_put_pmatrix(c, 0, pmatrix), "pll_core_edge_loglikelihood_ti");
  TRY(put_lnl_model(c, rate_cats, frequencies, rate_weights, pattern_weights, invar_proportion, invar_indices,
                    freqs_indices), "pll_core_edge_loglikelihood_ti");
  TRY(pllhip_edge_loglikelihood(c, 1, parent_scaler ? 0 : -1, 0, -1, 0, freqs_indices, persite_lnl, &lnl),
      "pll_core_edge_loglikelihood_ti");
  return lnl;
}

double pll_core_edge_loglikelihood_ti_4x4(unsigned int sites, unsigned int rate_cats, const double * parent_clv,
                                          const unsigned int * parent_scaler, const unsigned char * tipchars,
                                          const double * pmatrix, double * const * frequencies,
                                          const double * rate_weights, const unsigned int * pattern_weights,
                                          const double * invar_proportion, const int * invar_indices,
                                          const unsigned int * freqs_indices, double * persite_lnl,
                                          unsigned int attrib)
{
  return pll_core_edge_loglikelihood_ti(4, sites, rate_cats, parent_clv, parent_scaler, tipchars, NULL, 0,
                                        pmatrix, frequencies, rate_weights, pattern_weights, invar_proportion,
                                        invar_indices, freqs_indices, persite_lnl, attrib);
}

double pll_core_root_loglikelihood(unsigned int states, unsigned int sites, unsigned int rate_cats,
                                   const double * clv, const unsigned int * scaler,
                                   double * const * frequencies, const double * rate_weights,
                                   const unsigned int * pattern_weights, const double * invar_proportion,
                                   const int * invar_indices, const unsigned int * freqs_indices,
                                   double * persite_lnl, unsigned int attrib)
{
  double lnl = -INFINITY;
  pllhip_ctx_t * c = scratch(states, sites, rate_cats, 0, 1, max_index(freqs_indices, rate_cats) + 1, 1, 1, 0, attrib);
  if (!c) return -INFINITY;
  TRY(pllhip_put_clv(c, 0, clv), "pll_core_root_loglikelihood");
  if (scaler) TRY(pllhip_put_scaler(c, 0, scaler), "pll_core_root_loglikelihood");
  TRY(put_lnl_model(c, rate_cats, frequencies, rate_weights, pattern_weights, invar_proportion, invar_indices,
                    freqs_indices), "pll_core_root_loglikelihood");
  TRY(pllhip_root_loglikelihood(c, 0, scaler ? 0 : -1, freqs_indices, persite_lnl, &lnl),
      "pll_core_root_loglikelihood");
  return lnl;
}
#undef FAILVAL

/* ---- sumtable and derivatives: per-category model arrays (derivatives.c:59-64) ------------- */

#define FAILVAL PLL_FAILURE
static int put_category_models(pllhip_ctx_t * c, unsigned int rate_cats, double * const * eigenvals,
                               double * const * eigenvecs, double * const * inv_eigenvecs,
                               double * const * freqs, const double * prop_invar)
{
  unsigned int k;
  int rc;
  for (k = 0; k < rate_cats; ++k)
    if ((rc = pllhip_put_model(c, k, eigenvals ? eigenvals[k] : NULL, eigenvecs ? eigenvecs[k] : NULL,
                               inv_eigenvecs ? inv_eigenvecs[k] : NULL, freqs[k], prop_invar ? prop_invar[k] : 0.0)))
      return rc;
  return 0;
}

int pll_core_update_sumtable_ii(unsigned int states, unsigned int sites, unsigned int rate_cats,
                                const double * parent_clv, const double * child_clv,
                                const unsigned int * parent_scaler, const unsigned int * child_scaler,
                                double * const * eigenvecs, double * const * inv_eigenvecs,
                                double * const * freqs, double * sumtable, unsigned int attrib)
{
  unsigned int pi[PLL_AMD_MAX_RATE_CATS], k;
  pllhip_ctx_t * c;
  if (rate_cats > PLL_AMD_MAX_RATE_CATS)
  {
    pll_amd_set_error(PLL_ERROR_HIP_UNSUPPORTED, "pll_core_update_sumtable_ii: more than %d rate categories", PLL_AMD_MAX_RATE_CATS);
    return PLL_FAILURE;
  }
  c = scratch(states, sites, rate_cats, 0, 2, rate_cats, 1, 2, 0, attrib);
  if (!c) return PLL_FAILURE;
  for (k = 0; k < rate_cats; ++k) pi[k] = k;
  TRY(pllhip_put_clv(c, 0, parent_clv), "pll_core_update_sumtable_ii");
  TRY(pllhip_put_clv(c, 1, child_clv), "pll_core_update_sumtable_ii");
  if (parent_scaler) TRY(pllhip_put_scaler(c, 0, parent_scaler), "pll_core_update_sumtable_ii");
  if (child_scaler) TRY(pllhip_put_scaler(c, 1, child_scaler), "pll_core_update_sumtable_ii");
  TRY(put_category_models(c, rate_cats, NULL, eigenvecs, inv_eigenvecs, freqs, NULL), "pll_core_update_sumtable_ii");
  TRY(pllhip_update_sumtable(c, 0, parent_scaler ? 0 : -1, 1, child_scaler ? 1 : -1, pi, 0), "pll_core_update_sumtable_ii");
  TRY(pllhip_get_sumtable(c, 0, sumtable), "pll_core_update_sumtable_ii");
  return PLL_SUCCESS;
}

int pll_core_update_sumtable_ti(unsigned int states, unsigned int sites, unsigned int rate_cats,
                                const double * parent_clv, const unsigned char * left_tipchars,
                                const unsigned int * parent_scaler, double * const * eigenvecs,
                                double * const * inv_eigenvecs, double * const * freqs,
                                const unsigned int * tipmap, unsigned int tipmap_size, double * sumtable,
                                unsigned int attrib)
{
  unsigned int pi[PLL_AMD_MAX_RATE_CATS], k;
  pllhip_ctx_t * c;
  if (rate_cats > PLL_AMD_MAX_RATE_CATS)
  {
    pll_amd_set_error(PLL_ERROR_HIP_UNSUPPORTED, "pll_core_update_sumtable_ti: more than %d rate categories", PLL_AMD_MAX_RATE_CATS);
    return PLL_FAILURE;
  }
  /* tip 0 (the child), CLV 1 (the inner node), scale buffer 0 */
  c = scratch(states, sites, rate_cats, 1, 1, rate_cats, 1, 1, 1, attrib);
  if (!c) return PLL_FAILURE;
  for (k = 0; k < rate_cats; ++k) pi[k] = k;
  TRY(pllhip_put_tipchars(c, 0, left_tipchars), "pll_core_update_sumtable_ti");
  if (tipmap && tipmap_size) TRY(pllhip_put_tipmap(c, tipmap, tipmap_size), "pll_core_update_sumtable_ti");
  TRY(pllhip_put_clv(c, 1, parent_clv), "pll_core_update_sumtable_ti");
  if (parent_scaler) TRY(pllhip_put_scaler(c, 0, parent_scaler), "pll_core_update_sumtable_ti");
  TRY(put_category_models(c, rate_cats, NULL, eigenvecs, inv_eigenvecs, freqs, NULL), "pll_core_update_sumtable_ti");
  TRY(pllhip_update_sumtable(c, 1, parent_scaler ? 0 : -1, 0, -1, pi, 0), "pll_core_update_sumtable_ti");
  TRY(pllhip_get_sumtable(c, 0, sumtable), "pll_core_update_sumtable_ti");
  return PLL_SUCCESS;
}

int pll_core_update_sumtable_ti_4x4(unsigned int sites, unsigned int rate_cats, const double * parent_clv,
                                    const unsigned char * left_tipchars, const unsigned int * parent_scaler,
                                    double * const * eigenvecs, double * const * inv_eigenvecs,
                                    double * const * freqs, const unsigned int * tipmap, double * sumtable,
                                    unsigned int attrib)
{
  (void)tipmap;
  return pll_core_update_sumtable_ti(4, sites, rate_cats, parent_clv, left_tipchars, parent_scaler, eigenvecs,
                                     inv_eigenvecs, freqs, NULL, 0, sumtable, attrib);
}

int pll_core_likelihood_derivatives(unsigned int states, unsigned int sites, unsigned int rate_cats,
                                    const double * rate_weights, const unsigned int * parent_scaler,
                                    const unsigned int * child_scaler, const int * invariant,
                                    const unsigned int * pattern_weights, double branch_length,
                                    const double * prop_invar, double * const * freqs, const double * rates,
                                    double * const * eigenvals, const double * sumtable, double * d_f,
                                    double * dd_f, unsigned int attrib)
{
  unsigned int pi[PLL_AMD_MAX_RATE_CATS], i, j;
  pllhip_ctx_t * c;
  double * diag;
  int rc;
  /* (the scalers only matter to the ascertainment-bias terms, core_derivatives.c:683-686,
     which the array-level call does not have) */
  (void)parent_scaler;
  (void)child_scaler;
  if (rate_cats > PLL_AMD_MAX_RATE_CATS)
  {
    pll_amd_set_error(PLL_ERROR_HIP_UNSUPPORTED, "pll_core_likelihood_derivatives: more than %d rate categories", PLL_AMD_MAX_RATE_CATS);
    return PLL_FAILURE;
  }
  c = scratch(states, sites, rate_cats, 0, 1, rate_cats, 1, 0, 0, attrib);
  if (!c) return PLL_FAILURE;
  for (i = 0; i < rate_cats; ++i) pi[i] = i;
  TRY(put_category_models(c, rate_cats, eigenvals, NULL, NULL, freqs, prop_invar), "pll_core_likelihood_derivatives");
  TRY(pllhip_put_rates(c, rates, rate_weights), "pll_core_likelihood_derivatives");
  TRY(pllhip_put_pattern_weights(c, pattern_weights), "pll_core_likelihood_derivatives");
  TRY(pllhip_put_invariant(c, invariant), "pll_core_likelihood_derivatives");
  TRY(pllhip_put_sumtable(c, 0, sumtable), "pll_core_likelihood_derivatives");
  /* e^{lambda r t} and its t-derivatives: core_derivatives.c:560-575, same expression order */
  diag = (double *)malloc((size_t)rate_cats * states * 4 * sizeof(double));
  if (!diag)
  {
    pll_amd_set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate memory for diagptable");
    return PLL_FAILURE;
  }
  for (i = 0; i < rate_cats; ++i)
  {
    const double ki = rates[i] / (1.0 - (prop_invar ? prop_invar[i] : 0.0));
    double * dp = diag + (size_t)i * states * 4;
    for (j = 0; j < states; ++j, dp += 4)
    {
      dp[0] = exp(eigenvals[i][j] * ki * branch_length);
      dp[1] = eigenvals[i][j] * ki * dp[0];
      dp[2] = eigenvals[i][j] * ki * eigenvals[i][j] * ki * dp[0];
      dp[3] = 0;
    }
  }
  rc = pllhip_likelihood_derivatives(c, 0, -1, -1, pi, diag, d_f, dd_f);
  free(diag);
  if (rc) return pll_amd_fail_hip(rc, "pll_core_likelihood_derivatives");
  return PLL_SUCCESS;
}

/* ---- P-matrices ---------------------------------------------------------------------------- */

int pll_core_update_pmatrix(double ** pmatrix, unsigned int states, unsigned int rate_cats,
                            const double * rates, const double * branch_lengths,
                            const unsigned int * matrix_indices, const unsigned int * params_indices,
                            const double * prop_invar, double * const * eigenvals,
                            double * const * eigenvecs, double * const * inv_eigenvecs, unsigned int count,
                            unsigned int attrib)
{
  const unsigned int models = max_index(params_indices, rate_cats) + 1;
  unsigned int i, k;
  unsigned int * slots;
  pllhip_ctx_t * c;
  if (!count) return PLL_SUCCESS;
  /* matrix i of the call lives in slot i of the scratch context */
  c = scratch(states, 1, rate_cats, 0, 1, models, count, 0, 0, attrib & ~PLL_ATTRIB_RATE_SCALERS);
  if (!c) return PLL_FAILURE;
  for (k = 0; k < rate_cats; ++k)
  {
    const unsigned int pi = params_indices[k];
    TRY(pllhip_put_model(c, pi, eigenvals[pi], eigenvecs[pi], inv_eigenvecs[pi], NULL, prop_invar ? prop_invar[pi] : 0.0),
        "pll_core_update_pmatrix");
  }
  TRY(pllhip_put_rates(c, rates, NULL), "pll_core_update_pmatrix");
  slots = (unsigned int *)malloc(count * sizeof(unsigned int));
  if (!slots)
  {
    pll_amd_set_error(PLL_ERROR_MEM_ALLOC, "pll_core_update_pmatrix: out of memory");
    return PLL_FAILURE;
  }
  for (i = 0; i < count; ++i) slots[i] = i;
  i = (unsigned int)pllhip_update_pmatrices(c, params_indices, slots, branch_lengths, count);
  free(slots);
  if (i) return pll_amd_fail_hip((int)i, "pll_core_update_pmatrix");
  for (i = 0; i < count; ++i)
    TRY(pllhip_get_pmatrix(c, i, pmatrix[matrix_indices[i]]), "pll_core_update_pmatrix");
  return PLL_SUCCESS;
}
#undef FAILVAL
