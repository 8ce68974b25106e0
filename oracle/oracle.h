/*
 * oracle.h -- CPU restatement of the reference's likelihood kernels.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load liboracle.so; nothing under
 * libpll_amd/ links, dlopens or calls it, and the product has no CPU path.
 *
 * What it is: plain scalar C99, one function per reference "core" routine,
 * written from the reference's algorithm (file:line cited per function, paths
 * relative to the reference's src/).  It restates the arithmetic of the path
 * PLL_ATTRIB_ARCH_AVX2 selects, including its summation orders:
 *   4 states   products rounded separately, (x0+x1)+(x2+x3)       [AVX kernels]
 *   20 states  4 accumulators strided by j mod 4, fused (ii, lnL) or
 *              mul+add (ti), then (a0+a1)+(a2+a3)                 [AVX2 kernels]
 *   otherwise  left-to-right sums                                 [plain C kernels]
 * Built with -ffp-contract=off -mfma: fma() is fused only where spelled out.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_reference.py checks every
 * function here bit-for-bit (4 and 20 states) against oracle/_ref/libpll_ref.so,
 * the genuine reference compiled in place by oracle/Makefile, which itself
 * reproduces the reference's own golden outputs (tests/golden/reference_out/,
 * tests/test_reference_programs.py); tests/test_golden.py checks it against
 * the committed fixtures generated from that build (tests/golden/make_golden.py).
 *
 * Layouts are the reference's: CLV [site][rate][state], P-matrix
 * [rate][state][state], scalers uint per site (per site x rate in per-rate
 * mode).  states_padded == states throughout.
 */
#ifndef ORACLE_H_
#define ORACLE_H_

#include <stddef.h>

#define ORC_SCALE_FACTOR 0x1p+256
#define ORC_SCALE_THRESHOLD 0x1p-256
#define ORC_RATE_MAXDIFF 4

/* same 8 x 32-bit layout as pll_operation_t (pll.h:249-259) */
typedef struct orc_op
{
  unsigned int parent_clv;
  int parent_scaler;
  unsigned int child1_clv;
  unsigned int child1_matrix;
  int child1_scaler;
  unsigned int child2_clv;
  unsigned int child2_matrix;
  int child2_scaler;
} orc_op_t;

/* pll_core_update_pmatrix, core_pmatrix.c:24 (+ _4x4_avx core_pmatrix_avx.c:42,
 * _20x20_avx2 core_pmatrix_avx2.c:37).  One branch, all rate categories.
 * eigen arrays are given PER CATEGORY (already resolved through params_indices). */
void orc_update_pmatrix(unsigned int states, unsigned int rate_cats, double * pmat,
                        const double * rates, double branch_length,
                        const double * const * eigenvals, const double * const * eigenvecs,
                        const double * const * inv_eigenvecs, const double * prop_invar);

/* pll_core_update_partial_ii, core_partials.c:510 (4: core_partials_avx.c:366,
 * 20: core_partials_avx2.c:568).  scaler pointers may be NULL. */
void orc_update_partial_ii(unsigned int states, unsigned int sites, unsigned int rate_cats,
                           double * parent_clv, unsigned int * parent_scaler,
                           const double * left_clv, const double * right_clv,
                           const double * left_matrix, const double * right_matrix,
                           const unsigned int * left_scaler, const unsigned int * right_scaler,
                           int per_rate_scaling);

/* pll_core_update_partial_ti, core_partials.c:354 (4: core_partials_avx.c:899,
 * 20: core_partials_avx.c:1097).  tipmap is ignored for 4 states. */
void orc_update_partial_ti(unsigned int states, unsigned int sites, unsigned int rate_cats,
                           double * parent_clv, unsigned int * parent_scaler,
                           const unsigned char * left_tipchars, const double * right_clv,
                           const double * left_matrix, const double * right_matrix,
                           const unsigned int * right_scaler, const unsigned int * tipmap,
                           int per_rate_scaling);

/* pll_core_create_lookup + pll_core_update_partial_tt, core_partials.c:725,82
 * (4: core_partials_avx.c:262,581; 20: core_partials_avx.c:146,531) */
void orc_update_partial_tt(unsigned int states, unsigned int sites, unsigned int rate_cats,
                           double * parent_clv, unsigned int * parent_scaler,
                           const unsigned char * left_tipchars,
                           const unsigned char * right_tipchars, const double * left_matrix,
                           const double * right_matrix, const unsigned int * tipmap,
                           int per_rate_scaling);

/* pll_core_edge_loglikelihood_ii, core_likelihood.c:726 (4:
 * core_likelihood_avx.c:1079, 20: core_likelihood_avx2.c:333).  freqs /
 * prop_invar are per category; invariant and persite_lnl may be NULL. */
double orc_edge_loglikelihood_ii(unsigned int states, unsigned int sites, unsigned int rate_cats,
                                 const double * parent_clv, const unsigned int * parent_scaler,
                                 const double * child_clv, const unsigned int * child_scaler,
                                 const double * pmatrix, const double * const * freqs,
                                 const double * rate_weights,
                                 const unsigned int * pattern_weights,
                                 const double * prop_invar, const int * invariant,
                                 double * persite_lnl, int per_rate_scaling);

/* pll_core_edge_loglikelihood_ti_4x4 / _ti, core_likelihood.c:211,412 (4:
 * core_likelihood_avx.c:191, 20: core_likelihood_avx2.c:111) */
double orc_edge_loglikelihood_ti(unsigned int states, unsigned int sites, unsigned int rate_cats,
                                 const double * parent_clv, const unsigned int * parent_scaler,
                                 const unsigned char * tipchars, const unsigned int * tipmap,
                                 const double * pmatrix, const double * const * freqs,
                                 const double * rate_weights,
                                 const unsigned int * pattern_weights,
                                 const double * prop_invar, const int * invariant,
                                 double * persite_lnl, int per_rate_scaling);

/* pll_core_update_sumtable_ii / _ti, core_derivatives.c:125,277 (plain-C
 * summation order; the SIMD variants differ in the last bits only) */
void orc_update_sumtable_ii(unsigned int states, unsigned int sites, unsigned int rate_cats,
                            const double * parent_clv, const double * child_clv,
                            const unsigned int * parent_scaler,
                            const unsigned int * child_scaler,
                            const double * const * eigenvecs,
                            const double * const * inv_eigenvecs, const double * const * freqs,
                            double * sumtable, int per_rate_scaling);
void orc_update_sumtable_ti(unsigned int states, unsigned int sites, unsigned int rate_cats,
                            const double * inner_clv, const unsigned char * tipchars,
                            const unsigned int * inner_scaler, const double * const * eigenvecs,
                            const double * const * inv_eigenvecs, const double * const * freqs,
                            const unsigned int * tipmap, double * sumtable,
                            int per_rate_scaling);

/* pll_core_likelihood_derivatives, core_derivatives.c:501 (site loop :448-497) */
void orc_likelihood_derivatives(unsigned int states, unsigned int sites, unsigned int rate_cats,
                                const double * rate_weights, const int * invariant,
                                const unsigned int * pattern_weights, double branch_length,
                                const double * prop_invar, const double * const * freqs,
                                const double * rates, const double * const * eigenvals,
                                const double * sumtable, double * d_f, double * dd_f);

/* pll_update_partials, partials.c:177: executes `count` ops in order over flat
 * storage.  clv: [nodes][sites*rate_cats*states] (rows of pattern tips unused);
 * scalers: [nscalers][sites or sites*rate_cats]; tipchars: [tips][sites];
 * pmatrix: [nmat][rate_cats*states*states]. */
void orc_update_partials(unsigned int states, unsigned int sites, unsigned int rate_cats,
                         unsigned int tips, int pattern_tip, int per_rate_scaling,
                         double * clv, unsigned int * scalers, const unsigned char * tipchars,
                         const double * pmatrix, const unsigned int * tipmap,
                         const orc_op_t * ops, unsigned int count);

#endif
