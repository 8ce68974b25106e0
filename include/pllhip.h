/*
 * pllhip.h -- the thin C-ABI shim between the C host library (pll_* API,
 * include/pll_amd.h) and the hand-written HIP kernels for gfx950.
 *
 * Everything here is `extern "C"`, plain pointers, integers and sizes: no C++
 * and no torch types cross this boundary.  The host library is the only
 * intended caller, but the symbols are exported so that a binding from another
 * language can drive the device directly.
 *
 * Each compute entry point replaces one reference "core" routine; the
 * reference function it stands in for is cited as <file>:<line> relative to
 * the reference's src/ directory.  All pointers named `h_*` are HOST memory,
 * read or written synchronously during the call; device memory is never
 * exposed except through pllhip_dev_* accessors.
 *
 * Return value of every int function: 0 on success, otherwise a hipError_t /
 * ncclResult_t value (>0) or -1 for argument errors; pllhip_last_error()
 * returns a thread-local description.
 */
#ifndef PLLHIP_H_
#define PLLHIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLLHIP_EXPORT __attribute__((visibility("default")))

typedef struct pllhip_ctx pllhip_ctx_t;

/* Geometry of one partition on one device (reference: the scalar fields of
 * pll_partition_t, pll.h:202-215, after pll_partition_create, pll.c:399). */
typedef struct pllhip_shape
{
  int device;                 /* HIP device ordinal                        */
  unsigned int states;        /* 4 = DNA, 20 = AA, anything >= 2 accepted  */
  unsigned int rate_cats;
  unsigned int sites;         /* sites held by THIS device (incl. asc-bias extra sites) */
  unsigned int tips;
  unsigned int clv_buffers;
  unsigned int rate_matrices;
  unsigned int prob_matrices;
  unsigned int scale_buffers;
  int pattern_tip;            /* tips are 1-byte codes, not CLVs           */
  int rate_scalers;           /* per-(site,rate) scalers instead of per-site */
  unsigned int asc_states;    /* how many of `sites` are the trailing one-per-state
                                 ascertainment-bias sites (pll.c:492-495); 0 = none */
} pllhip_shape_t;

/* same layout as pll_operation_t (pll.h:249-259) */
typedef struct pllhip_op
{
  unsigned int parent_clv;
  int parent_scaler;
  unsigned int child1_clv;
  unsigned int child1_matrix;
  int child1_scaler;
  unsigned int child2_clv;
  unsigned int child2_matrix;
  int child2_scaler;
} pllhip_op_t;

PLLHIP_EXPORT const char * pllhip_last_error(void);
PLLHIP_EXPORT int pllhip_device_count(int * count);

/* ---- context: owns every device buffer of one partition ---- */
PLLHIP_EXPORT int pllhip_ctx_create(const pllhip_shape_t * shape, pllhip_ctx_t ** out);
/* One partition over several devices of this process: the sites are split into contiguous
 * ranges (boundaries on multiples of 256 sites; the ascertainment-bias sites stay with the last
 * range), one ordinary context per entry of `devices` (an ordinal may repeat: two shards then
 * share a device -- how the sharding is tested on a one-GPU box).  The returned context is used
 * like any other: every call fans out to the shards, per-site arrays are split / gathered, P-matrices
 * and model are replicated, and lnL / derivative results are the host sum, in shard order, of
 * the per-shard values (deterministic; 8 doubles).  shape->device is ignored.  Not available
 * to such a context: site repeats, pllhip_comm_init, pllhip_dev_clv. */
PLLHIP_EXPORT int pllhip_ctx_create_sharded(const pllhip_shape_t * shape, const int * devices,
                                            unsigned int ndevices, pllhip_ctx_t ** out);
/* how many shards a context has (1 for an ordinary one) and the first site of shard i */
PLLHIP_EXPORT unsigned int pllhip_shard_count(pllhip_ctx_t * ctx);
PLLHIP_EXPORT unsigned int pllhip_shard_first_site(pllhip_ctx_t * ctx, unsigned int shard);
PLLHIP_EXPORT void pllhip_ctx_destroy(pllhip_ctx_t * ctx);
PLLHIP_EXPORT int pllhip_wait(pllhip_ctx_t * ctx);

/* ---- host -> device ---- */
/* encoded tip sequence, `sites` bytes (pll.c:825-883) */
PLLHIP_EXPORT int pllhip_put_tipchars(pllhip_ctx_t * ctx, unsigned int tip,
                                      const unsigned char * h_chars);
/* tipmap[code] = state bitmask, `maxstates` entries (pll.c:305-325) */
PLLHIP_EXPORT int pllhip_put_tipmap(pllhip_ctx_t * ctx, const unsigned int * h_tipmap,
                                    unsigned int maxstates);
/* a full CLV, sites*rate_cats*states doubles (pll.c:905-1045) */
PLLHIP_EXPORT int pllhip_put_clv(pllhip_ctx_t * ctx, unsigned int clv_index,
                                 const double * h_clv);
/* a tip CLV given as ONE states-vector per site (stride doubles apart),
 * replicated over the rate categories on the device (pll.c:1001-1020) */
PLLHIP_EXPORT int pllhip_put_tip_clv_persite(pllhip_ctx_t * ctx, unsigned int clv_index,
                                             const double * h_site_vectors,
                                             unsigned int stride);
PLLHIP_EXPORT int pllhip_put_pattern_weights(pllhip_ctx_t * ctx, const unsigned int * h_w);
/* NULL clears the invariant-site index array (models.c:558-647) */
PLLHIP_EXPORT int pllhip_put_invariant(pllhip_ctx_t * ctx, const int * h_invariant);
PLLHIP_EXPORT int pllhip_put_rates(pllhip_ctx_t * ctx, const double * h_rates,
                                   const double * h_rate_weights);
/* eigen system + frequencies + proportion of invariant sites of one rate
 * matrix; eigenvecs/inv_eigenvecs are states x states row-major
 * (models.c:251-331) */
PLLHIP_EXPORT int pllhip_put_model(pllhip_ctx_t * ctx, unsigned int params_index,
                                   const double * h_eigenvals,
                                   const double * h_eigenvecs,
                                   const double * h_inv_eigenvecs,
                                   const double * h_freqs,
                                   double prop_invar);

/* a P-matrix / a scale buffer given by the caller (the array-level pll_core_* entry points, whose
 * operands all come from the host) */
PLLHIP_EXPORT int pllhip_put_pmatrix(pllhip_ctx_t * ctx, unsigned int matrix_index, const double * h_pmatrix);
PLLHIP_EXPORT int pllhip_put_scaler(pllhip_ctx_t * ctx, unsigned int scaler_index, const unsigned int * h_scaler);

/* ---- device -> host ---- */
PLLHIP_EXPORT int pllhip_get_clv(pllhip_ctx_t * ctx, unsigned int clv_index, double * h_clv);
PLLHIP_EXPORT int pllhip_get_scaler(pllhip_ctx_t * ctx, unsigned int scaler_index,
                                    unsigned int * h_scaler);
PLLHIP_EXPORT int pllhip_get_pmatrix(pllhip_ctx_t * ctx, unsigned int matrix_index,
                                     double * h_pmatrix);
/* matrices first .. first + count - 1 in ONE copy (they are contiguous on the device and in the host mirror alike) */
PLLHIP_EXPORT int pllhip_get_pmatrices(pllhip_ctx_t * ctx, unsigned int first, unsigned int count, double * h_pmatrices);

/* Several CLVs / scale buffers to host memory in ONE launch and one wait (round 6: the host mirrors that small
 * partitions keep current by themselves, include/pll_amd.h).  h must come from pllhip_host_alloc (pinned, mapped into
 * the device's address space: the copy is a kernel that writes it over the bus, no staging); kind 0 = CLV `index`
 * (sites * rate_cats * states doubles), 1 = scale buffer `index`.  Not for sharded contexts and not for CLVs stored
 * by class (site repeats): those go through pllhip_get_clv / pllhip_get_scaler. */
typedef struct pllhip_mirror_job
{
  unsigned int kind, index;
  void * h;
} pllhip_mirror_job_t;
PLLHIP_EXPORT void * pllhip_host_alloc(size_t bytes);
PLLHIP_EXPORT void pllhip_host_free(void * p);
PLLHIP_EXPORT int pllhip_mirror_batch(pllhip_ctx_t * ctx, const pllhip_mirror_job_t * h_jobs, unsigned int count);

/* ---- compute ---- */

/* replaces pll_core_update_pmatrix (core_pmatrix.c:24; AVX2-flag kernels
 * core_pmatrix_avx.c:42 for 4 states, core_pmatrix_avx2.c:37 for 20).
 * params_indices has rate_cats entries; the other arrays `count`. */
PLLHIP_EXPORT int pllhip_update_pmatrices(pllhip_ctx_t * ctx,
                                          const unsigned int * h_params_indices,
                                          const unsigned int * h_matrix_indices,
                                          const double * h_branch_lengths,
                                          unsigned int count);

/* replaces the op loop of pll_update_partials (partials.c:177) and the three
 * kernels pll_core_update_partial_{tt,ti,ii} (core_partials.c:82,354,510;
 * AVX2-flag kernels core_partials_avx.c:366,581,899,1097 and
 * core_partials_avx2.c:568) including pll_core_create_lookup
 * (core_partials.c:725).  Ops are executed in list order. */
PLLHIP_EXPORT int pllhip_update_partials(pllhip_ctx_t * ctx, const pllhip_op_t * h_ops,
                                         unsigned int count);

/* The op-list planner of the 4-state whole-list kernel (partials_fused.hip) on its own -- host
 * logic, no device needed: the order it gives the list (order_out[pos] = position in ops), how
 * many inner operands have to be copied back from HBM with `nslots` slots per wave (values
 * that gave their slot up, operands written by earlier calls), and, if slots_out is not NULL,
 * 6 ints per op in the new order: left / right / parent slot, the slots the inherited counts
 * are read from, and flags (bit 0 / 1: the left / right operand is copied into its slot at the
 * top of the op before).  Returns 0, 1 if the kernel does not take the list's shape (the
 * per-level launches run it then), -1 on bad indices. */
PLLHIP_EXPORT int pllhip_fused_plan_dry(unsigned int tips, unsigned int clv_buffers,
                                        unsigned int scale_buffers, int pattern_tip,
                                        const pllhip_op_t * ops, unsigned int count,
                                        unsigned int nslots, unsigned int * order_out,
                                        unsigned int * reloads_out, int * slots_out);

/* Host logic, no device (round 5): the op list as independent sub-lists ("segments": ops that share no buffer any of
 * them writes -- the two sides of the root edge of a full traversal), which the whole-list kernels hand out as
 * (tile of sites, segment) work items when the tiles alone do not fill the chip (partials_fused.hpp).  seg_out[i] =
 * segment of op i, segment 0 the longest, every segment at least two ops; returns the number of segments (1: the list
 * does not split; at most min(max_segments, 8)). */
PLLHIP_EXPORT unsigned int pllhip_fused_segments_dry(unsigned int tips, unsigned int clv_buffers,
                                                     unsigned int scale_buffers, int pattern_tip,
                                                     const pllhip_op_t * ops, unsigned int count,
                                                     unsigned int max_segments, unsigned int * seg_out);

/* Environment switches (round 5).  1: `name` is one of the switches a client may set (INTEGRATION.md section 6 lists
 * them), read as it stands; 0: a developer's knob -- tile shapes, grid caps, experiments -- which the library reads
 * only under PLLHIP_DEVELOPER=1, so that a stray variable cannot move a production run off the tested configuration. */
PLLHIP_EXPORT int pllhip_env_is_user_switch(const char * name);
/* 1: `name` is set in the environment and the library would act on it right now; 0: unset, or ignored. */
PLLHIP_EXPORT int pllhip_env_is_honoured(const char * name);
/* PLLHIP_DEVELOPER is read once per process; this reads it again (for tests that change it). */
PLLHIP_EXPORT void pllhip_env_reload(void);

/* Host logic, no device: which path a partition below 16,384 sites takes for this op list (4 or 20 states) -- 1: the
 * whole-list kernel, 0: the per-level launches, -1: an index out of range -- and, if the pointers are not NULL, the
 * two estimated times in microseconds (partials.hip: a launch per dependency level and op kind plus the bytes, against
 * a fixed preparation plus a time per op; constants measured on one MI355X, profiles/r4_small_partitions_ab.txt). */
PLLHIP_EXPORT int pllhip_small_partition_estimate(unsigned int states, unsigned int sites, unsigned int tips,
                                                  unsigned int clv_buffers, int pattern_tip, const pllhip_op_t * ops,
                                                  unsigned int count, double * whole_us_out, double * level_us_out);

/* The scaling certificate (20 states; libpll_amd/csrc/hip/ctx.hpp).  The whole-list kernel runs
 * the mat-vec of tip-inner ops on the matrix cores (fused multiply-adds; the reference rounds products and sums
 * separately, core_partials_avx.c:1229-1284), so those CLVs agree with the reference's to ~1e-15 per op -- and every
 * scaling decision taken on such a value is certified: a largest entry within a window of 2^-256 far wider than the
 * accumulated difference raises a flag, and the list is run again in the reference's order before anything reads
 * its results.  out4 = {op lists launched with the test, flags raised, lists run again, uncertified}: while
 * out4[3] == 0 every scaler count of the partition is the reference's (core_partials_avx2.c:752-800).
 * PLLHIP_AA_TI_MFMA=0: reference order everywhere, every CLV bit for bit, nothing to certify. */
PLLHIP_EXPORT int pllhip_cert_stats(pllhip_ctx_t * ctx, unsigned long long * out4);

/* Host logic of the same planner, no device: where the 4-state whole-list kernel keeps each op's tip characters.
 * tips[i]: bit 0 / 1 = op i (in the PLANNED order) has a left / right tip row.  chars_out[i]: bits 0-7 / 8-15 the
 * first lane of the left / right row in the wave's character registers, bit 16 / 17 = has a left / right tip;
 * batch_out[i]: the batch of 64 x 16 bytes the op's rows are fetched with.  Returns the number of batches. */
PLLHIP_EXPORT unsigned int pllhip_fused_char_batches_dry(const unsigned int * tips, unsigned int count,
                                                         unsigned int rate_cats, unsigned int * chars_out,
                                                         unsigned int * batch_out);

/* replaces pll_core_update_partial_tt proper (core_partials.c:82-200): the parent CLV of a tip-tip
 * node is, per site, row ((code1 << log2_maxstates) + code2) of the lookup table the caller built
 * with pll_core_create_lookup -- a gather on the device from the uploaded table (h_lookup:
 * `rows` rows of rate_cats * states doubles) by the two tips' characters already in the
 * context; the parent's scale buffer (if >= 0) is cleared. */
PLLHIP_EXPORT int pllhip_partial_tt_from_lookup(pllhip_ctx_t * ctx, unsigned int parent_clv, int parent_scaler,
                                                unsigned int tip1, unsigned int tip2, const double * h_lookup,
                                                size_t rows, unsigned int log2_maxstates);

/* replaces pll_core_edge_loglikelihood_ii / _ti / _ti_4x4
 * (core_likelihood.c:726,412,211).  A clv index < tips in pattern-tip mode
 * selects the tip-inner kernel.  h_persite_lnl may be NULL.  The returned
 * value is the sum over this device's sites (and, once pllhip_comm_init has
 * been called, over all ranks). */
PLLHIP_EXPORT int pllhip_edge_loglikelihood(pllhip_ctx_t * ctx,
                                            unsigned int parent_clv, int parent_scaler,
                                            unsigned int child_clv, int child_scaler,
                                            unsigned int matrix_index,
                                            const unsigned int * h_freqs_indices,
                                            double * h_persite_lnl,
                                            double * h_lnl);

/* replaces pll_core_root_loglikelihood (core_likelihood.c:25) */
PLLHIP_EXPORT int pllhip_root_loglikelihood(pllhip_ctx_t * ctx,
                                            unsigned int clv_index, int scaler_index,
                                            const unsigned int * h_freqs_indices,
                                            double * h_persite_lnl,
                                            double * h_lnl);

/* replaces pll_core_update_sumtable_ii / _ti (core_derivatives.c:125,277).
 * The table stays on the device in slot `slot` (0..PLLHIP_SUMTABLE_MAX_SLOTS-1; a slot's
 * buffer -- one CLV's size -- is allocated when it is first used and released by
 * pllhip_release_sumtable or with the context).  pllhip_sumtable_budget: how many slots
 * the host layer should keep alive at most (a byte budget, env PLL_AMD_SUMTABLE_SLOTS). */
#define PLLHIP_SUMTABLE_MAX_SLOTS 256
PLLHIP_EXPORT unsigned int pllhip_sumtable_budget(pllhip_ctx_t * ctx);
PLLHIP_EXPORT int pllhip_release_sumtable(pllhip_ctx_t * ctx, unsigned int slot);
PLLHIP_EXPORT int pllhip_update_sumtable(pllhip_ctx_t * ctx,
                                         unsigned int parent_clv, int parent_scaler,
                                         unsigned int child_clv, int child_scaler,
                                         const unsigned int * h_params_indices,
                                         unsigned int slot);
PLLHIP_EXPORT int pllhip_put_sumtable(pllhip_ctx_t * ctx, unsigned int slot,
                                      const double * h_sumtable);
PLLHIP_EXPORT int pllhip_get_sumtable(pllhip_ctx_t * ctx, unsigned int slot,
                                      double * h_sumtable);

/* replaces the site loop of pll_core_likelihood_derivatives
 * (core_derivatives.c:501; AVX2 kernel core_derivatives_avx2.c:523).
 * h_diagptable: rate_cats*states*4 doubles built by the host exactly as
 * core_derivatives.c:560-575 does.  Outputs are derivatives of -lnL. */
PLLHIP_EXPORT int pllhip_likelihood_derivatives(pllhip_ctx_t * ctx, unsigned int slot,
                                                int parent_scaler, int child_scaler,
                                                const unsigned int * h_params_indices,
                                                const double * h_diagptable,
                                                double * h_d_f, double * h_dd_f);

/* Ascertainment-bias correction (likelihood.c:24-119,170-247,321-414;
 * core_derivatives.c:654-727): which of PLL_ATTRIB_AB_LEWIS / _FELSENSTEIN /
 * _STAMATAKIS (the attribute bits, 0 = off) the log-likelihood and derivative
 * calls apply over the context's asc_states trailing sites, and the sum of the
 * pattern weights of the ordinary sites (pll_partition_t::pattern_weight_sum). */
PLLHIP_EXPORT int pllhip_set_asc(pllhip_ctx_t * ctx, int asc_type, unsigned int pattern_weight_sum);

/* Site repeats (host/repeats.c, hip/repeats.hip; no counterpart in the reference
 * snapshot).  Finds the classes of CLV slot `parent` when it is computed from `child1`
 * and `child2`: sites whose rows at both children agree share a row.  If there are at
 * most max_classes of them the slot is from now on stored by class -- that many rows,
 * also in the scale buffer written together with it -- and *classes says how many;
 * otherwise (or when an inner child is itself stored per site) *classes = 0 and the
 * slot is stored per site.  pllhip_update_partials, the log-likelihood calls and
 * pllhip_update_sumtable follow the maps; pllhip_get_clv returns the rows as stored,
 * pllhip_get_site_id the site -> row map to expand them with. */
PLLHIP_EXPORT int pllhip_identify_repeats(pllhip_ctx_t * ctx, unsigned int parent,
                                          unsigned int child1, unsigned int child2,
                                          unsigned int max_classes, unsigned int * classes);
PLLHIP_EXPORT int pllhip_get_site_id(pllhip_ctx_t * ctx, unsigned int clv_index,
                                     unsigned int * h_site_id);
/* rows CLV slot clv_index is stored in (0: one per site).  A context sharded over several devices identifies per shard
 * and reports *classes = 0 to the caller (its mirrors arrive expanded); this is the sum over its shards. */
PLLHIP_EXPORT unsigned int pllhip_repeats_rows(pllhip_ctx_t * ctx, unsigned int clv_index);

/* ---- multi-GPU: one process per GPU, RCCL sum of the scalar results ---- */
PLLHIP_EXPORT int pllhip_comm_unique_id(void * id128);
/* which RCCL the process uses: the file the collective symbols were bound to -- the copy already
 * mapped by the host program (e.g. PyTorch's) if there is one, else the system's; "" before the
 * first pllhip_comm_* call */
PLLHIP_EXPORT const char * pllhip_rccl_path(void);
PLLHIP_EXPORT int pllhip_comm_init(pllhip_ctx_t * ctx, int rank, int nranks,
                                   const void * id128);
/* all-reduces this context has entered so far: the ranks of a job must agree on it at every point (a check for
 * clients and tests; a rank whose scaling certificate trips looks at its flag BEFORE the evaluation, likelihood.hip) */
PLLHIP_EXPORT unsigned long long pllhip_comm_reduces(pllhip_ctx_t * ctx);

/* ---- HIP-event stopwatch on the context's stream ---- */
PLLHIP_EXPORT int pllhip_timer_start(pllhip_ctx_t * ctx);
PLLHIP_EXPORT int pllhip_timer_stop_ms(pllhip_ctx_t * ctx, float * ms);
/* the last stop's time on every shard's own stream (returns the number of shards; fills at most `cap`) */
PLLHIP_EXPORT unsigned int pllhip_timer_shard_ms(pllhip_ctx_t * ctx, float * ms, unsigned int cap);

/* Per-launch kernel timing for bench.py's roofline figure: while enabled, every
 * launch of the hot kernels is bracketed by a HIP event pair on the context's
 * stream.  pllhip_profile_read drains the stream and returns, per kernel class,
 * the number of launches and their summed duration since the last reset. */
#define PLLHIP_PROF_PARTIALS_II 0
#define PLLHIP_PROF_PARTIALS_TI 1
#define PLLHIP_PROF_PARTIALS_TT 2
#define PLLHIP_PROF_LNL 3
#define PLLHIP_PROF_SUMTABLE 4
#define PLLHIP_PROF_DERIVATIVES 5
#define PLLHIP_PROF_PMATRIX 6
#define PLLHIP_PROF_KINDS 7
PLLHIP_EXPORT int pllhip_profile_enable(pllhip_ctx_t * ctx, int on);
PLLHIP_EXPORT int pllhip_profile_read(pllhip_ctx_t * ctx, unsigned int * launches /*[KINDS]*/,
                                      double * total_ms /*[KINDS]*/);

/* Measurement only (bench.py: roofline.box_ceiling): NOTHING BUT the stores of an op list -- every parent CLV and scale
 * buffer of `ops`, a wave's tile of each in turn, the tile walk and cache policy of the whole-list kernels, no loads, no
 * arithmetic -- `reps` times behind one untimed pass, timed with HIP events on the context's stream: what a write
 * stream with the list kernel's own addresses reaches on THIS device now.  OVERWRITES those CLVs and scale buffers:
 * run the list again before reading anything.  Not for sharded contexts. */
PLLHIP_EXPORT int pllhip_write_ceiling(pllhip_ctx_t * ctx, const pllhip_op_t * h_ops, unsigned int count,
                                       unsigned int reps, float * ms_per_pass, double * bytes_per_pass);
/* Where the CLV arena lies (ctx.hip "Where an arena lies"): the speed of a partition's stores depends on where in
 * device memory it was placed (5.7-7.3 TB/s for the same list on one device), so an arena of 384 MB or more is
 * allocated up to PLLHIP_PLACEMENT_TRIES times (environment, default 12; 1: take the first), each place written with the
 * list kernels' own store pattern with the clock running, the first fast one or else the fastest kept (and zeroed),
 * the others freed -- never with less than another arena's worth + 4 GB of
 * device memory left free.  Returns the number of places tried (0: no search), gbs[i] = GB/s of those stores
 * on place i (at most `cap`), *kept = the one the partition lives in.  A sharded context: its first shard's. */
PLLHIP_EXPORT int pllhip_placement_info(pllhip_ctx_t * ctx, double * gbs, unsigned int cap, int * kept);
/* Measurement only: the CLV arena zeroed once more by the kernel that zeroes it at creation, timed -- GB/s of a
 * contiguous non-temporal write stream over the partition's own memory (where an arena lies decides how fast it can be
 * written, ctx.hip "Where an arena lies").  OVERWRITES every CLV.  Not for sharded contexts. */
PLLHIP_EXPORT int pllhip_arena_fill_bandwidth(pllhip_ctx_t * ctx, double * gbs);
/* What the last op list planned by the 20-state whole-list kernel is made of: {ops, tip-tip ops ahead of the list,
 * tip-tip ops in the list (one gather each), table lookups, inner-inner ops on the matrix cores, tip-inner ops on
 * the matrix cores, tip-inner ops on the vector unit, operands reloaded from HBM}; zeros if none was planned. */
PLLHIP_EXPORT int pllhip_aa_list_kinds(pllhip_ctx_t * ctx, unsigned int * out8);

/* raw device pointer of a CLV (for tools that share HBM buffers, e.g. a
 * torch tensor wrapped around it); NULL if out of range */
PLLHIP_EXPORT void * pllhip_dev_clv(pllhip_ctx_t * ctx, unsigned int clv_index);

#ifdef __cplusplus
}
#endif
#endif /* PLLHIP_H_ */
