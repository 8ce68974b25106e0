// ctx.hpp -- internal: device context shared by the .hip translation units.
// Not installed; the public surface is include/pllhip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#include "pllhip.h"

#define PLLHIP_MAX_RATE_CATS 64
#define PLLHIP_REDUCE_BLOCKS 65536 /* upper bound on blocks of a reducing kernel */
#define PLLHIP_HOSTSUM_MAX 8192    /* workgroup sums (x components) the host adds itself; larger grids: k_final_sum */

struct ncclComm;

struct pllhip_rep_work;

struct pllhip_ctx
{
  pllhip_shape_t sh;
  hipStream_t stream = nullptr;

  // A context made by pllhip_ctx_create_sharded owns no device memory itself: it is a GROUP of
  // ordinary contexts, one per device, each holding a contiguous range of the sites
  // (shard i: [shard_lo[i], shard_lo[i + 1]) of the group's sites; boundaries on multiples of
  // 256 sites).  Every entry point of pllhip.h called on a group fans out (shard.hip;
  // PLLHIP_ALL_SHARDS below); results that are sums over sites are added on the host in shard
  // order.  `defer`: set on a shard while its group collects -- the result-returning calls then
  // enqueue everything and return without waiting (pllhip_result_wait fetches).
  std::vector<pllhip_ctx *> shards;
  std::vector<size_t> shard_lo;
  bool shard_threads = true, shard_poll = true; // env PLLHIP_SHARD_THREADS / PLLHIP_SHARD_POLL = 0 (read at creation)
  bool shard_pin = false; // a shard's thread runs on the cores of its device's NUMA node (more than one distinct device; env PLLHIP_SHARD_PIN=0: never)
  struct pllhip_shard_pool * pool = nullptr; // (round 5) the group's enqueueing threads, one per shard but the first (shard.hip)
  bool defer = false;
  // (round 5) what the group needs to wait for a shard's enqueued result the way an unsharded context waits for its
  // own -- polling the word / the workgroup sums in host-mapped memory instead of one hipStreamSynchronize per
  // shard after the other: the call's sequence number, whether the host adds the workgroup
  // sums, whether a word is published, whether copies follow the kernel (then the stream is waited for)
  unsigned long long pending_seq = 0;
  bool pending_hostsum = false, pending_spin = false, pending_stream_work = false;

  size_t span = 0;         // states * rate_cats doubles per site
  size_t clv_elems = 0;    // sites * span
  size_t scaler_elems = 0; // sites (per-site mode) or sites * rate_cats
  // distance between two CLVs / scale buffers in their arenas: the elements plus
  // PLLHIP_TAIL_SITES sites of slack EACH, so that a kernel may also STORE a whole last tile
  size_t clv_stride = 0, scaler_stride = 0;
  size_t tip_stride = 0;   // bytes between two tips' code rows
  size_t pmat_elems = 0;   // rate_cats * states * states

  // HBM-resident partition data
  double * clv_arena = nullptr;        // all CLVs back to back
  std::vector<double *> clv;           // [tips + clv_buffers], nullptr for pattern tips
  unsigned char * tipchars = nullptr;  // [tips][tip_stride]
  unsigned int * scaler_arena = nullptr;
  double * pmatrix = nullptr;          // [prob_matrices][rate_cats][states][states]
  // model, per rate matrix
  double * eigenvals = nullptr;        // [rate_matrices][states]
  double * eigenvecs = nullptr;        // [rate_matrices][states*states]
  double * inv_eigenvecs = nullptr;
  double * freqs = nullptr;            // [rate_matrices][states]
  double * prop_invar = nullptr;       // [rate_matrices]
  double * rates = nullptr;            // [rate_cats]
  double * rate_weights = nullptr;     // [rate_cats]
  unsigned int * pattern_weights = nullptr; // [sites]
  int * invariant = nullptr;           // [sites] or nullptr
  bool any_prop_invar = false;
  std::vector<double> h_prop_invar;
  unsigned int * tipmap = nullptr;     // [256]
  unsigned int maxstates = 0;
  double * sumtable[PLLHIP_SUMTABLE_MAX_SLOTS] = {}; // allocated on first use
  double * lnl_scratch = nullptr; // CLV-sized: per-state lnL terms of the two-pass kernels (likelihood.hip)
  // whole-list kernel (partials_fused.hip): the plan's device copy and two pinned staging buffers
  void * d_plan = nullptr;
  void * d_sink = nullptr; // 1 KB that the stores of lanes past the last site go to
  unsigned int tile_counter_phase = 0;     // (4 states: which of the two sets of tile counters the next launch hands out from)
  unsigned int * d_tile_counter = nullptr; // whole-list kernels: next tile to hand out (PLLHIP_TILE_COUNTER_BYTES: one per
                                           // group of eight workgroups, 128 bytes apart -- partials_fused.hip)
  // the op list of the last whole-list launch (its plan is still on the device: an identical
  // list -- the usual case while branch lengths or model parameters are optimised -- is
  // launched again without planning or upload)
  std::vector<pllhip_op_t> fused_last_ops;
  struct pllhip_level_cache * level_cache = nullptr; // the same for the per-level path (partials.hip)
  unsigned int fused_last_jobs = 0, fused_last_count = 0, fused_last_nslots = 0;
  size_t fused_last_jobs_offset = 0; // pair-table jobs within d_plan
  size_t fused_last_rowtab_offset = 0;     // ... and of the tip-character row table (partials_fused.hip)
  size_t fused_last_segs_offset = 0, fused_last_srcs_offset = 0; // ... the segment table, the reload sources
  unsigned int fused_last_nsegs = 1, fused_last_longest = 0;      // segments of the kept plan; ops of its longest
  float last_timer_ms = 0.f;               // pllhip_timer_stop_ms's last result on this context's stream
  unsigned char * fused_zero_row = nullptr; // [sites + slack] zeros: the "tip" of an op without one
  int fused_last_mode = 0;
  // Kept plans (the whole-list kernel's records, the per-level path's arguments) hold device
  // addresses and indices.  Whatever reallocates a buffer such a plan may reference bumps
  // layout_epoch; a kept plan is only reused while its epoch is the current one.
  unsigned int layout_epoch = 0, fused_last_epoch = 0;
  double * d_pairtab = nullptr; // pair tables of the tip-tip ops of the current op list
  // 20 states: scratch of the lookup ops (partials_aa_mfma.hip, k_aa_cherry_rounds)
  double * cherry_pool = nullptr;
  unsigned char * cherry_codes = nullptr;
  unsigned int cherry_ms = 0; // maxstates the scratch was sized for
  unsigned char * cherry_zero = nullptr;     // [sites] zero characters (the absent second tip of a tip-inner lookup op)
  double * cherry_pool_all = nullptr;        // the tables of ALL lookup ops of a list (partials_aa_fused.hip)
  unsigned int cherry_pool_all_ops = 0;      // lookup ops it has room for
  bool cherry_pool_failed = false;           // its allocation failed once: no lookup ops on this context any more
  // a shard of a group, root lnL with per-rate scale buffers: the entries the reference would read (see
  // pllhip_group_root_loglikelihood, shard.hip), and the switch that makes pllhip_root_loglikelihood use them
  unsigned int * root_counts = nullptr;
  const unsigned int * root_scaler_override = nullptr;
  // 20 states, chunk launches (partials_aa_mfma.hip): per-site verdicts of an op's earlier chunks when the op scales in place
  unsigned int * split_verdicts = nullptr;
  unsigned int split_verdicts_ops = 0;        // ops of a launch it has room for
  struct pllhip_aa_fused_cache * aa_fused = nullptr; // 20-state whole-list kernel: its kept plan
  size_t pairtab_elems = 0;
  void * h_plan[2] = {nullptr, nullptr};
  hipEvent_t plan_done[2] = {nullptr, nullptr};
  bool plan_pending[2] = {false, false};
  size_t plan_cap = 0;
  int plan_next = 0;
  int fused_debug = 0;   // env PLLHIP_FUSED_DEBUG (read when the context is created, like every other client's switch:
                         // the hot path looked it up in the environment several times per call -- ADVICE r5)
  bool no_fused = false; // env PLLHIP_FUSED=0: one launch per dependency level instead
  bool force_fused = false; // env PLLHIP_FUSED=2: also for partitions too small for it to pay (tests)

  // reductions: per-block partial sums, then a fixed-order final pass
  double * block_partials = nullptr;   // [PLLHIP_REDUCE_BLOCKS][2]
  double * d_result = nullptr;         // [4]
  unsigned int * d_zero = nullptr;     // [4] zeros
  double * d_tiptab = nullptr;         // 20 states: [2][maxstates][rate_cats][20] tip row sums of the current op
  size_t tiptab_elems = 0;
  double * h_result = nullptr;         // pinned, host-mapped [4]: [0..2] results, [3] the sequence word the host spins on
  unsigned long long result_seq = 0;   // number of the last result-returning launch
  bool no_spin = false;                // env PLLHIP_SPIN=0: wait for the stream instead (A/B measurements)
  bool no_hostsum = false;             // env PLLHIP_HOSTSUM=0: k_final_sum instead of the host's sum of workgroup sums
  int fuse_forced = -1;                // env PLLHIP_FUSE_REDUCE=0/1: the final sum never / always inside the reducing kernel
  unsigned int fuse_max_grid = 128;    // env PLLHIP_FUSE_MAX_GRID (all read when the context is created)
  double * h_result_dev = nullptr;     // device address of h_result
  // workgroup sums of a result-returning kernel, written by the kernel straight into host memory and added by
  // the host (likelihood.hip: pllhip_result_wait_host): {value, sequence number} per workgroup and component
  double2 * h_partials = nullptr, * h_partials_dev = nullptr;
  unsigned int hostsum_grid = 0, hostsum_ncomp = 0, hostsum_width = 256; // the call in flight (width: threads of the sum it stands for)
  unsigned int * d_counter = nullptr;  // arrival counter of the reducing kernels
  double * d_persite = nullptr;        // [sites], lazily allocated

  // staging for small per-call parameter arrays
  void * h_stage = nullptr;            // pinned
  void * d_stage = nullptr;
  size_t stage_bytes = 0;

  ncclComm * comm = nullptr;
  int nranks = 1;
  unsigned long long comm_reduces = 0; // all-reduces entered (pllhip_comm_reduces: every rank must count alike)

  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int num_cus = 256;
  // Grid cap of the streaming kernels (env PLLHIP_BLOCKS_PER_CU).  Measured on
  // MI355X, 1,000,000 sites: one pass over the data with as many workgroups as
  // it takes (3907) runs the 4-state CLV kernel in 65 us; capping the grid so
  // that every wave makes 2 or 3 grid-stride passes costs 70-72 us.  So the cap
  // is set where it only matters for alignments beyond ~16 M sites per GPU.
  int blocks_per_cu = 256;
  int asc_type = 0;                      // PLL_ATTRIB_AB_* bits (0 = no correction)
  unsigned int asc_weight_sum = 0;
  double * d_asc = nullptr;              // [3] correction terms added by the final sum
  const double * pending_extra = nullptr; // consumed by the next pllhip_reduce_out
  // site repeats (host/repeats.c): per CLV slot, rows stored (0 = one per site), the
  // site -> row map, and per row the rows of the two children it is computed from
  struct node_rows
  {
    unsigned int classes = 0;
    unsigned int * site_id = nullptr; // [sites + slack]
    unsigned int * perm = nullptr;    // [sites]: the sites in class order (what a parent's identification starts from)
    unsigned int * perm_class = nullptr; // [sites]: the class of perm[i] (read in order instead of gathered from site_id)
    unsigned int * lrow = nullptr;    // [classes + slack]
    unsigned int * rrow = nullptr;
    size_t row_cap = 0;
  };
  std::vector<node_rows> rows;          // empty unless repeats were ever identified
  std::vector<int> scaler_owner;        // per scale buffer: the CLV slot it was last written with (a buffer written
                                        // with a CLV stored by class is stored by class too); kept once `rows` exists
  struct pllhip_rep_work * rep_work = nullptr; // sort / scan buffers of repeats.hip
  size_t clv_arena_bytes = 0;            // all CLVs of the partition
  size_t clv_arena_alloc_bytes = 0; // what was allocated for the arena (strides included)
  int placement_tries = 0, placement_best = 0; // where the CLV arena lies: places tried, the one kept (ctx.hip)
  std::vector<double> placement_gbs;          // GB/s of a zeroing pass over each place tried
  bool no_batch = false;                 // PLLHIP_NO_BATCH=1: one launch per op (measurements)
  int nt_override = -1;                  // PLLHIP_NT=0/1 forces the cache policy (measurements); 2: and the whole-list
                                         // kernel's count stores, which follow the partition's size otherwise
  // 20 states: 1 = bit-exact vector kernels only (env PLLHIP_AA_EXACT=1);
  // 0 = matrix-core kernels where they exist (last-bit differences, see
  // partials_aa_mfma.hip)
  int aa_exact = 0;
  // 20-state whole-list kernel: the ONE mat-vec of a tip-inner op on the matrix cores (fused chains) instead of the
  // vector unit in the reference's non-fused order -- +6 % on a 200-taxon random tree, +42 % on a ladder.  Round 5:
  // opt-in, because tip-inner CLVs (and what is computed from them) then agree with the reference's to ~1e-15 per op
  // instead of bit for bit, and a scaler count could differ when an entry lies within that of 2^-256.  Round 6: the
  // DEFAULT (env PLLHIP_AA_TI_MFMA=0: off), behind the scaling certificate below, which keeps the counts the reference's.
  bool aa_ti_mfma = true;
  // ---- the scaling certificate (round 6; partials.hip: pllhip_cert_resolve)
  // clv_err[i]: a bound on the relative difference, entry by entry, between CLV i and the reference's (0: bit for
  // bit; > 0: written by a matrix-core tip-inner op, or computed from such a CLV -- PLLHIP_CERT_OP_ERR per op plus
  // the operands' bounds); n_inexact: how many are > 0 (0: nothing below costs anything).  An op that scales such a
  // value also tests whether its largest entry lies within a window of the threshold eight times wider than the
  // bound: outside it, the decision -- hence the count -- is the reference's.  Inside it, the kernel raises *h_cert
  // (host memory); the host looks at the word before anything else reads or changes what the list read or wrote:
  // cert_kind 1 (the list ran tip-inner ops on the matrix cores) -- the list (cert_ops) is run again in the
  // reference's order; cert_kind 2 (reference order on marked operands, nothing to run again), a second trip of a
  // re-run whose operands were marked, and a bound too large for any window count as `uncertified`.
  std::vector<double> clv_err;
  unsigned int n_inexact = 0;
  unsigned int * h_cert = nullptr, * h_cert_dev = nullptr;
  bool cert_pending = false, cert_force_exact = false;
  int cert_kind = 0;
  std::vector<pllhip_op_t> cert_ops;
  unsigned long long cert_stats[4] = {0, 0, 0, 0}; // lists launched with the test, trips, re-runs, uncertified

  // optional per-launch timing (pllhip_profile_*): one event pair per launch
  bool profiling = false;
  std::vector<hipEvent_t> prof_events;   // pairs
  std::vector<int> prof_kind;            // PLLHIP_PROF_* per pair
  size_t prof_used = 0;                  // pairs in use
};

// RAII helper: brackets one kernel launch with events when profiling is on
struct pllhip_prof_scope
{
  pllhip_ctx * c;
  size_t slot;
  bool on;
  pllhip_prof_scope(pllhip_ctx * ctx, int kind);
  void stop();
  ~pllhip_prof_scope() { stop(); }
};

void pllhip_set_error(const char * fmt, ...);

// Environment switches.  The ones a client may rely on are the table in ctx.hip (pllhip_user_switches; INTEGRATION.md
// section 6 documents each one); every other PLLHIP_* variable is a developer's knob -- tile shapes, grid caps,
// experiments whose A/B results are in DESIGN.md -- and is read only under PLLHIP_DEVELOPER=1: a stray variable cannot
// move a production run off the configuration the test suite covers, and the library says so once on stderr when it
// ignores one.  (forty switches, each a configuration nobody crosses with the others.)
const char * pllhip_env(const char * name);

// Run `expr` on every shard of a group context and return (inside `expr`: s = the shard, lo =
// its first site within the group); falls through for an ordinary context.
// (A call on a group visits every device of the group; the caller's current device is put back
// when it returns -- a client that works with devices of its own must not find it changed.)
struct pllhip_device_guard
{
  int prev = -1;
  pllhip_device_guard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
  ~pllhip_device_guard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define PLLHIP_ALL_SHARDS(c, expr)                              \
  do {                                                          \
    if (!(c)->shards.empty()) {                                 \
      pllhip_device_guard guard_;                               \
      for (size_t si_ = 0; si_ < (c)->shards.size(); ++si_) {   \
        pllhip_ctx * s = (c)->shards[si_];                      \
        const size_t lo = (c)->shard_lo[si_];                   \
        (void)lo;                                               \
        const int rc_ = (expr);                                 \
        if (rc_) return rc_;                                    \
      }                                                         \
      return 0;                                                 \
    }                                                           \
  } while (0)

// (round 5) The same on every shard AT ONCE, from one host thread per shard (shard.hip: pllhip_group_parallel): the
// calls of the hot path -- P-matrices, op lists, sumtables, the result-returning calls -- cost 5-15 us of launches
// per shard, and a host thread that visits eight devices in turn starts the last one 50-100 us after the first:
// half the run time of an eighth of BASELINE config 2 or 5.  Falls through for an ordinary context.
#include <functional>
int pllhip_group_parallel(pllhip_ctx * g, const std::function<int(pllhip_ctx *, size_t)> & fn);
#define PLLHIP_ALL_SHARDS_PAR(c, expr)                                                                          \
  do {                                                                                                          \
    if (!(c)->shards.empty())                                                                                   \
      return pllhip_group_parallel((c), [&](pllhip_ctx * s, size_t lo) -> int { (void)lo; return (expr); });   \
  } while (0)

// shard.hip: the group forms of the calls that return sums over sites
int pllhip_group_edge_loglikelihood(pllhip_ctx * c, unsigned int parent_clv, int parent_scaler,
                                    unsigned int child_clv, int child_scaler, unsigned int matrix_index,
                                    const unsigned int * h_freqs_indices, double * h_persite_lnl, double * h_lnl);
int pllhip_group_root_loglikelihood(pllhip_ctx * c, unsigned int clv_index, int scaler_index,
                                    const unsigned int * h_freqs_indices, double * h_persite_lnl, double * h_lnl);
int pllhip_group_likelihood_derivatives(pllhip_ctx * c, unsigned int slot, int parent_scaler, int child_scaler,
                                        const unsigned int * h_params_indices, const double * h_diagptable,
                                        double * h_d_f, double * h_dd_f);
void pllhip_group_destroy(pllhip_ctx * c);

#define HIP_TRY(expr)                                                         \
  do {                                                                        \
    hipError_t e_ = (expr);                                                   \
    if (e_ != hipSuccess) {                                                   \
      pllhip_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                       __FILE__, __LINE__);                                   \
      return (int)e_;                                                         \
    }                                                                         \
  } while (0)

static inline bool pllhip_is_tip(const pllhip_ctx * c, unsigned int clv_index)
{
  return c->sh.pattern_tip && clv_index < c->sh.tips;
}

static inline unsigned int * pllhip_scaler_ptr(const pllhip_ctx * c, int idx)
{
  return idx < 0 ? nullptr : c->scaler_arena + (size_t)idx * c->scaler_stride;
}

static inline const unsigned char * pllhip_tip_ptr(const pllhip_ctx * c, unsigned int tip)
{
  return c->tipchars + (size_t)tip * c->tip_stride;
}

static inline double * pllhip_pmat_ptr(const pllhip_ctx * c, unsigned int idx)
{
  return c->pmatrix + (size_t)idx * c->pmat_elems;
}

// Cache policy of the streaming kernels.  While all CLVs of the partition together fit
// the 256 MiB Infinity Cache, the default policy keeps a parent written by one op on die
// for the op that reads it; beyond that nothing survives until it is needed again and
// non-temporal loads and stores are faster.  Measured (4 states, 64 taxa, whole
// evaluation, G site-updates/s, default / non-temporal): 25 k sites (200 MB of CLVs)
// 15.0 / 13.7; 50 k (400 MB) 17.1 / 17.9; 100 k 17.3 / 18.8; 250 k 18.7 / 20.1.
#define PLLHIP_INFINITY_CACHE_BYTES ((size_t)256 << 20) /* MI355X: 256 MiB memory-side cache */
static inline bool pllhip_use_nt(const pllhip_ctx * c)
{
  if (c->nt_override >= 0) return c->nt_override != 0;
  return c->clv_arena_bytes >= PLLHIP_INFINITY_CACHE_BYTES;
}

// Grid size for a streaming kernel over `items` lanes-worth of work: all of it
// in one pass if that fits the cap (blocks_per_cu x CUs), otherwise the smallest
// whole number of grid-stride passes, with the grid shrunk so every block does
// the SAME number of passes -- a last pass that only part of the grid takes
// showed up as 7 % of the CLV kernel's time.
static inline unsigned int pllhip_stream_grid(const pllhip_ctx * c, size_t items,
                                              unsigned int block)
{
  size_t need = (items + block - 1) / block;
  const size_t cap = (size_t)c->num_cus * c->blocks_per_cu;
  if (need < 1) need = 1;
  if (need <= cap) return (unsigned int)need;
  const size_t passes = (need + cap - 1) / cap;
  return (unsigned int)((need + passes - 1) / passes);
}

// ---- shared between partials.hip and derivatives.hip ----
struct PartialsArgs
{
  double * __restrict__ parent;
  const double * __restrict__ left;      // inner child "1" (ii only)
  const double * __restrict__ right;     // inner child (ii: child 2, ti: the inner one)
  const unsigned char * __restrict__ ltip; // tip child (ti), tip child 1 (tt)
  const unsigned char * __restrict__ rtip; // tip child 2 (tt)
  const double * __restrict__ lmat;      // P of left / tip child  [R][S][S]
  const double * __restrict__ rmat;      // P of right child
  unsigned int * pscaler;                // may alias nothing else; nullptr = no scaling
  const unsigned int * lscaler;
  const unsigned int * rscaler;
  const unsigned int * __restrict__ tipmap;
  const unsigned int * zero;             // one device word holding 0 (stand-in for absent scalers)
  const double * ltab;                   // precomputed tip row sums [code][rate][state] (20 states)
  const double * rtab;
  unsigned int sites, rate_cats, states, maxstates;
  // site repeats: for each row of the parent, the row of the left / right child it is
  // computed from (a tip character when that child is a tip); nullptr = same index
  const unsigned int * lidx;
  const unsigned int * ridx;
  // 20 states, a launch over a CHUNK of the rate categories (partials_aa_mfma.hip, SPLIT): its first category;
  // rate_cats stays the CLV's count, lmat / rmat / ltab are the chunk's, lidx is the per-site verdict buffer
  unsigned int rate_first, pad_;
  // (round 6) the scaling certificate: not null = the op's scaling test also looks for a largest entry within
  // 2^-pad_ (relative) of the threshold and raises this word (host memory) -- k_aa_ii_mfma only
  unsigned int * cert;
};
static_assert(sizeof(PartialsArgs) * 24 <= 3900, "a batch of ops travels as kernel arguments (4 KB)");

// Several mutually independent ops (one tree level) run in ONE launch:
// blockIdx.y selects the op.  The op descriptors travel as kernel arguments
// (24 x 152 B < the 4 KB kernarg segment), so batching needs no staging copy.
// Every per-site device array carries this many sites of zeroed slack behind its last
// element, so that a wave working on the last (partial) round of 64 sites may load
// unconditionally and unclamped; what it computes there is masked out of every result.
#define PLLHIP_TAIL_SITES 64

// one turn of a host spin loop (the bounded polls of host-mapped result words)
#if defined(__x86_64__) || defined(__i386__)
#define PLLHIP_CPU_RELAX() __builtin_ia32_pause()
#elif defined(__aarch64__)
#define PLLHIP_CPU_RELAX() asm volatile("yield" ::: "memory")
#else
#define PLLHIP_CPU_RELAX() asm volatile("" ::: "memory")
#endif
#define PLLHIP_TILE_COUNTER_BYTES (256 * 128)

#define PLLHIP_BATCH_MAX 24
struct PartialsBatch
{
  PartialsArgs op[PLLHIP_BATCH_MAX];
};

enum { SCALE_NONE = 0, SCALE_SITE = 1, SCALE_RATE = 2 };

void pllhip_rep_work_free(pllhip_ctx * c); // repeats.hip
void pllhip_level_cache_free(pllhip_ctx * c); // partials.hip

// partials_gen_tile.hip: state counts other than 4 and 20
bool pllhip_gen_tile_covers(const pllhip_ctx * c);
int pllhip_launch_gen_batch(pllhip_ctx * c, const PartialsBatch & b, unsigned int count, int kind, int mode);

// kind: 0 = inner-inner, 1 = tip-inner (tip on the left), 2 = tip-tip
int pllhip_launch_partials(pllhip_ctx * c, const PartialsArgs & a, int kind, int mode,
                           int prof_kind);
int pllhip_allreduce_result(pllhip_ctx * c, unsigned int count);
// 20-state fast kernels (matrix cores / round-based tip-tip) for `count` mutually
// independent ops of one kind and mode; returns 1 if the case is not covered
bool pllhip_aa_fast_covers(const pllhip_ctx * c, int kind);
bool pllhip_aa_chunks_enabled(); // 20 states, rate_cats other than 1, 2, 4 on the matrix-core kernels (PLLHIP_AA_CHUNKS)
unsigned int pllhip_aa_lookup_budget(const pllhip_ctx * c);
int pllhip_launch_aa_batch(pllhip_ctx * c, PartialsBatch & b, unsigned int count, int kind, int mode);
bool pllhip_aa_cherry_covers(const pllhip_ctx * c, int mode);
bool pllhip_aa_cherry_pays(const pllhip_ctx * c, unsigned int lookups, unsigned int levels);
int pllhip_launch_aa_cherries(pllhip_ctx * c, const PartialsArgs * ops, const PartialsArgs * kid1,
                              const PartialsArgs * kid2, unsigned int count, int mode);
// the tables of a list's lookup ops, all at once: what each op's parent is the product of
struct AaLookupTables
{
  const double * tl, * tr;                    // [pair][rate][state]
  const unsigned char * t1, * t2, * t3, * t4; // characters: pair 1 = (t1, t2), pair 2 = (t3, t4)
};
// One table of a lookup op as a job for k_af_prepare (partials_aa_fused.hip; round 4: the tables of a whole list in
// the launch that prepares its matrices, instead of six launches of the tabulating kernels): row (c1 ms + c2) =
// pm x (tip table of kl [c1] (.) tip table of kr [c2]) in the order of the kernel the op itself would have run --
// mode 0: inner-inner (fused chains, core_partials_avx2.c:632-750), mode 1: tip-inner (products and sums rounded
// separately, core_partials_avx.c:1229-1284) --, or, mode 2, rows (c1 ms + 0) = the tip table of kl itself.
struct AaLookupJob
{
  const double * pm, * kl, * kr;
  double * dst;
  unsigned int mode, pad;
};
// jobs == nullptr: the tables are built here and now (launches of the tabulating kernels); otherwise only their
// places are assigned and jobs[2 i], jobs[2 i + 1] describe the two tables of op i
int pllhip_aa_lookup_tables(pllhip_ctx * c, const PartialsArgs * ops, const PartialsArgs * kid1,
                            const PartialsArgs * kid2, unsigned int count, AaLookupTables * out,
                            AaLookupJob * jobs = nullptr);
// partials_aa_fused.hip: a 20-state op list in one site-blocked launch; returns 1 if the list (or
// the partition) is not one it takes -- the caller then launches per level
int pllhip_aa_fused_update(pllhip_ctx * c, const pllhip_op_t * ops, unsigned int count);
// ceiling.hip: the list kernels' store pattern over a candidate place for the CLVs (n streams of stride_b bytes): GB/s;
// 1: no tile for this shape
int pllhip_probe_clv_streams(pllhip_ctx * c, char * base, size_t n, size_t stride_b, double * gbs);
// The scaling certificate (see pllhip_ctx::clv_err).  What one op adds to the bound: either summation order
// (core_partials_avx2.c:632-750 fused, core_partials_avx.c:1229-1284 not) rounds a sum of 20 non-negative products at
// most 8 times per term, the product of the two factors once: both results lie within 17 units of roundoff (2^-53) of
// the exact value, which moves with the operands by the sum of THEIR bounds -- 34 units per op, 40 taken.  The
// window of a list: 8 x the largest bound of its testing ops, at least 2^-44 (a 200-taxon tree: ~4e-12 -- one trip
// in a million evaluations); at most 2^-21 (the kernels' two-instruction pre-test looks at the high word only: 2^-20)
// -- a bound beyond that (only an op list that feeds one CLV to both sides of an op, again and again, doubles it
// that fast: never a tree) cannot be certified.
#define PLLHIP_CERT_OP_ERR (40.0 * 0x1p-53)
#define PLLHIP_CERT_WINDOW_MIN 0x1p-44
#define PLLHIP_CERT_WINDOW_MAX 0x1p-21
// Looks at the flag of the last certified list, if one is pending (waits for the stream unless `drained` says a
// result of a later launch has already arrived); *rerun: the list was run again -- results computed from its CLVs
// since are stale.  Called at the top of every entry point that reads or changes what a list read or wrote.
int pllhip_cert_resolve(pllhip_ctx * c, bool * rerun = nullptr, bool drained = false);
// The scaling certificate (ctx.hpp): an entry point that reads or changes what the last certified op list read or wrote
// looks at that list's flag first -- and runs the list again in the reference's order if it was raised.
#define PLLHIP_CERT_FIRST(c)                       \
  do {                                             \
    if ((c)->cert_pending) {                       \
      const int rc_cert_ = pllhip_cert_resolve(c); \
      if (rc_cert_) return rc_cert_;               \
    }                                              \
  } while (0)
// per-level path (partials.hip) and whole-list kernel (partials_aa_fused.hip): marks after the list, which ops test
void pllhip_cert_mark_clv(pllhip_ctx * c, unsigned int clv_index, double err);
static inline double pllhip_cert_err(const pllhip_ctx * c, unsigned int clv_index)
{
  return (c->n_inexact && clv_index < c->clv_err.size()) ? c->clv_err[clv_index] : 0.0;
}
// partials.hip: one op resolved into kernel arguments (kind 0 inner-inner, 1 tip-inner, 2 tip-tip)
int pllhip_resolve_op(pllhip_ctx * c, const pllhip_op_t & op, PartialsArgs & a, int & kind, int & mode);
void pllhip_aa_fused_free(pllhip_ctx * c);
