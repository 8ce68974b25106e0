// partials.hip -- Felsenstein pruning: parent CLV <- (P_l . left) (.) (P_r . right)
//
// Replaces pll_update_partials' op loop (partials.c:177) and the kernels
//   inner-inner  pll_core_update_partial_ii  core_partials.c:510
//                AVX2-flag path: 4 states core_partials_avx.c:366, 20 states core_partials_avx2.c:568
//   tip-inner    pll_core_update_partial_ti  core_partials.c:354
//                AVX2-flag path: core_partials_avx.c:899 (4), :1097 (20)
//   tip-tip      pll_core_create_lookup + pll_core_update_partial_tt
//                core_partials.c:725,82; AVX2-flag path core_partials_avx.c:262,146,581,531
//
// Kernels:
//   4 states, rate_cats 1/2/4/8/16   k_dna_partials: one lane per 16 bytes, rounds of 64 sites (below)
//   20 states, rate_cats 1/2/4       partials_aa_mfma.hip (matrix cores)
//   every other shape up to 64 states (and 20 states bit-exact, PLLHIP_AA_EXACT=1)
//                                    partials_gen_tile.hip
//   more than 64 states              k_gen_partials: one lane per site (below)
// CLVs are [site][rate][state] with state fastest.  Per-site scaling needs "all rate_cats x
// states entries < 2^-256": each lane tests its own entries, then one __ballot
// gives every lane the bits of the lanes that share its site.
//
// pllhip_update_partials (bottom of the file) schedules an op list by dependency
// level and launches one kernel per (level, kind, scaling mode).
//
// Roofline: pure HBM stream.  4 states: 396 B and 240 flop per site-update
// (0.6 flop/B); nothing but the child CLVs, the parent CLV and 12 B of
// scalers moves.  Tip lookups (<= 15 KB) live in LDS and are built by the
// workgroup itself, so a tip-tip op reads 2 B/site and writes the CLV.
//
// Arithmetic order is the reference's: see numerics.hpp.
#include <algorithm>
#include <vector>

#include "ctx.hpp"
#include "numerics.hpp"
#include "partials_fused.hpp"
#include <stdlib.h>


template <int RC>
__device__ __forceinline__ bool site_all(bool lane_flag)
{
  return group_all<RC>(lane_flag);
}

// ---------------------------------------------------------------- 4 states
//
// One lane per 16 BYTES (two states): lanes 2m and 2m+1 own one (site, rate)
// element, so every global load/store instruction of a wave is one contiguous
// 1 KiB -- whole 128-B lines per instruction, for reads and writes alike.  The
// pair swaps its two doubles with a DPP quad_perm (VALU, no LDS), after which
// lane h computes output states 2h and 2h+1.  Summation order is unchanged:
//   row . v = (m0 v0 + m1 v1) + (m2 v2 + m3 v3)
// lane 0 forms (own) + (partner), lane 1 forms (own) + (partner) with own =
// columns 2,3 -- the same two partial sums added in the other order, which is
// bitwise identical because IEEE addition commutes.

// row sums of one P-matrix for the 16 DNA ambiguity codes, in LDS:
// tab[(code * RC + k) * 4 + i] = sum_{j in code} P[k][i][j]
template <int RC>
__device__ __forceinline__ void build_tip_table4(double * tab, const double * __restrict__ mat)
{
  for (unsigned int t = threadIdx.x; t < 16 * RC * 4; t += blockDim.x)
  {
    const unsigned int code = t / (RC * 4), ki = t % (RC * 4);
    tab[t] = masksum4(mat + ki * 4, code);
  }
}

// All three 4-state CLV updates.  KIND 0 = inner-inner, 1 = tip-inner, 2 = tip-tip.
//
// W = 2*RC lanes make one site; a wave works in ROUNDS of 64 sites = W sub-steps
// of 64/W sites.  Per round the per-site data (tip codes, inherited scaler
// counts, the new scaler count) moves once, one element per lane.
// Lane l owns site (l % W) * (64/W) + l / W of the round: in sub-step j = l % W
// its group is exactly that site, so the site's scaling decision is already in
// the lane when the round ends.
//
// GATHER (site repeats): "site" then means a row of the parent, and each child is read
// at the row a.lidx / a.ridx names for it (a tip character when the child is a tip;
// nullptr = same index as the parent row).  The row indices of a round are one
// coalesced load per child; every sub-step takes its own with a __shfl.
template <int RC, int MODE, bool NT, int KIND, bool GATHER>
__global__ __launch_bounds__(256) void k_dna_partials(PartialsBatch batch)
{
  const PartialsArgs & a = batch.op[blockIdx.y];
  constexpr unsigned int W = 2 * RC;
  constexpr unsigned int SPS = 64 / W;
  __shared__ double tabl[KIND >= 1 ? 16 * RC * 4 : 1];
  __shared__ double tabr[KIND == 2 ? 16 * RC * 4 : 1];
  const unsigned int lane = threadIdx.x & 63u;
  const unsigned int h = lane & 1u;
  const unsigned int k = (lane >> 1) & (RC - 1);
  half_rows pl, pr;
  if (KIND == 0) pl.load(a.lmat, k, h);
  if (KIND <= 1) pr.load(a.rmat, k, h);
  if (KIND >= 1) build_tip_table4<RC>(tabl, a.lmat);
  if (KIND == 2) build_tip_table4<RC>(tabr, a.rmat);
  if (KIND >= 1) __syncthreads();

  const size_t sites = a.sites;
  const size_t total = sites * W; // 16-byte granules
  const size_t rounds = (sites + 63) / 64;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  const unsigned int own = (lane & (W - 1)) * SPS + lane / W; // owned site within a round
  const double2 * __restrict__ L = reinterpret_cast<const double2 *>(a.left);
  const double2 * __restrict__ R = reinterpret_cast<const double2 *>(a.right);
  double2 * __restrict__ out = reinterpret_cast<double2 *>(a.parent);
  // absent scalers read a zero word, unconditionally: a load inside a branch
  // makes the compiler wait for it -- and for every CLV load in flight -- at once
  const unsigned int * ls = (KIND == 0 && a.lscaler) ? a.lscaler : a.zero;
  const unsigned int * rs = (KIND <= 1 && a.rscaler) ? a.rscaler : a.zero;
  const bool has_l = (KIND == 0 && a.lscaler), has_r = (KIND <= 1 && a.rscaler);

  for (size_t r = wave; r < rounds; r += nwaves)
  {
    const size_t site0 = r * 64;
    const size_t n_own = site0 + own;
    const bool own_ok = n_own < sites;
    unsigned int codes_l = 0, codes_r = 0, base = 0;
    // rows of the two children for the 64 parent rows of this round (the lists carry
    // zeroed slack, so lanes past the last row gather row 0)
    unsigned int li_round = (unsigned int)(site0 + lane), ri_round = li_round;
    if (GATHER && a.lidx) li_round = a.lidx[site0 + lane];
    if (GATHER && a.ridx) ri_round = a.ridx[site0 + lane];
    if (KIND >= 1)
      codes_l = (GATHER && a.lidx) ? li_round : ((site0 + lane < sites) ? a.ltip[site0 + lane] : 0);
    if (KIND == 2)
      codes_r = (GATHER && a.ridx) ? ri_round : ((site0 + lane < sites) ? a.rtip[site0 + lane] : 0);
    if (MODE == SCALE_SITE && KIND != 2)
    {
      size_t nl = n_own, nr = n_own;
      if (GATHER)
      {
        nl = (unsigned int)__shfl((int)li_round, (int)own, 64);
        nr = (unsigned int)__shfl((int)ri_round, (int)own, 64);
      }
      base = ls[(has_l && own_ok) ? nl : 0] + rs[(has_r && own_ok) ? nr : 0];
    }
    bool own_scaled = false;

#pragma unroll
    for (unsigned int j = 0; j < W; ++j)
    {
      const size_t g = (site0 + (size_t)j * SPS) * W + lane;
      const bool act = g < total;
      const size_t gc = act ? g : 0;
      const int src = (int)(j * SPS + lane / W); // lane holding this site's per-round data
      // where the children's (site, rate) elements are: the parent's own granule index,
      // or row * W + position within the row
      size_t gl = gc, gr = gc;
      if (GATHER)
      {
        if (KIND == 0) gl = (size_t)(unsigned int)__shfl((int)li_round, src, 64) * W + (lane & (W - 1));
        if (KIND <= 1) gr = (size_t)(unsigned int)__shfl((int)ri_round, src, 64) * W + (lane & (W - 1));
      }
      double p0, p1;
      if (KIND == 0)
      {
        const double2 lo = ld16<NT>(L + gl), ro = ld16<NT>(R + gr);
        const double2 lp = make_double2(dpp_pair_swap(lo.x), dpp_pair_swap(lo.y));
        const double2 rp = make_double2(dpp_pair_swap(ro.x), dpp_pair_swap(ro.y));
        p0 = pl.dot(0, lo, lp) * pr.dot(0, ro, rp);
        p1 = pl.dot(1, lo, lp) * pr.dot(1, ro, rp);
      }
      else if (KIND == 1)
      {
        const double2 ro = ld16<NT>(R + gr);
        const unsigned int code = (unsigned int)__shfl((int)codes_l, src, 64) & 15u;
        const double2 rp = make_double2(dpp_pair_swap(ro.x), dpp_pair_swap(ro.y));
        const double2 tl = *reinterpret_cast<const double2 *>(tabl + (code * RC + k) * 4 + 2 * h);
        p0 = tl.x * pr.dot(0, ro, rp);
        p1 = tl.y * pr.dot(1, ro, rp);
      }
      else
      {
        // The reference materialises a 256-entry pair table (core_partials_avx.c:262);
        // the two 16-entry row-sum tables give the same products without it.
        const unsigned int cl = (unsigned int)__shfl((int)codes_l, src, 64) & 15u;
        const unsigned int cr = (unsigned int)__shfl((int)codes_r, src, 64) & 15u;
        const double2 tl = *reinterpret_cast<const double2 *>(tabl + (cl * RC + k) * 4 + 2 * h);
        const double2 tr = *reinterpret_cast<const double2 *>(tabr + (cr * RC + k) * 4 + 2 * h);
        p0 = tl.x * tr.x;
        p1 = tl.y * tr.y;
      }

      // scaling rule of core_partials_avx.c:486-527; tip-tip never scales and
      // clears its scaler (core_partials_avx.c:598-599)
      bool scale = false;
      if (KIND != 2 && MODE != SCALE_NONE)
      {
        const bool small = (p0 < PLLHIP_SCALE_THRESHOLD) & (p1 < PLLHIP_SCALE_THRESHOLD);
        scale = (MODE == SCALE_RATE) ? group_all<2>(small || !act) : group_all<W>(small || !act);
        if (scale)
        {
          p0 *= PLLHIP_SCALE_FACTOR;
          p1 *= PLLHIP_SCALE_FACTOR;
        }
      }
      // measured at 1 M sites: the write-only tip-tip stream is faster with plain stores
      // (22.5 vs 28.3 us), the read-read-write streams with non-temporal ones (64.5 vs 70.7 us)
      if (act) st16<(NT && KIND != 2)>(out + g, p0, p1);
      if (MODE == SCALE_RATE)
      {
        // one count per (site, rate): 4 bytes per 32 bytes of CLV, kept per sub-step
        const size_t e = gc >> 1;
        const unsigned int inh = (KIND == 2) ? 0u : ls[has_l ? (gl >> 1) : 0] + rs[has_r ? (gr >> 1) : 0];
        if (act && h == 0) a.pscaler[e] = inh + (scale ? 1u : 0u);
      }
      if (MODE == SCALE_SITE && (lane & (W - 1)) == j) own_scaled = scale;
    }
    if (MODE == SCALE_SITE && own_ok) a.pscaler[n_own] = base + (own_scaled ? 1u : 0u);
  }
}

// ------------------------------------------------- any state count (fallback)
//
// One lane per SITE, looping over rate categories and states; P-matrices and
// tip tables are read through L1/L2.  The last resort: more than 64 states (everything
// else runs on the kernels of partials_gen_tile.hip).
// Orders: 4 states pairwise; 20 states the AVX2-flag order (ii fused, ti not);
// otherwise left-to-right like the plain C kernels.

__device__ __forceinline__ double gen_dot(const double * __restrict__ row, const double * v,
                                          unsigned int S, bool fused)
{
  const dview vv{v};
  if (S == 4) return dot4(row, v[0], v[1], v[2], v[3]);
  if (S == 20) return fused ? dot_strided4<true>(row, vv, S) : dot_strided4<false>(row, vv, S);
  return dot_seq(row, vv, S);
}

__device__ __forceinline__ double gen_tipsum(const double * __restrict__ row, unsigned int mask,
                                             unsigned int S)
{
  return (S == 4) ? masksum4(row, mask) : masksum_seq(row, mask, S);
}

template <int KIND> // 0 = ii, 1 = ti, 2 = tt
__global__ __launch_bounds__(128) void k_gen_partials(PartialsArgs a, int mode)
{
  const unsigned int S = a.states, R = a.rate_cats;
  const size_t span = (size_t)S * R;
  for (size_t n = blockIdx.x * (size_t)blockDim.x + threadIdx.x; n < a.sites;
       n += (size_t)gridDim.x * blockDim.x)
  {
    double * par = a.parent + n * span;
    const double * lc = (KIND == 0) ? a.left + n * span : nullptr;
    const double * rc = (KIND != 2) ? a.right + n * span : nullptr;
    unsigned int lmask = 0, rmask = 0;
    if (KIND >= 1)
    {
      const unsigned int c = a.ltip[n];
      lmask = (S == 4) ? c : a.tipmap[c];
    }
    if (KIND == 2)
    {
      const unsigned int c = a.rtip[n];
      rmask = (S == 4) ? c : a.tipmap[c];
    }
    bool site_small = true;
    for (unsigned int k = 0; k < R; ++k)
    {
      const double * lm = a.lmat + (size_t)k * S * S;
      const double * rm = a.rmat + (size_t)k * S * S;
      bool rate_small = true;
      for (unsigned int i = 0; i < S; ++i)
      {
        const double x = (KIND == 0) ? gen_dot(lm + i * S, lc + k * S, S, true)
                                     : gen_tipsum(lm + i * S, lmask, S);
        const double y = (KIND == 2) ? gen_tipsum(rm + i * S, rmask, S)
                                     : gen_dot(rm + i * S, rc + k * S, S, KIND == 0);
        const double p = x * y;
        par[k * S + i] = p;
        rate_small = rate_small && (p < PLLHIP_SCALE_THRESHOLD);
      }
      if (KIND != 2 && mode == SCALE_RATE)
      {
        unsigned int base = 0;
        if (KIND == 0 && a.lscaler) base += a.lscaler[n * R + k];
        if (a.rscaler) base += a.rscaler[n * R + k];
        if (rate_small)
          for (unsigned int i = 0; i < S; ++i) par[k * S + i] *= PLLHIP_SCALE_FACTOR;
        a.pscaler[n * R + k] = base + (rate_small ? 1u : 0u);
      }
      site_small = site_small && rate_small;
    }
    if (KIND == 2)
    {
      if (mode == SCALE_SITE) a.pscaler[n] = 0u;
      if (mode == SCALE_RATE)
        for (unsigned int k = 0; k < R; ++k) a.pscaler[n * R + k] = 0u;
    }
    else if (mode == SCALE_SITE)
    {
      unsigned int base = 0;
      if (KIND == 0 && a.lscaler) base += a.lscaler[n];
      if (a.rscaler) base += a.rscaler[n];
      if (site_small)
        for (unsigned int t = 0; t < span; ++t) par[t] *= PLLHIP_SCALE_FACTOR;
      a.pscaler[n] = base + (site_small ? 1u : 0u);
    }
  }
}

// ---------------------------------------------------------------- dispatch

#define LAUNCH_DNA_GATHER(RCV, MODEV, NTV, KINDV)                                           \
  do {                                                                                     \
    if (gather) k_dna_partials<RCV, MODEV, NTV, KINDV, true><<<grid, 256, 0, s>>>(b);      \
    else k_dna_partials<RCV, MODEV, NTV, KINDV, false><<<grid, 256, 0, s>>>(b);            \
  } while (0)

#define LAUNCH_DNA_KIND(RCV, MODEV, NTV)                          \
  do {                                                            \
    if (kind == 0) LAUNCH_DNA_GATHER(RCV, MODEV, NTV, 0);         \
    else if (kind == 1) LAUNCH_DNA_GATHER(RCV, MODEV, NTV, 1);    \
    else LAUNCH_DNA_GATHER(RCV, MODEV, NTV, 2);                   \
  } while (0)

#define LAUNCH_DNA_MODE(RCV, NTV)                                 \
  do {                                                            \
    if (mode == SCALE_NONE) LAUNCH_DNA_KIND(RCV, 0, NTV);         \
    else if (mode == SCALE_SITE) LAUNCH_DNA_KIND(RCV, 1, NTV);    \
    else LAUNCH_DNA_KIND(RCV, 2, NTV);                            \
  } while (0)

static bool fast_rc(unsigned int rc)
{
  return rc == 1 || rc == 2 || rc == 4 || rc == 8 || rc == 16;
}

// 4-state ops of one kind and scaling mode, mutually independent, in one launch
static int pllhip_launch_dna_batch(pllhip_ctx * c, const PartialsBatch & b, unsigned int count,
                                   int kind, int mode)
{
  const PartialsArgs & a = b.op[0];
  const unsigned int R = a.rate_cats;
  hipStream_t s = c->stream;
  // a wave consumes 64 sites per round; with site repeats the ops of a launch differ in
  // their number of rows: the grid covers the largest, a workgroup past its op's rows
  // returns at once
  size_t rows_max = 0;
  bool gather = false;
  for (unsigned int i = 0; i < count; ++i)
  {
    if (b.op[i].sites > rows_max) rows_max = b.op[i].sites;
    gather = gather || b.op[i].lidx || b.op[i].ridx;
  }
  const dim3 grid(pllhip_stream_grid(c, (rows_max + 63) / 64 * 64, 256), count);
  // the non-temporal variants exist for the common 4-category case only; rows that are
  // gathered are re-read by several parent rows and stay on the default policy
  const bool nt = (R == 4) && pllhip_use_nt(c) && !gather;
  switch (R)
  {
    case 1: LAUNCH_DNA_MODE(1, false); break;
    case 2: LAUNCH_DNA_MODE(2, false); break;
    case 4: if (nt) LAUNCH_DNA_MODE(4, true); else LAUNCH_DNA_MODE(4, false); break;
    case 8: LAUNCH_DNA_MODE(8, false); break;
    default: LAUNCH_DNA_MODE(16, false); break;
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// kind: 0 = inner-inner, 1 = tip-inner (tip on the left), 2 = tip-tip
int pllhip_launch_partials(pllhip_ctx * c, const PartialsArgs & a_in, int kind, int mode,
                           int prof_kind)
{
  PartialsArgs a = a_in;
  const unsigned int R = a.rate_cats;
  hipStream_t s = c->stream;
  pllhip_prof_scope prof(c, prof_kind >= 0 ? prof_kind : PLLHIP_PROF_PARTIALS_II + kind);

  if (a.states == 4 && fast_rc(R))
  {
    PartialsBatch b;
    b.op[0] = a;
    HIP_TRY(pllhip_launch_dna_batch(c, b, 1, kind, mode) ? hipErrorLaunchFailure : hipSuccess);
    return 0;
  }
  else if (a.states == 20 && pllhip_aa_fast_covers(c, kind))
  {
    // matrix-core / round-based kernels (partials_aa_mfma.hip)
    PartialsBatch b;
    b.op[0] = a;
    return pllhip_launch_aa_batch(c, b, 1, kind, mode);
  }
  else if (pllhip_gen_tile_covers(c))
  {
    // LDS-tiled kernels for any other state count (partials_gen_tile.hip)
    PartialsBatch b;
    b.op[0] = a;
    return pllhip_launch_gen_batch(c, b, 1, kind, mode);
  }
  else
  {
    const unsigned int grid = pllhip_stream_grid(c, a.sites, 128);
    if (kind == 0) k_gen_partials<0><<<grid, 128, 0, s>>>(a, mode);
    if (kind == 1) k_gen_partials<1><<<grid, 128, 0, s>>>(a, mode);
    if (kind == 2) k_gen_partials<2><<<grid, 128, 0, s>>>(a, mode);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// the plan of the previous per-level call of a context (see pllhip_update_partials)
struct pllhip_planned_op
{
  PartialsArgs a;
  unsigned int key; // level << 8 | kind << 4 | mode
  unsigned int plain_key; // the same without the lookup kind (20 states)
  unsigned int order;
};
struct pllhip_level_cache
{
  std::vector<pllhip_op_t> last_ops;
  unsigned int last_maxstates = 0;
  unsigned int epoch = 0; // ctx->layout_epoch the plan was made under
  std::vector<pllhip_planned_op> plan;
  std::vector<PartialsArgs> by_pos;
  std::vector<std::pair<int, int>> cherry;
};
void pllhip_level_cache_free(pllhip_ctx * c)
{
  delete c->level_cache;
  c->level_cache = nullptr;
}

// resolves one op into kernel arguments; kind and mode as in pllhip_launch_partials
static int resolve_op(pllhip_ctx * c, const pllhip_op_t & op, PartialsArgs & a, int & kind, int & mode)
{
  const unsigned int nodes = (unsigned int)c->clv.size();
  if (op.parent_clv >= nodes || op.child1_clv >= nodes || op.child2_clv >= nodes ||
      !c->clv[op.parent_clv])
  {
    pllhip_set_error("pllhip_update_partials: clv index out of range or parent is a tip");
    return -1;
  }
  if (op.child1_matrix >= c->sh.prob_matrices || op.child2_matrix >= c->sh.prob_matrices)
  {
    pllhip_set_error("pllhip_update_partials: matrix index out of range");
    return -1;
  }
  const int nsc = (int)c->sh.scale_buffers;
  if (op.parent_scaler >= nsc || op.child1_scaler >= nsc || op.child2_scaler >= nsc)
  {
    pllhip_set_error("pllhip_update_partials: scaler index out of range");
    return -1;
  }

  const bool t1 = pllhip_is_tip(c, op.child1_clv), t2 = pllhip_is_tip(c, op.child2_clv);
  memset(&a, 0, sizeof(a));
  a.parent = c->clv[op.parent_clv];
  a.pscaler = pllhip_scaler_ptr(c, op.parent_scaler);
  a.tipmap = c->tipmap;
  a.zero = c->d_zero;
  a.sites = c->sh.sites;
  a.rate_cats = c->sh.rate_cats;
  a.states = c->sh.states;
  a.maxstates = c->maxstates;
  if (t1 && t2)
  {
    kind = 2;
    a.ltip = pllhip_tip_ptr(c, op.child1_clv);
    a.rtip = pllhip_tip_ptr(c, op.child2_clv);
    a.lmat = pllhip_pmat_ptr(c, op.child1_matrix);
    a.rmat = pllhip_pmat_ptr(c, op.child2_matrix);
  }
  else if (t1 || t2)
  {
    // the tip is always presented as the "left" child (partials.c:91-112)
    kind = 1;
    const unsigned int tip = t1 ? op.child1_clv : op.child2_clv;
    const unsigned int inner = t1 ? op.child2_clv : op.child1_clv;
    a.ltip = pllhip_tip_ptr(c, tip);
    a.right = c->clv[inner];
    a.lmat = pllhip_pmat_ptr(c, t1 ? op.child1_matrix : op.child2_matrix);
    a.rmat = pllhip_pmat_ptr(c, t1 ? op.child2_matrix : op.child1_matrix);
    a.rscaler = pllhip_scaler_ptr(c, t1 ? op.child2_scaler : op.child1_scaler);
  }
  else
  {
    kind = 0;
    a.left = c->clv[op.child1_clv];
    a.right = c->clv[op.child2_clv];
    a.lmat = pllhip_pmat_ptr(c, op.child1_matrix);
    a.rmat = pllhip_pmat_ptr(c, op.child2_matrix);
    a.lscaler = pllhip_scaler_ptr(c, op.child1_scaler);
    a.rscaler = pllhip_scaler_ptr(c, op.child2_scaler);
  }
  if ((kind == 0 && (!a.left || !a.right)) || (kind == 1 && !a.right))
  {
    pllhip_set_error("pllhip_update_partials: child CLV missing");
    return -1;
  }
  if (kind >= 1 && a.states != 4 && c->maxstates == 0)
  {
    pllhip_set_error("pllhip_update_partials: tipmap not uploaded");
    return -1;
  }

  mode = !a.pscaler ? SCALE_NONE : (c->sh.rate_scalers ? SCALE_RATE : SCALE_SITE);

  // site repeats: the parent may be stored by class (then the host has named, per
  // class, the row of each child), and an inner child may be (then a parent stored per
  // site reads it through the child's site -> row map)
  if (!c->rows.empty())
  {
    const pllhip_ctx::node_rows & pr = c->rows[op.parent_clv];
    const unsigned int * m1, * m2; // row maps for child 1 / child 2
    if (pr.classes)
    {
      a.sites = pr.classes;
      m1 = pr.lrow;
      m2 = pr.rrow;
    }
    else
    {
      m1 = (!t1 && c->rows[op.child1_clv].classes) ? c->rows[op.child1_clv].site_id : nullptr;
      m2 = (!t2 && c->rows[op.child2_clv].classes) ? c->rows[op.child2_clv].site_id : nullptr;
    }
    if (kind == 1 && !t1)
    {
      a.lidx = m2; // the tip (child 2) is presented as the left child
      a.ridx = m1;
    }
    else
    {
      a.lidx = m1;
      a.ridx = m2;
    }
  }
  return 0;
}

int pllhip_resolve_op(pllhip_ctx * c, const pllhip_op_t & op, PartialsArgs & a, int & kind, int & mode)
{
  return resolve_op(c, op, a, kind, mode);
}

// Small partitions (fewer workgroup tiles than the device has slots: one "round"), round 4.  Below the sizes from
// which the whole-list kernels always pay, the faster path depends on the LIST: a whole-list launch costs a fixed
// preparation plus a time per op that does not depend on the site count (a tile walks the list alone), the per-level
// path a launch per dependency level and op kind plus the bytes.  Both are estimated from the list -- its dependency
// levels, its op kinds -- with constants measured on one MI355X (tools/small_partitions_ab.sh,
// profiles/r4_small_partitions_ab.txt: 64-taxon balanced and 200-taxon random trees, 2,000-16,000 sites, full and
// partial traversals; the per-level path of a 200-taxon random tree is 27 levels = 60 launches, that of a balanced
// 64-taxon tree 6 levels).  Returns true when the whole-list launch is estimated faster.  Either path gives the
// same bits; PLLHIP_FUSED=0 / 2 still forces one.  (Pure host logic, exported for the CPU tests: 1 = whole list, 0 = per
// level, -1 = an index out of range.)
extern "C" int pllhip_small_partition_estimate(unsigned int states, unsigned int sites, unsigned int tips,
                                               unsigned int clv_buffers, int pattern_tip, const pllhip_op_t * ops,
                                               unsigned int count, double * whole_us_out, double * level_us_out)
{
  static thread_local std::vector<unsigned short> level, made_by_tt;
  const size_t nclv = (size_t)tips + clv_buffers;
  level.assign(nclv, 0);
  made_by_tt.assign(nclv, 0);
  unsigned int per_level_kind[64][3];
  memset(per_level_kind, 0, sizeof(per_level_kind));
  unsigned int n_tt = 0, n_ti = 0, n_ii = 0, n_lookup = 0, launches = 0;
  for (unsigned int i = 0; i < count; ++i)
  {
    const pllhip_op_t & op = ops[i];
    if (op.parent_clv >= nclv || op.child1_clv >= nclv || op.child2_clv >= nclv) return -1; // (the path taken reports it)
    const bool t1 = pattern_tip && op.child1_clv < tips, t2 = pattern_tip && op.child2_clv < tips;
    const unsigned int l = 1u + std::max<unsigned int>(level[op.child1_clv], level[op.child2_clv]);
    const unsigned int kind = t1 && t2 ? 2u : (t1 || t2 ? 1u : 0u);
    if (kind == 2u) ++n_tt;
    else if (kind == 1u)
    {
      if (made_by_tt[t1 ? op.child2_clv : op.child1_clv]) ++n_lookup; else ++n_ti;
    }
    else
    {
      if (made_by_tt[op.child1_clv] && made_by_tt[op.child2_clv]) ++n_lookup; else ++n_ii;
    }
    level[op.parent_clv] = (unsigned short)(l < 65535u ? l : 65535u);
    made_by_tt[op.parent_clv] = kind == 2u;
    if (l < 64u)
    {
      if (per_level_kind[l][kind]++ % PLLHIP_BATCH_MAX == 0) ++launches;
    }
    else ++launches;
  }
  // Round 5: a list that splits into independent segments (partials_fused.hpp) is walked segment by segment by
  // different waves at these sizes -- the walk that counts is the longest segment's (a full traversal: the longer side
  // of the root edge), and a list seen for the first time pays its planning on this path (4 states: 0.16 us per op;
  // 20 states: the fixed term 32 -> 40 us).  A three-op list on the 4-state whole-list path is one launch since the tile
  // counters come in two sets (no table launch, no memset): 49 against 55-58 us per step from Python.  Measured again with the segments in (profiles/r5_small_partitions_ab.txt): 4 states x 64 taxa x
  // 12,000 sites 64 (per level) against 54 us, 20 states x 64 taxa x 2,000 sites 115 against 106.
  double walk = 1.0;
  {
    unsigned int nsc = 0;
    for (unsigned int i = 0; i < count; ++i)
    {
      const pllhip_op_t & op = ops[i];
      for (int sc : {op.parent_scaler, op.child1_scaler, op.child2_scaler})
        if (sc >= 0 && (unsigned int)sc + 1 > nsc) nsc = (unsigned int)sc + 1;
    }
    const FusedGeom geom = {nclv, nsc, tips, pattern_tip != 0};
    static thread_local std::vector<unsigned int> seg_of;
    const unsigned int nsegs = pllhip_fused_segments(geom, ops, count, PLLHIP_FUSED_MAX_SEGS, seg_of);
    if (nsegs > 1)
    {
      unsigned int longest = 0;
      for (unsigned int sg = 0; sg < nsegs; ++sg)
        longest = std::max(longest, (unsigned int)std::count(seg_of.begin(), seg_of.end(), sg));
      walk = (double)longest / count;
    }
  }
  double whole_us, level_us;
  if (states == 4)
  {
    whole_us = 12.0 + 0.16 * count + 0.70 * count * walk; // (fixed + planning a list of this length + the walk)
    level_us = 5.5 * launches + (double)count * sites * 265.0 / 8.5e6;
  }
  else
  {
    // (20 states: the per-level path adds table launches per level -- tip tables, the lookup ops' tables and kernels)
    whole_us = 40.0 + (1.6 * n_tt + 2.0 * n_lookup + 3.2 * n_ii + 3.4 * n_ti) * walk;
    level_us = 7.5 * 2.2 * launches + (double)count * sites * 1300.0 / 7.0e6;
  }
  if (whole_us_out) *whole_us_out = whole_us;
  if (level_us_out) *level_us_out = level_us;
  return whole_us < level_us ? 1 : 0;
}

static bool whole_list_pays_when_small(const pllhip_ctx * c, const pllhip_op_t * ops, unsigned int count)
{
  return pllhip_small_partition_estimate(c->sh.states, c->sh.sites, c->sh.tips, (unsigned int)(c->clv.size() - c->sh.tips),
                                         c->sh.pattern_tip ? 1 : 0, ops, count, nullptr, nullptr) == 1;
}

// ---- the scaling certificate (round 6; ctx.hpp) ----
// The whole-list 20-state kernel runs the mat-vec of a tip-inner op on the matrix cores: fused multiply-adds where the
// reference (core_partials_avx.c:1229-1284) rounds products and sums separately -- 1e-15 per op, and from then on
// everything computed from that CLV.  north_star asks for scaler counts bit for bit.  A count can only differ where
// a scaling decision does, and a decision -- "every entry of the site (or of its rate block) below 2^-256",
// core_partials_avx2.c:752-800 -- only where the largest entry lies within the accumulated difference of the
// threshold.  So every op that scales a marked value tests for a largest entry within a window 170 (reference order:
// 2.6) times wider than the bound on that difference, and raises a word in host memory.  The host looks at the word
// before anything else reads or changes what the list read or wrote, and on a raised flag runs the list again with
// every op in the reference's order.  Then: no flag => every count of the call is the reference's; flag => the list
// as the round-5 default ran it.  What is left is a list of reference-order arithmetic on operands that an EARLIER
// call left marked (a partial traversal): nothing to run again; a largest entry within the narrow window there is
// counted as "uncertified" (pllhip_cert_stats; never seen: the window is 6e-11 relative).
void pllhip_cert_mark_clv(pllhip_ctx * c, unsigned int idx, double err)
{
  if (!(err > 0.0) && !c->n_inexact) return;
  if (c->clv_err.size() != c->clv.size()) c->clv_err.assign(c->clv.size(), 0.0);
  if (idx >= c->clv_err.size()) return;
  if (c->clv_err[idx] > 0.0 && !(err > 0.0)) --c->n_inexact;
  if (!(c->clv_err[idx] > 0.0) && err > 0.0) ++c->n_inexact;
  c->clv_err[idx] = err > 0.0 ? err : 0.0;
}

int pllhip_cert_resolve(pllhip_ctx * c, bool * rerun, bool drained)
{
  if (rerun) *rerun = false;
  if (!c->cert_pending) return 0;
  HIP_TRY(hipSetDevice(c->sh.device));
  if (!drained) HIP_TRY(hipStreamSynchronize(c->stream));
  c->cert_pending = false;
  const unsigned int flag = *(volatile unsigned int *)c->h_cert;
  if (!flag) return 0;
  *(volatile unsigned int *)c->h_cert = 0u;
  ++c->cert_stats[1];
  if (c->cert_kind != 1)
  {
    ++c->cert_stats[3];
    return 0;
  }
  // the list again, every op in the reference's order (its operands from earlier calls are as they were: a list
  // that overwrites one of them never runs on the matrix cores' tip-inner path, partials_aa_fused.hip)
  ++c->cert_stats[2];
  if (c->fused_debug) fprintf(stderr, "pllhip: scaling certificate raised, %zu ops run again in the reference's order\n", c->cert_ops.size());
  const std::vector<pllhip_op_t> again(c->cert_ops);
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->cert_force_exact = true;
  int rc = pllhip_update_partials(c, again.data(), (unsigned int)again.size());
  c->cert_force_exact = false;
  if (rc) return rc;
  if (rerun) *rerun = true;
  if (c->cert_pending) rc = pllhip_cert_resolve(c); // (operands marked by earlier calls: the narrow window)
  return rc;
}

extern "C" int pllhip_cert_stats(pllhip_ctx_t * c, unsigned long long * out4)
{
  for (int t = 0; t < 4; ++t) out4[t] = 0;
  if (!c->shards.empty())
  {
    for (pllhip_ctx * s : c->shards)
      for (int t = 0; t < 4; ++t) out4[t] += s->cert_stats[t];
    return 0;
  }
  for (int t = 0; t < 4; ++t) out4[t] = c->cert_stats[t];
  return 0;
}

extern "C" int pllhip_update_partials(pllhip_ctx_t * c, const pllhip_op_t * ops, unsigned int count)
{
  PLLHIP_ALL_SHARDS_PAR(c, pllhip_update_partials(s, ops, count)); // (enqueued on every device; nothing waits)
  HIP_TRY(hipSetDevice(c->sh.device));
  if (c->cert_pending)
  {
    const int rc = pllhip_cert_resolve(c);
    if (rc) return rc;
  }
  if (!c->rows.empty())
  {
    // (site repeats: which CLV each scale buffer belongs to -- a shard of a group expands its own mirrors, ctx.hip)
    if (c->scaler_owner.size() != c->sh.scale_buffers) c->scaler_owner.assign(c->sh.scale_buffers, -1);
    for (unsigned int i = 0; i < count; ++i)
      if (ops[i].parent_scaler >= 0 && (unsigned int)ops[i].parent_scaler < c->sh.scale_buffers)
        c->scaler_owner[ops[i].parent_scaler] = (int)ops[i].parent_clv;
  }
  // Dependencies between the ops of a list follow BUFFER INDICES, not tree shape
  // (unrooted trees reuse CLV slots, partials.c:184-212).  Each op gets a level:
  // one more than the highest level among the earlier ops it must not overtake --
  // the writers of what it reads (RAW), the writer and the readers of what it writes
  // (WAW, WAR), on CLVs and on scale buffers alike.  Ops of one level are mutually
  // independent whatever their position in the list, so each level runs as one
  // launch per (kind, scaling mode) with blockIdx.y selecting the op.  A balanced
  // tree gives one launch per tree level; a post-order list of a random 200-taxon
  // tree, where kinds alternate, gives 3 launches per level instead of one per run of
  // equal kinds (111 -> see DESIGN.md 2.1).
  const bool no_batch = c->no_batch;
  const bool dna_fast = c->sh.states == 4 && fast_rc(c->sh.rate_cats);
  const bool aa_fast = c->sh.states == 20 && pllhip_aa_fast_covers(c, 0) &&
                       (!c->sh.pattern_tip || pllhip_aa_fast_covers(c, 2));
  const bool gen_fast = !dna_fast && !aa_fast && pllhip_gen_tile_covers(c);
  const bool batchable = (dna_fast || aa_fast || gen_fast) && !no_batch;
  if (!batchable)
  {
    for (unsigned int i = 0; i < count; ++i)
    {
      PartialsArgs a;
      int kind, mode;
      int rc = resolve_op(c, ops[i], a, kind, mode);
      if (rc) return rc;
      if ((rc = pllhip_launch_partials(c, a, kind, mode, -1))) return rc;
    }
    return 0;
  }

  // 4 states, more than one op, no site repeats: the whole list in one site-blocked
  // launch (partials_fused.hip)
  // (a wave takes the list op by op, ~1 us each: with fewer tiles than about one per SIMD --
  // 16 k sites at 4 rate categories -- the per-level launches are faster; measured for 64 taxa,
  // whole list vs per level: 8 k sites 74 vs 63 us, 16 k 78 vs 79, 24 k 84 vs 93, 33 k 100 vs 121,
  // 50 k 118 vs 164.  A property of the device -- tiles against SIMDs -- not a tuned number.)
  const size_t fused_tile_sites = (size_t)PLLHIP_FUSED_J * 64 / (2 * c->sh.rate_cats);
  static const bool small_rule = !(pllhip_env("PLLHIP_FUSED_SMALL") && atoi(pllhip_env("PLLHIP_FUSED_SMALL")) == 0);
  const bool whole_list_kind = (dna_fast && (c->sh.rate_cats <= 4 || c->sh.rate_cats == 8)) || (aa_fast && c->sh.rate_cats == 4);
  // (asked only below the sizes from which the whole-list kernels always pay -- the two rules below, in THIS device's
  // compute units: 16,384 sites on an MI355X's 256; derived, not a constant of its own, so that no size falls
  // between the two rules on another part.  The cost constants of the estimate are this part's, measured.)
  const size_t always_pays_from = dna_fast ? fused_tile_sites * (size_t)c->num_cus * 4 : (size_t)32 * c->num_cus * 2;
  const bool small_pays = small_rule && whole_list_kind && !c->no_fused && !c->force_fused && c->rows.empty() && count >= 2 &&
                          (size_t)c->sh.sites < always_pays_from && whole_list_pays_when_small(c, ops, count);
  const bool fused_pays = (size_t)c->sh.sites / fused_tile_sites >= (size_t)c->num_cus * 4 || c->force_fused || small_pays;
  // (8 rate categories: a tile is 8 sites, a P-matrix a whole 1 KB block per wave, a site's lanes a DPP row of 16.
  // Round 1's first attempt spilled and lost -- 5.6 against 10.2 G site-updates/s per level; the rebuilt kernel
  // needs 136-147 registers for it: 500 k sites x 64 taxa 1.49 ms against 3.07 per level, 20.9 against 10.1 G/s.)
  // (short lists -- the path to the root after one branch changed -- have little to keep on chip, but they
  // are one launch instead of one per op, and half the bytes: 1 M sites, whole list vs per level, 2 ops 134 vs
  // 134 us, 3 ops 180 vs 200, 5 ops 239 vs 282, 7 ops 192 vs 309, 15 ops 380 vs 662; 50 k sites 14 vs 17, 17 vs
  // 26, 27 vs 38, 25 vs 27, 35 vs 47 (tools/partial_traversal_timing.py, profiles/r3_short_lists.txt).  Until
  // round 3 they lost -- 3 ops 260 vs 200 us -- to the tile counter, not to their reloads: see the kernel.)
  if (dna_fast && (c->sh.rate_cats <= 4 || c->sh.rate_cats == 8) && !c->no_fused && fused_pays && c->rows.empty() && count >= 2)
  {
    if (c->fused_last_ops.size() == count && !c->fused_debug &&
        c->fused_last_epoch == c->layout_epoch &&
        memcmp(c->fused_last_ops.data(), ops, (size_t)count * sizeof(pllhip_op_t)) == 0)
    {
      pllhip_prof_scope prof(c, PLLHIP_PROF_PARTIALS_II);
      return pllhip_relaunch_fused(c);
    }
    c->fused_last_ops.clear();
    std::vector<PartialsArgs> args(count);
    std::vector<int> kinds(count), modes(count);
    for (unsigned int i = 0; i < count; ++i)
    {
      int rc = resolve_op(c, ops[i], args[i], kinds[i], modes[i]);
      if (rc) return rc;
    }
    // Three workgroups per CU (12 waves) hide the per-op latencies better than two, but
    // leave one LDS slot less per wave (6 against 7 at 4 rate categories): the 12-wave
    // configuration whenever the planner can keep every operand in a slot with it (values
    // that give their slot up, and operands written by earlier calls, are copied back from
    // HBM by LDS-DMA one op ahead).  A list the planner does not take -- counts that were not
    // written together with their CLV -- runs per level.
    const FusedGeom geom = {c->clv.size(), c->sh.scale_buffers, c->sh.tips, c->sh.pattern_tip != 0};
    // (PLLHIP_FUSED_WGS=2: the 8-wave, 7-slot configuration at once -- tests run both)
    unsigned int first_wgs = pllhip_env("PLLHIP_FUSED_WGS") && atoi(pllhip_env("PLLHIP_FUSED_WGS")) == 2 ? 2u : 3u;
#ifdef PLLHIP_FUSED_WPS4
    if (pllhip_env("PLLHIP_FUSED_WGS") && atoi(pllhip_env("PLLHIP_FUSED_WGS")) == 4) first_wgs = 4u; // (tool build)
#endif
    // Round 5: independent sub-lists (the two sides of the root edge of a full traversal) as SEGMENTS of one launch
    // -- (tile, segment) work items -- while the tiles alone do not fill the chip's wave slots eight times over:
    // below that a launch's time is quantised by rounds of the list's length (partials_fused.hpp).
    // PLLHIP_FUSED_SEGMENTS=0 / n: never / up to n whatever the size.
    unsigned int max_segs = (size_t)c->sh.sites / fused_tile_sites < (size_t)c->num_cus * 12 * 8 ? PLLHIP_FUSED_MAX_SEGS : 1u;
    if (const char * e = pllhip_env("PLLHIP_FUSED_SEGMENTS")) max_segs = (unsigned int)std::max(1, atoi(e));
    std::vector<unsigned int> seg_of;
    unsigned int nsegs = pllhip_fused_segments(geom, ops, count, max_segs, seg_of);
    std::vector<std::vector<FusedOp>> fplans;
    unsigned int nslots = 0;
    int rc = 1;
    for (; rc > 0; nsegs = 1) // (segments the planner does not take: once more as one list)
    {
      for (unsigned int wgs = first_wgs; rc > 0 && wgs >= 2u; --wgs)
      {
        nslots = pllhip_fused_slots(c, wgs);
        fplans.assign(nsegs, std::vector<FusedOp>());
        rc = 0;
        for (unsigned int sg = 0; sg < nsegs && rc == 0; ++sg)
        {
          unsigned int reloads = 0;
          if (nsegs == 1)
          {
            rc = pllhip_fused_plan(geom, ops, args.data(), kinds.data(), count, nslots, fplans[0], &reloads);
            continue;
          }
          std::vector<pllhip_op_t> sops;
          std::vector<PartialsArgs> sargs;
          std::vector<int> skinds, where;
          for (unsigned int i = 0; i < count; ++i)
            if (seg_of[i] == sg)
            {
              sops.push_back(ops[i]);
              sargs.push_back(args[i]);
              skinds.push_back(kinds[i]);
              where.push_back((int)i);
            }
          rc = pllhip_fused_plan(geom, sops.data(), sargs.data(), skinds.data(), (unsigned int)sops.size(), nslots, fplans[sg], &reloads);
          for (FusedOp & f : fplans[sg]) f.list_pos = where[f.list_pos];
        }
      }
      if (nsegs == 1) break;
    }
    if (rc < 0) return rc;
    if (rc == 0)
    {
      pllhip_prof_scope prof(c, PLLHIP_PROF_PARTIALS_II);
      rc = pllhip_launch_fused(c, fplans, nslots);
      if (rc == 0) c->fused_last_ops.assign(ops, ops + count);
      if (rc <= 0) return rc;
    }
    // (a list shape the kernel does not take: per-level launches below)
  }

  // 20 states, 4 rate categories: the whole list in one site-blocked launch on the matrix cores
  // (partials_aa_fused.hip); from one workgroup tile (32 sites) per workgroup slot of the device on
  if (aa_fast && c->sh.rate_cats == 4 && !c->no_fused && count >= 2 &&
      ((size_t)c->sh.sites / 32 >= (size_t)c->num_cus * 2 || c->force_fused || small_pays))
  {
    pllhip_prof_scope prof(c, PLLHIP_PROF_PARTIALS_II);
    const int rc = pllhip_aa_fused_update(c, ops, count);
    if (rc <= 0) return rc;
    // (a list or a partition it does not take: per-level launches below)
  }

  // The plan of the previous per-level call is kept: an identical op list (the usual case
  // while branch lengths or model parameters are optimised) goes straight to the launches.
  // (Not with site repeats: the row maps in the arguments change with the classes.)
  typedef pllhip_planned_op Planned;
  if (!c->level_cache) c->level_cache = new pllhip_level_cache();
  pllhip_level_cache & lc = *c->level_cache;
  std::vector<Planned> & plan = lc.plan;
  std::vector<PartialsArgs> & by_pos = lc.by_pos;
  std::vector<std::pair<int, int>> & cherry_kids = lc.cherry;
  const bool plan_kept = c->rows.empty() && lc.last_ops.size() == count && lc.last_maxstates == c->maxstates &&
                         lc.epoch == c->layout_epoch &&
                         memcmp(lc.last_ops.data(), ops, (size_t)count * sizeof(pllhip_op_t)) == 0;
  if (!plan_kept)
  {
  lc.last_ops.clear();
  plan.assign(count, Planned());
  // 20 states: an inner-inner op whose children are both tip-tip results of this list is
  // a table lookup (partials_aa_mfma.hip, k_aa_cherry_rounds): kind 3 here, with the two
  // producing ops remembered
  std::vector<int> tt_writer(c->clv.size(), -1);           // list op that wrote the CLV, if it was tip-tip
  cherry_kids.assign(count, {-1, -1});
  // highest level that wrote / has read each buffer since its last write
  std::vector<unsigned int> clv_w(c->clv.size(), 0u), clv_r(c->clv.size(), 0u);
  std::vector<unsigned int> sc_w(c->sh.scale_buffers, 0u), sc_r(c->sh.scale_buffers, 0u);
  auto upto = [](unsigned int & m, unsigned int v) { if (v > m) m = v; };
  for (unsigned int i = 0; i < count; ++i)
  {
    int kind, mode;
    int rc = resolve_op(c, ops[i], plan[i].a, kind, mode);
    if (rc) return rc;
    const pllhip_op_t & op = ops[i];
    unsigned int lvl = 0;
    upto(lvl, clv_w[op.child1_clv]);
    upto(lvl, clv_w[op.child2_clv]);
    upto(lvl, clv_w[op.parent_clv]);
    upto(lvl, clv_r[op.parent_clv]);
    if (op.child1_scaler >= 0) upto(lvl, sc_w[op.child1_scaler]);
    if (op.child2_scaler >= 0) upto(lvl, sc_w[op.child2_scaler]);
    if (op.parent_scaler >= 0)
    {
      upto(lvl, sc_w[op.parent_scaler]);
      upto(lvl, sc_r[op.parent_scaler]);
    }
    ++lvl;
    clv_w[op.parent_clv] = lvl;
    clv_r[op.parent_clv] = 0;
    upto(clv_r[op.child1_clv], lvl);
    upto(clv_r[op.child2_clv], lvl);
    if (op.parent_scaler >= 0)
    {
      sc_w[op.parent_scaler] = lvl;
      sc_r[op.parent_scaler] = 0;
    }
    if (op.child1_scaler >= 0) upto(sc_r[op.child1_scaler], lvl);
    if (op.child2_scaler >= 0) upto(sc_r[op.child2_scaler], lvl);
    const int plain_kind = kind;
    if (aa_fast && kind == 0 && tt_writer[op.child1_clv] >= 0 && tt_writer[op.child2_clv] >= 0 &&
        pllhip_aa_cherry_covers(c, mode))
    {
      cherry_kids[i] = {tt_writer[op.child1_clv], tt_writer[op.child2_clv]};
      kind = 3;
    }
    else if (aa_fast && kind == 1 && pllhip_aa_cherry_covers(c, mode))
    {
      // tip-inner over a tip-tip result: the same lookup with the tip's own table on one side
      const unsigned int inner = pllhip_is_tip(c, op.child1_clv) ? op.child2_clv : op.child1_clv;
      if (tt_writer[inner] >= 0)
      {
        cherry_kids[i] = {-2, tt_writer[inner]};
        kind = 3;
      }
    }
    tt_writer[op.parent_clv] = (plain_kind == 2) ? (int)i : -1;
    plan[i].key = (lvl << 8) | ((unsigned int)kind << 4) | (unsigned int)mode;
    plan[i].plain_key = (lvl << 8) | ((unsigned int)plain_kind << 4) | (unsigned int)mode;
    plan[i].order = i;
  }
  // Lookup ops or not: they save (1932 - 646) bytes per site and op at the kernels' ~5.5 TB/s,
  // and cost about eight small launches (~6 us each, tables for up to 12 ops) per tree level
  // that has any.  Used when the saving is at least twice the cost -- 64 taxa (16 such ops on
  // one level): from ~26 k sites on, measured crossover 20-30 k (20 k sites 425 vs 398 us per
  // evaluation, 50 k 766 vs 901).  PLLHIP_AA_CHERRY=0 / 2: never / whatever the size.
  {
    unsigned int lookups = 0;
    std::vector<unsigned int> levels;
    for (unsigned int i = 0; i < count; ++i)
      if (((plan[i].key >> 4) & 15u) == 3)
      {
        ++lookups;
        levels.push_back(plan[i].key >> 8);
      }
    std::sort(levels.begin(), levels.end());
    const size_t nlevels = std::unique(levels.begin(), levels.end()) - levels.begin();
    if (lookups && !pllhip_aa_cherry_pays(c, lookups, (unsigned int)nlevels))
      for (unsigned int i = 0; i < count; ++i) plan[i].key = plan[i].plain_key;
  }

  // (the producers' arguments by list position: the sort below moves the entries)
  by_pos.clear();
  if (aa_fast)
  {
    by_pos.resize(count);
    for (unsigned int i = 0; i < count; ++i) by_pos[i] = plan[i].a;
  }
  std::stable_sort(plan.begin(), plan.end(),
                   [](const Planned & x, const Planned & y) { return x.key < y.key; });
  if (c->rows.empty())
  {
    lc.last_ops.assign(ops, ops + count);
    lc.last_maxstates = c->maxstates;
    lc.epoch = c->layout_epoch;
  }
  } // !plan_kept

  // The scaling certificate on this path: its kernels work in the reference's order, so a value is marked only when
  // an operand is; the marks are walked in list order, and an op that scales a marked value tests (k_aa_ii_mfma).
  std::vector<unsigned char> op_marked;
  unsigned int cert_log2 = 0; // the window of this call: 2^-cert_log2, relative
  if (c->n_inexact)
  {
    op_marked.assign(count, 0);
    double worst = 0.0;
    for (unsigned int i = 0; i < count; ++i)
    {
      const pllhip_op_t & op = ops[i];
      const double in = (pllhip_is_tip(c, op.child1_clv) ? 0.0 : pllhip_cert_err(c, op.child1_clv)) +
                        (pllhip_is_tip(c, op.child2_clv) ? 0.0 : pllhip_cert_err(c, op.child2_clv));
      const double e = in > 0.0 ? in + PLLHIP_CERT_OP_ERR : 0.0;
      pllhip_cert_mark_clv(c, op.parent_clv, e);
      op_marked[i] = e > 0.0 && op.parent_scaler >= 0;
      if (op_marked[i] && e > worst) worst = e;
    }
    if (worst > 0.0)
    {
      c->cert_pending = true;
      c->cert_kind = 2;
      ++c->cert_stats[0];
      double w = 8.0 * worst;
      if (w > PLLHIP_CERT_WINDOW_MAX) ++c->cert_stats[3]; // (no window is wide enough: uncertified as it stands)
      w = std::min(std::max(w, PLLHIP_CERT_WINDOW_MIN), PLLHIP_CERT_WINDOW_MAX);
      int ex = 0;
      (void)frexp(w, &ex);           // w = m 2^ex, m in [0.5, 1): 2^ex >= w
      cert_log2 = (unsigned int)(-ex);
    }
  }
  PartialsBatch b;
  for (unsigned int i = 0; i < count;)
  {
    const unsigned int key = plan[i].key;
    const int kind = (int)((key >> 4) & 15u), mode = (int)(key & 15u);
    if (kind == 3)
    {
      std::vector<PartialsArgs> cops, k1, k2;
      while (i < count && plan[i].key == key)
      {
        cops.push_back(plan[i].a);
        if (cherry_kids[plan[i].order].first >= 0) k1.push_back(by_pos[cherry_kids[plan[i].order].first]);
        else
        {
          PartialsArgs none; // marks a tip-inner lookup op (no producing op on the tip's side)
          memset(&none, 0, sizeof(none));
          k1.push_back(none);
        }
        k2.push_back(by_pos[cherry_kids[plan[i].order].second]);
        ++i;
      }
      pllhip_prof_scope prof(c, PLLHIP_PROF_PARTIALS_II);
      int rc = pllhip_launch_aa_cherries(c, cops.data(), k1.data(), k2.data(), (unsigned int)cops.size(), mode);
      if (rc) return rc;
      continue;
    }
    unsigned int nb = 0;
    while (i < count && plan[i].key == key && nb < PLLHIP_BATCH_MAX)
    {
      b.op[nb] = plan[i].a;
      b.op[nb].cert = (!op_marked.empty() && op_marked[plan[i].order]) ? c->h_cert_dev : nullptr;
      if (b.op[nb].cert) b.op[nb].pad_ = cert_log2;
      ++nb;
      ++i;
    }
    pllhip_prof_scope prof(c, PLLHIP_PROF_PARTIALS_II + kind);
    int rc = dna_fast ? pllhip_launch_dna_batch(c, b, nb, kind, mode)
             : aa_fast ? pllhip_launch_aa_batch(c, b, nb, kind, mode)
                       : pllhip_launch_gen_batch(c, b, nb, kind, mode);
    if (rc) return rc;
  }
  return 0;
}
