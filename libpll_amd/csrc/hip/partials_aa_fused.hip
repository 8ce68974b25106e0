// partials_aa_fused.hip -- a whole op list of 20-state CLV updates in ONE kernel, site-blocked.
//
// Replaces, for pll_update_partials (partials.c:177-213) on 20-state data with 4 rate
// categories, the one-launch-per-tree-level path of partials.hip / partials_aa_mfma.hip, in
// which every child CLV makes a round trip through HBM (1932 B per inner-inner site-update:
// core_partials_avx2.c:568-803 reads two children and writes the parent).  As in the 4-state
// whole-list kernel (partials_fused.hip) a tile of sites is taken through the WHOLE list and
// what an op produced stays on chip for the op that consumes it: the list becomes a write
// stream of 644 B per site-update.
//
//   tile      a wave owns 8 sites = 2 sub-tiles of (4 sites x 4 rates) = the 16 columns of one
//             v_mfma_f64_4x4x4_4b: block b of the instruction is rate b, so ONE A operand
//             serves all four rates (each block has its own A) and an A operand fetched from
//             LDS is used for both sub-tiles.  (With the per-level kernel's tile of 16 sites x
//             one rate a value kept on chip would cost 10 KB per slot.)
//   slots     values live in REGISTERS, not LDS: the accumulator layout of the MFMA (lane =
//             (q, rate, site), states 4q..4q+3 and 16+q) is also its B-operand layout, so a
//             parent is consumed as it was produced -- 20 VGPRs per value and tile, five
//             values per wave (a balanced 64-taxon tree needs four, 128 taxa five); the slot
//             number of an op is wave-uniform and selects by a scalar branch.
//   matrices  the two 12.8 KB P-matrix blocks of an op are staged in LDS once per WORKGROUP
//             (four waves in lock-step: two barriers per op) by LDS-DMA, double-buffered by
//             halves -- the left block of op i+1 arrives while op i multiplies by its right
//             block -- in "operand order": the 64 lanes of one MFMA read 512 contiguous bytes
//             (k_af_prepare permutes every matrix of the list once per call).
//   order     the reference's, op kind by op kind -- CLVs and scaler counts are its bits.  Inner-inner ops
//             (core_partials_avx2.c:632-750): four FMA chains strided by j mod 4, then (a0+a1)+(a2+a3); the MFMA adds its
//             four products as a chain of FMAs in k order (tools/mfma_order_probe.hip), so a k-chunk {m, m+4, m+8, m+12}
//             is four steps of chain m, and the fifth step (state 16 + m) is one FMA on the vector unit (round 4: it was
//             a second, zero-padded MFMA).  Tip-inner ops (core_partials_avx.c:1097-1340, multiply and add rounded
//             separately): their one mat-vec runs on the vector unit (af_matvec_plain, round 4).
//   tip-tip   ops run INSIDE the list (round 4: always; PLLHIP_AA_TT_INSIDE=0 puts them ahead again as a launch of
//             k_aa_tt_rounds): the parent depends on the two tip characters only, so each op gathers ONE row per site
//             from a pair table that k_af_prepare makes per op (AfPairJob).  An inner-inner op over two tip-tip results
//             -- or a tip-inner op over one -- is a table lookup over character pairs as well (partials_aa_mfma.hip):
//             two rows, one multiplication, done per tile in the 16-bytes-per-lane layout of the stores.
//   reload    an operand that has no slot (written by an earlier call, or by a tip-tip op whose
//             reader is not a lookup, or evicted) is copied from HBM by LDS-DMA during the op
//             before its reader.
//   stores    a finished tile goes through a per-wave LDS stage into the 16-bytes-per-lane
//             layout: 5 KB contiguous per wave and op.
//
// The order of the list and the slots come from the planner of partials_fused.hip (depth
// first, heavier subtree first; Belady for evictions).
//
// Roofline: HBM writes, 644 B per site-update (640 B CLV + 4 B scaler count) + tip
// characters; the matrix cores run 80 MFMAs per 16 columns and child (16 cycles each):
// about 0.6 of the time the stores need.  DESIGN.md 2.2c has the budget and the measurements.
#include <algorithm>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>

#include "ctx.hpp"
#include "numerics.hpp"
#include "partials_fused.hpp"

#define PLL_LDS __attribute__((address_space(3)))
#define PLL_GLOBAL __attribute__((address_space(1)))

namespace
{
constexpr int AF_J = 2;                    // sub-tiles per wave
constexpr int AF_WS = 4 * AF_J;            // sites of a wave's tile
constexpr int AF_WGS = 4 * AF_WS;          // sites of a workgroup's tile (four waves)
constexpr int AF_TILE_B = AF_WS * 640;     // bytes of a wave's tile of a CLV
constexpr int AF_NSLOT = 5;                // values a wave keeps in registers
constexpr int AF_MAT_PIECES = 13;          // 1 KB pieces of a matrix block in operand order (12.5, padded)
constexpr int AF_MAT_B = AF_MAT_PIECES * 1024;
constexpr int AF_LDS_B = 3 * AF_MAT_B + 4 * 2 * AF_TILE_B + 16; // two workgroups per CU: 158 of its 160 KB

// One record per op, read through the scalar cache: everything the wave needs while the op runs.
struct AaRec
{
  unsigned long long parent;      // CLV the op writes
  unsigned long long pscaler;     // its scale buffer (0: none)
  unsigned long long tab_l;       // lookup: table of pair 1; tip-inner: the tip's table [code][rate][state]
  unsigned long long tab_r;       // lookup: table of pair 2
  unsigned long long row[4];      // tip rows OF THE OP AFTER NEXT: lookup (t1, t2), (t3, t4); tip-inner row[0]; else rows of zeros
  unsigned int xoff, yoff;        // byte offsets (operand order): left block OF THE NEXT OP, right block of this op
  unsigned int flags;             // AF_* below; bits 8-9: kind of the next op
  unsigned int slots;             // lslot | rslot << 4 | pslot << 8 | ra_slot << 12 | rb_slot << 16
  unsigned long long ra_src, ra_cnt; // reload A: done during this op for the NEXT op's left operand
  unsigned long long rb_src, rb_cnt; // reload B: the next op's right operand
  unsigned int pad[4];
};
static_assert(sizeof(AaRec) == 128, "thirty-two words per op");
constexpr unsigned int AF_KIND_MASK = 3u;   // 0 inner-inner, 1 tip-inner, 2 lookup
constexpr unsigned int AF_HAS_PSLOT = 4u;
constexpr unsigned int AF_SCALING = 8u;
constexpr unsigned int AF_LCNT = 16u;
constexpr unsigned int AF_RCNT = 32u;
constexpr unsigned int AF_RELOAD_A = 64u;
constexpr unsigned int AF_RELOAD_B = 128u;
constexpr unsigned int AF_SYNC_LEFT = 2048u;   // an inner-inner op behind a lookup: a barrier of its own ahead of the left products
constexpr unsigned int AF_ZERO_COUNTS = 1024u; // (bits 8-9: the next op's kind) a tip-tip op inside the list: never
                                               // scales, but clears its scale buffer (core_partials_avx.c:598-599)
constexpr unsigned int AF_ONE_TABLE = 4096u;   // a lookup with ONE table (round 4: a tip-tip op's pair table): nothing to multiply
// (round 5) The records of a segment are LINKED: bits 15.. of `flags` name the record of the op that runs next -- the
// last op's names op 0's again, AF_LAST set --, the header's `yoff` the segment's last record.  The kernel then
// carries neither an op count nor an op number through the op loop: with segments both would be values of the
// item instead of kernel arguments, two scalar registers more in a loop that has none to spare (the two records in
// flight are 56 of them).
constexpr unsigned int AF_TI_MFMA = 16384u;   // (opt-in, PLLHIP_AA_TI_MFMA=1) a tip-inner op whose mat-vec runs on the matrix cores
constexpr unsigned int AF_LAST = 8192u;
// (round 6) the scaling certificate: an op whose scaling test also looks for a largest entry within rounding distance of
// the threshold (partials_aa_fused_op.inc); the segment's HEADER record carries what the rare path needs -- `parent`:
// the address of the call's flag word (host memory), `pscaler`: the window as a double, 2^-256 x its relative half-width
constexpr unsigned int AF_CERT = 32768u;
// (round 6) a value that a later op of the SAME list copies back from HBM (the planner gave its slot away): stored with
// the default cache policy instead of the non-temporal one, so that the copy -- which the reading op waits for with
// everything else in flight -- comes out of L2 / the memory-side cache a few ops later instead of out of DRAM behind
// the write stream (profiles/r6_replay_counters_c3.txt: a list with three such copies sees twice the read latency)
constexpr unsigned int AF_KEEP = 65536u;
constexpr unsigned int AF_NEXT_SHIFT = 17u;

struct AfMatJob
{
  const double * src;           // [rate][row][column]
  unsigned long long dst_off;   // bytes into the operand-order buffer
  unsigned long long plain;     // 0: operand order of the matrix cores; 1: row order of the vector unit (af_matvec_plain)
};
struct AfTipJob
{
  const double * lmat;          // the tip's matrices
  unsigned long long dst_off;   // bytes into the tip-table buffer
};
// Round 4: a tip-tip op of the list as ONE gather.  parent = tip table of the left matrix [character 1] (.) tip table
// of the right matrix [character 2] depends on the two characters only: maxstates^2 rows of 640 bytes, tabulated per
// op by k_af_prepare with the very multiplication the list kernel did per site (same bits).  The op then gathers one
// row per site instead of two and multiplies nothing: half the gather instructions and LDS reads of a tip-tip op --
// and the texture path is what the lock-step of a workgroup's waves queues on.
struct AfPairJob
{
  const double * lmat, * rmat;
  unsigned long long dst_off;   // bytes into the pair-table buffer
};

// ---- per list: every matrix block of the list in operand order, every tip table
// Operand order: 25 blocks of 64 doubles, block (t, b), t = row group, b = 0..3 the first MFMA of
// chain b, b = 4 the source of the second; lane (q, rate, i) of block (t, b) holds
// P_rate[G_t(i)][4q + b] (b < 4) or P_rate[G_t(i)][16 + q], G_t(i) = 4i + t (t < 4), 16 + i:
// the accumulator of group t then puts state 4q + t (or 16 + q) in lane q.
// the entries of a 20-entry row selected by a state mask, added in ascending order (masksum_seq over registers: the
// others contribute +0.0, which changes no bit of a sum that starts at +0.0)
__device__ __forceinline__ double af_masksum20(const double (&r)[20], unsigned int mask)
{
  double a = 0.0;
#pragma unroll
  for (int jj = 0; jj < 20; ++jj) a += ((mask >> jj) & 1u) ? r[jj] : 0.0;
  return a;
}

constexpr unsigned int AF_PAIR_WGS = 4;  // workgroups per pair table: a range of first characters each
constexpr int AF_C1_MAX = 8;             // first characters of such a range at most (maxstates <= 32)
__global__ __launch_bounds__(256) void k_af_prepare(const AfMatJob * __restrict__ mj, unsigned int nmat,
                                                    const AfTipJob * __restrict__ tj, unsigned int ntip,
                                                    char * aorder, char * titab,
                                                    const unsigned int * __restrict__ tipmap, unsigned int ms,
                                                    unsigned int * __restrict__ tile_counter,
                                                    const AfPairJob * __restrict__ pj, unsigned int npair, char * pairtab,
                                                    const AaLookupJob * __restrict__ lj, unsigned int nlk)
{
  const unsigned int b = blockIdx.x;
  __shared__ double sh_child[32 * 80];        // a lookup table's child over (c1, every c2)
  __shared__ double sh_m0[80 * 21], sh_m1[80 * 21]; // a job's matrices, rows of 21 (a lane per row: 4-way conflicts)
  __shared__ double sh_l[AF_C1_MAX * 80];     // left factors of a job's first character(s)
  // a matrix [rate][row][column] = 80 rows of 20, coalesced from memory into rows of 21
  auto af_stage_rows = [&](const PLL_GLOBAL double * m, double * sh) __attribute__((always_inline)) {
    double v[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) v[k] = m[threadIdx.x + 256u * k < 1600u ? threadIdx.x + 256u * k : 0u];
#pragma unroll
    for (int k = 0; k < 7; ++k)
    {
      const unsigned int e = threadIdx.x + 256u * k;
      if (e < 1600u) sh[(e / 20u) * 21u + e % 20u] = v[k];
    }
  };
  // (the list kernel's tile counter: reset here instead of by a fill kernel of its own, round 4)
  if (b == 0 && threadIdx.x == 0) *tile_counter = 0u;
  // (round 6: every job asks for all it will read before it looks at any of it -- the rows of its matrices, and the
  // state masks of the up to eleven second characters c2 = g, g + 3, ... a thread walks (maxstates <= 32): a mask
  // fetched inside the walk was a round trip to L2 per character, eight in a row; profiles/r6_aa_prepare_ab.txt)
  constexpr int AF_C2_MAX = 11;
  if (b < nmat)
  {
    const PLL_GLOBAL double * src = (const PLL_GLOBAL double *)mj[b].src;
    double * out = reinterpret_cast<double *>(aorder + mj[b].dst_off);
    const bool plain = mj[b].plain != 0;
    constexpr int AF_MAT_STEPS = (AF_MAT_B / 8 + 255) / 256;
    double v[AF_MAT_STEPS];
#pragma unroll
    for (int k = 0; k < AF_MAT_STEPS; ++k)
    {
      const unsigned int idx = threadIdx.x + 256u * k;
      unsigned int from = 0u;
      if (!plain)
      {
        const unsigned int blk = idx >> 6, lane = idx & 63u, t = blk / 5, bb = blk - 5 * t;
        const unsigned int q = lane >> 4, i = lane & 3u;
        unsigned int rate = (lane >> 2) & 3u;
        unsigned int row = t < 4 ? 4 * i + t : 16 + i, col = bb < 4 ? 4 * q + bb : 16 + q;
        if (bb == 4)
        {
          // (round 4) the fifth step of the four chains -- columns 16..19 -- runs on the vector unit (af_group): lane
          // class (q, rate) reads ITS output row's four entries as 32 contiguous bytes: entry (4 (4q + rate) + m)
          const unsigned int cls = lane >> 2, m = lane & 3u, cq = cls >> 2;
          rate = cls & 3u;
          row = t < 4 ? 4 * cq + t : 16 + cq;
          col = 16 + m;
        }
        from = rate * 400 + row * 20 + col;
      }
      else
      {
        // Row order (the right block of a tip-inner op, af_matvec_plain): 50 pieces of 16 x 16 bytes; piece
        // (pass, t, c) holds, for lane class (q, rate), columns 4c + 2 pass and 4c + 2 pass + 1 of the row whose
        // output lane q keeps in x[.][t] -- state 4q + t, or 16 + q for t = 4.
        const unsigned int e = idx >> 1, half = idx & 1u, piece = e >> 4, cls = e & 15u;
        const unsigned int q = cls >> 2, rate = cls & 3u, pass = piece / 25, rem = piece - 25 * pass, t = rem / 5, cc = rem - 5 * t;
        const unsigned int row = t < 4 ? 4 * q + t : 16 + q, col = 4 * cc + 2 * pass + half;
        from = rate * 400 + row * 20 + col;
      }
      v[k] = src[idx < 1600 ? from : 0u]; // (unconditional: the seven requests leave together)
    }
#pragma unroll
    for (int k = 0; k < AF_MAT_STEPS; ++k)
    {
      const unsigned int idx = threadIdx.x + 256u * k;
      if (idx < AF_MAT_B / 8) out[idx] = idx < 1600 ? v[k] : 0.0;
    }
  }
  else if (b - nmat < ntip)
  {
    // tab[code][k][i] = sum_{j in tipmap[code]} P[k][i][j] (core_partials_avx.c:1140-1177), as k_aa_tip_tables: the
    // row's selected entries added in ascending order (adding +0.0 for the others changes no bit)
    double * out = reinterpret_cast<double *>(titab + tj[b - nmat].dst_off);
    if (threadIdx.x < 240)
    {
      const unsigned int ki = threadIdx.x % 80u, g = threadIdx.x / 80u;
      const PLL_GLOBAL double * lm = (const PLL_GLOBAL double *)tj[b - nmat].lmat + ki * 20;
      double r[20];
      unsigned int msk[AF_C2_MAX];
#pragma unroll
      for (int jj = 0; jj < 20; ++jj) r[jj] = lm[jj];
#pragma unroll
      for (int k = 0; k < AF_C2_MAX; ++k) msk[k] = g + 3u * k < ms ? tipmap[g + 3u * k] : 0u;
#pragma unroll
      for (int k = 0; k < AF_C2_MAX; ++k)
        if (g + 3u * k < ms) out[(g + 3u * k) * 80 + ki] = af_masksum20(r, msk[k]);
    }
  }
  else if (b - nmat - ntip < npair * AF_PAIR_WGS)
  {
    // A tip-tip op's pair table: row (c1 ms + c2) = left factor of c1 (.) right factor of c2 -- the two masked row
    // sums of k_aa_tip_tables / the branch above and the ONE multiplication of k_aa_tt_rounds (a masked sum = the
    // row's selected entries added in ascending order; adding +0.0 for the others changes no bit).
    // Round 6, third form (profiles/r6_aa_prepare_ab.txt; tools/aa_prepare_parts.sh says what each kind of job costs):
    // a workgroup per (op, first character) that pulled its rows of both matrices straight into registers was bound
    // by those loads -- a lane per row is sixty-four cache lines per instruction, 23 workgroups per op each fetching
    // all of it.  Now AF_PAIR_WGS workgroups per op, each a range of first characters: the right matrix comes ONCE,
    // coalesced, through LDS (rows of 21: a lane per row is then a 4-way bank conflict, not 16-way), the right factors
    // of a thread's second characters are formed once and serve every first character of the range.
    const unsigned int job = (b - nmat - ntip) / AF_PAIR_WGS, part = (b - nmat - ntip) - job * AF_PAIR_WGS;
    const unsigned int per = (ms + AF_PAIR_WGS - 1u) / AF_PAIR_WGS, c1_0 = part * per; // (per <= AF_C1_MAX: ms <= 32)
    const AfPairJob & j = pj[job];
    af_stage_rows((const PLL_GLOBAL double *)j.rmat, sh_m0);
    unsigned int msk[AF_C2_MAX];
    const unsigned int ki = threadIdx.x % 80u, g = threadIdx.x / 80u;
#pragma unroll
    for (int k = 0; k < AF_C2_MAX; ++k) msk[k] = g + 3u * k < ms ? tipmap[g + 3u * k] : 0u;
    if (threadIdx.x < 80)
    {
      // the left factors of the range's first characters (one thread per row: its 160 bytes straight from memory)
      const PLL_GLOBAL double * lm = (const PLL_GLOBAL double *)j.lmat + ki * 20;
      double lrow[20];
#pragma unroll
      for (int jj = 0; jj < 20; ++jj) lrow[jj] = lm[jj];
#pragma unroll
      for (int t = 0; t < AF_C1_MAX; ++t)
        if (t < per && c1_0 + t < ms) sh_l[t * 80 + ki] = af_masksum20(lrow, tipmap[c1_0 + t]);
    }
    __syncthreads();
    if (threadIdx.x < 240)
    {
      double r[20], l[AF_C1_MAX];
#pragma unroll
      for (int jj = 0; jj < 20; ++jj) r[jj] = sh_m0[ki * 21 + jj];
#pragma unroll
      for (int t = 0; t < AF_C1_MAX; ++t) l[t] = (t < per && c1_0 + t < ms) ? sh_l[t * 80 + ki] : 0.0;
      double * out = reinterpret_cast<double *>(pairtab + j.dst_off);
#pragma unroll
      for (int k = 0; k < AF_C2_MAX; ++k)
        if (g + 3u * k < ms)
        {
          const double rf = af_masksum20(r, msk[k]);
#pragma unroll
          for (int t = 0; t < AF_C1_MAX; ++t)
            if (t < per && c1_0 + t < ms) out[((size_t)(c1_0 + t) * ms + g + 3u * k) * 80 + ki] = l[t] * rf;
        }
    }
  }
  else if (b - nmat - ntip - npair * AF_PAIR_WGS < nlk * ms)
  {
    // (round 4) a lookup op's table, one workgroup per (table, character 1): rows (c1 ms + c2) = P x child over the
    // character pairs the child -- a tip-tip result -- can be: the child as the branch above makes it, then the
    // mat-vec in the order of the kernel the op would have run (AaLookupJob, ctx.hpp).  Until then six launches of
    // those kernels ahead of every list (tip tables, tip-tip over all pairs, inner-inner x "ones"): 60 us.
    // Round 6: the child's right matrix and P come coalesced through LDS (rows of 21), the left factor of c1 from the
    // first 80 threads (a row each, straight from memory, while the others stage); LDS: 27 KB of matrices + the
    // children, three workgroups per CU; two barriers.
    const unsigned int job = (b - nmat - ntip - npair * AF_PAIR_WGS) / ms, c1 = (b - nmat - ntip - npair * AF_PAIR_WGS) - job * ms;
    const AaLookupJob & j = lj[job];
    double * out = j.dst + (size_t)c1 * ms * 80;
    if (j.mode == 2u)
    {
      // the tip's own factor: row (c1, 0) is its table's row c1 (the op's second character row is all zeros)
      if (threadIdx.x < 80) out[threadIdx.x] = masksum_seq(j.kl + (size_t)threadIdx.x * 20, tipmap[c1], 20);
      return;
    }
    af_stage_rows((const PLL_GLOBAL double *)j.kr, sh_m0);
    af_stage_rows((const PLL_GLOBAL double *)j.pm, sh_m1);
    const unsigned int ki = threadIdx.x % 80u, g = threadIdx.x / 80u; // (threads 240..255: no row)
    unsigned int msk[AF_C2_MAX];
#pragma unroll
    for (int k = 0; k < AF_C2_MAX; ++k) msk[k] = g + 3u * k < ms ? tipmap[g + 3u * k] : 0u;
    if (threadIdx.x < 80)
    {
      const PLL_GLOBAL double * lm = (const PLL_GLOBAL double *)j.kl + ki * 20;
      double lrow[20];
#pragma unroll
      for (int jj = 0; jj < 20; ++jj) lrow[jj] = lm[jj];
      sh_l[ki] = af_masksum20(lrow, tipmap[c1]);
    }
    __syncthreads();
    double prow[20];
    if (threadIdx.x < 240)
    {
      double r[20];
#pragma unroll
      for (int jj = 0; jj < 20; ++jj) r[jj] = sh_m0[ki * 21 + jj];
      const double l = sh_l[ki];
#pragma unroll
      for (int k = 0; k < AF_C2_MAX; ++k)
        if (g + 3u * k < ms) sh_child[(g + 3u * k) * 80 + ki] = l * af_masksum20(r, msk[k]);
#pragma unroll
      for (int jj = 0; jj < 20; ++jj) prow[jj] = sh_m1[ki * 21 + jj];
    }
    __syncthreads();
    if (threadIdx.x < 240)
    {
      // the thread's row of P in registers, the children of its second characters out of LDS
      const unsigned int kk = ki / 20u;
      const bool fused = j.mode == 0u;
#pragma unroll
      for (int k = 0; k < AF_C2_MAX; ++k)
        if (g + 3u * k < ms)
        {
          const double * v = sh_child + (g + 3u * k) * 80 + kk * 20;
          out[(g + 3u * k) * 80 + ki] = fused ? dot_strided4<true>(prow, v, 20u) : dot_strided4<false>(prow, v, 20u);
        }
    }
  }
}

#ifdef PLLHIP_AF_TIMING
// (tool build: PLLHIP_AF_EXP=mask switches parts of the kernel off -- wrong results, for timing only:
// 1 no matrix-core products, 2 no stores, 4 no gathers, 8 no block staging, 16 no barriers, 32 no wait for the
// operands copied back from HBM, 64 no such copies at all)
__device__ unsigned int af_exp_mask;
#define AF_EXP(bit) (exp_mask & (bit))
#else
#define AF_EXP(bit) false
#endif

// ---- device helpers
typedef const unsigned int __attribute__((address_space(4))) * af_words;
// words [FIRST, FIRST + N) of record i, through the scalar cache
template <int FIRST, int N>
struct AfW
{
  unsigned int w[N];
  __device__ __forceinline__ unsigned int operator[](int t) const { return w[t - FIRST]; }
  __device__ __forceinline__ unsigned long long quad(int t) const
  {
    return (unsigned long long)w[t - FIRST] | ((unsigned long long)w[t - FIRST + 1] << 32);
  }
};
template <int FIRST, int N>
__device__ __forceinline__ AfW<FIRST, N> af_load(const AaRec * plan, unsigned int i)
{
  const af_words p = (af_words)(unsigned long long)(plan + i) + FIRST;
  AfW<FIRST, N> r;
#pragma unroll
  for (int t = 0; t < N; ++t) r.w[t] = p[t];
  return r;
}

// Every global access of the kernel is "wave-uniform 64-bit base (SGPRs) + 32-bit lane offset": one
// VGPR per address instead of two, and nothing per-lane to keep (or spill) across the op loop.
typedef char PLL_GLOBAL * af_gptr;
__device__ __forceinline__ af_gptr af_base(unsigned long long uniform_address)
{
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)uniform_address);
  const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(uniform_address >> 32));
  return (af_gptr)(((unsigned long long)hi << 32) | lo);
}

// 64 lanes x 16 bytes from global memory straight into LDS (lane l lands at lds_b + 16 l).  Inline
// assembly on purpose: the compiler neither counts it nor waits for it; the waits are ours.  The
// instruction's immediate offset moves BOTH addresses (tools/dma_offset_probe.hip), so N consecutive
// KiB cost one M0 and N instructions.
// M0 (the LDS address of an LDS-DMA) is a register the compiler reserves for itself, so an asm block may not simply
// declare it clobbered: it is saved and put back around every run of DMA instructions -- two scalar instructions per
// run, ~30 of the ~165 a wave executes per op.  The compiler never reads M0 in these kernels (tools/check_agprs.py
// looks: no instruction outside these blocks mentions m0), so the assembly could simply OWN it
// (-DPLLHIP_AF_SAVE_M0=0).  Measured late in round 4, three interleaved pairs on one box: C3 1.713-1.724 against
// 1.710-1.713 ms, 200-taxon random tree 3.989 against 3.981 -- the scalar unit is not on the waves' critical path;
// not worth leaning on a reserved register: the default saves.
#ifndef PLLHIP_AF_PIPE
#define PLLHIP_AF_PIPE 0
#endif
#ifndef PLLHIP_AF_SAVE_M0
#define PLLHIP_AF_SAVE_M0 1
#endif
template <int N>
__device__ __forceinline__ void af_dma_run(unsigned int lds_b, unsigned long long uniform_src, unsigned int lane16)
{
  static_assert(N == 1 || N == 4, "immediate offsets reach 4095");
#if PLLHIP_AF_SAVE_M0
  unsigned int m0_saved;
  if (N == 1)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_saved) : "s"(lds_b), "v"(lane16), "s"(uniform_src) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
                 "global_load_lds_dwordx4 %2, %3 offset:1024\n\tglobal_load_lds_dwordx4 %2, %3 offset:2048\n\t"
                 "global_load_lds_dwordx4 %2, %3 offset:3072\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_saved) : "s"(lds_b), "v"(lane16), "s"(uniform_src) : "memory");
#else
  if (N == 1)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 : : "s"(lds_b), "v"(lane16), "s"(uniform_src) : "memory");
  else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:3072"
                 : : "s"(lds_b), "v"(lane16), "s"(uniform_src) : "memory");
#endif
}
// a gather: five KiB of LDS from five lane offsets each (voff[it] relative to table); the first four
// share an M0 -- the immediate offset that advances the LDS address advances the global one as well,
// which a base moved back by 3 KiB and lane offsets moved forward by (3 - it) KiB undo
__device__ __forceinline__ void af_dma_gather5(unsigned int lds_b, unsigned long long table, const unsigned int (&voff)[5])
{
  const unsigned int v0 = voff[0] + 3072u, v1 = voff[1] + 2048u, v2 = voff[2] + 1024u;
#if PLLHIP_AF_SAVE_M0
  unsigned int m0_saved;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %8\n\t"
               "global_load_lds_dwordx4 %4, %8 offset:1024\n\tglobal_load_lds_dwordx4 %5, %8 offset:2048\n\t"
               "global_load_lds_dwordx4 %6, %8 offset:3072\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %7, %9\n\ts_mov_b32 m0, %0"
               : "=&s"(m0_saved)
               : "s"(lds_b), "s"(lds_b + 4096u), "v"(v0), "v"(v1), "v"(v2), "v"(voff[3]), "v"(voff[4]), "s"(table - 3072ull), "s"(table)
               : "memory");
#else
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %7\n\t"
               "global_load_lds_dwordx4 %3, %7 offset:1024\n\tglobal_load_lds_dwordx4 %4, %7 offset:2048\n\t"
               "global_load_lds_dwordx4 %5, %7 offset:3072\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %6, %8"
               :
               : "s"(lds_b), "s"(lds_b + 4096u), "v"(v0), "v"(v1), "v"(v2), "v"(voff[3]), "v"(voff[4]), "s"(table - 3072ull), "s"(table)
               : "memory");
#endif
}

struct AfSlot
{
  double v[AF_J][5];
  unsigned int c[AF_J];
};

// x[j][t] = (MUL ? x[j][t] : 1) * (P . column)[state 4q + t | 16 + q], reference order;
// mat_lane: the block in LDS (operand order) + 8 lane.  Software-pipelined by hand: the A operands of
// row group t + 1 are fetched before the sixteen MFMAs of group t are issued, and the three adds of
// group t wait until those of group t + 1 are in the pipe -- a wave has nobody to hide its LDS and
// matrix-core latencies behind but itself (two waves per SIMD).
struct AfAops
{
  double a1[4];   // A operands of the four chains' first MFMA (k-chunks {m, m+4, m+8, m+12})
  double2 a5[2];  // the lane's own output row, columns 16..19: the chains' fifth step, on the vector unit
};
__device__ __forceinline__ void af_fetch_a(const char * mat_lane, const char * mat_cls, int t, AfAops & a)
{
#pragma unroll
  for (int m = 0; m < 4; ++m) a.a1[m] = *reinterpret_cast<const double *>(mat_lane + (t * 5 + m) * 512);
  a.a5[0] = *reinterpret_cast<const double2 *>(mat_cls + (t * 5 + 4) * 512);
  a.a5[1] = *reinterpret_cast<const double2 *>(mat_cls + (t * 5 + 4) * 512 + 16);
}
// Round 4: the fifth step of a chain (state 16 + m) was a second MFMA whose A operand is zero except in k-slot m --
// half of all the matrix-core instructions of the kernel, 16 cycles each, for one useful product out of sixteen.
// It is one FMA on the vector unit now: acc[m] = fma(P[row][16 + m], column[16 + m], acc[m]) -- the very operation the
// zero-padded MFMA performed (its other three steps added +0 products: tools/mfma_order_probe.hip), same bits.
// c5[j][m] = entry 16 + m of the lane's column: held by lane q' = m of the column (B operand word 4), fetched once
// per child through the LDS crossbar (af_column_tail).
__device__ __forceinline__ void af_group(const AfAops & a, const double (&b)[AF_J][5], const double (&c5)[AF_J][4],
                                         double (&acc)[AF_J][4])
{
#pragma unroll
  for (int j = 0; j < AF_J; ++j)
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[j][m] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.a1[m], b[j][m], 0.0, 0, 0, 0);
  const double a5[4] = {a.a5[0].x, a.a5[0].y, a.a5[1].x, a.a5[1].y};
#pragma unroll
  for (int j = 0; j < AF_J; ++j)
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[j][m] = fma(a5[m], c5[j][m], acc[j][m]);
}
__device__ __forceinline__ void af_column_tail(const double (&b)[AF_J][5], unsigned int lane, double (&c5)[AF_J][4])
{
#pragma unroll
  for (int m = 0; m < 4; ++m)
  {
    const unsigned int src = ((lane & 15u) + 16u * m) * 4u;
#pragma unroll
    for (int j = 0; j < AF_J; ++j)
    {
      const int lo = __builtin_amdgcn_ds_bpermute((int)src, __double2loint(b[j][4]));
      const int hi = __builtin_amdgcn_ds_bpermute((int)src, __double2hiint(b[j][4]));
      c5[j][m] = __hiloint2double(hi, lo);
    }
  }
}
template <bool MUL>
__device__ __forceinline__ void af_sum(const double (&acc)[AF_J][4], int t, double (&x)[AF_J][5])
{
#pragma unroll
  for (int j = 0; j < AF_J; ++j)
  {
    const double sum = (acc[j][0] + acc[j][1]) + (acc[j][2] + acc[j][3]);
    x[j][t] = MUL ? x[j][t] * sum : sum;
    // (the value is wanted here, see rate_matvec_chain in aa_mfma.hpp)
    asm volatile("" : "+v"(x[j][t]));
  }
}
template <bool MUL>
__device__ __forceinline__ void af_matvec(const char * mat_lane, const char * mat_cls, unsigned int lane,
                                          const double (&b)[AF_J][5], double (&x)[AF_J][5])
{
  // One row group at a time: its A operands, eight MFMAs, eight FMAs, the sums.  (Rounds 3's version fetched the
  // operands of group t + 1 ahead of the MFMAs of group t and deferred the sums -- worth 1 % then, and 32 registers,
  // which the column tail and the wider fifth-step operands need now: the kernel must stay within 128, see the slots.)
  double c5[AF_J][4];
#if (PLLHIP_AF_PIPE & 1)
  // (round 5, PLLHIP_AF_PIPE bit 0: the A operands of row group t + 1 requested before the eight MFMAs of group t are
  // issued -- with eight MFMAs per group instead of round 3's sixteen the LDS round trip is a third of a group's time)
  AfAops a_cur, a_nxt;
  af_fetch_a(mat_lane, mat_cls, 0, a_cur);
  af_column_tail(b, lane, c5);
#pragma unroll
  for (int t = 0; t < 5; ++t)
  {
    double acc[AF_J][4];
    if (t < 4) af_fetch_a(mat_lane, mat_cls, t + 1, a_nxt);
    af_group(a_cur, b, c5, acc);
    af_sum<MUL>(acc, t, x);
    if (t < 4) a_cur = a_nxt;
    __builtin_amdgcn_sched_barrier(0);
  }
#else
  af_column_tail(b, lane, c5);
#pragma unroll
  for (int t = 0; t < 5; ++t)
  {
    AfAops a;
    double acc[AF_J][4];
    af_fetch_a(mat_lane, mat_cls, t, a);
    af_group(a, b, c5, acc);
    af_sum<MUL>(acc, t, x);
    __builtin_amdgcn_sched_barrier(0);
  }
#endif
}

// x[j][t] = tip factor * (P . column)[state 4q + t | 16 + q] in the order of the reference's TIP-INNER kernel
// (core_partials_avx.c:1229-1284; reached under the AVX2 flag too, core_partials.c:427-441): the four chains
// strided by j mod 4 and the pairwise tree, every step a multiplication and THEN an addition.  The matrix
// cores cannot round a product on its own, so a tip-inner op's one mat-vec runs on the vector unit:
//   * the child (b, operand layout: five of a column's twenty entries per lane) goes through the wave's stage 0
//     in the layout of a CLV, and every lane reads its column back, ten entries at a time (chains 0 and 1,
//     then chains 2 and 3: 40 registers instead of 80 for the two sub-tiles);
//   * the block is staged in ROW order (k_af_prepare): one 16-byte read gives two columns of the lane's row,
//     the four lanes of a site group read the same address, the sixteen (q, rate) classes 256 contiguous bytes;
//     every entry is read once and serves both sub-tiles;
//   * the tip's factor stays where it was gathered to (stage 1) until the very multiplication that needs it:
//     held in registers from the top of the op it cost 20 of the 128, and the kernel spilled;
//   * per lane 200 multiplications + 230 additions + 10: ~1800 issue cycles per wave and op where the matrix
//     cores need 1280 -- and 70 KB of LDS traffic per wave and op (50 block + 20 column) against 26.
// stage_col: stage 0 + the lane's column (n, rate); stage_b / stage_b5: stage 0 + the lane's own entries
// (af_read_tile); tip_b / tip_b5: the same places in stage 1; ymat: the block + 16 (4q + rate).
__device__ __forceinline__ void af_matvec_plain(char * stage_b, char * stage_b5, const char * stage_col, const char * ymat,
                                                const char * tip_b, const char * tip_b5, const double (&b)[AF_J][5],
                                                double (&x)[AF_J][5])
{
#pragma unroll
  for (int j = 0; j < AF_J; ++j)
  {
    *reinterpret_cast<double2 *>(stage_b + j * 2560) = make_double2(b[j][0], b[j][1]);
    *reinterpret_cast<double2 *>(stage_b + j * 2560 + 16) = make_double2(b[j][2], b[j][3]);
    *reinterpret_cast<double *>(stage_b5 + j * 2560) = b[j][4];
  }
  asm volatile("" ::: "memory"); // (LDS operations of a wave execute in order; the compiler reasons per thread)
  // The entries of output row t + 1 are requested as those of row t are used up, each into the register its
  // predecessor has just left (the reads return in order; a compiler barrier per entry keeps them where they are
  // written: left alone the compiler fetches all five rows ahead, a hundred registers -- or none ahead, and every row
  // waits for LDS): the row's latency hides behind the previous row's forty operations.
  auto row_entry = [&](int pass, int t, int cc) __attribute__((always_inline)) {
    return *reinterpret_cast<const double2 *>(ymat + ((pass * 25 + t * 5 + cc) * 256));
  };
  double2 p[5];
#pragma unroll
  for (int cc = 0; cc < 5; ++cc) p[cc] = row_entry(0, 0, cc);
#pragma unroll
  for (int pass = 0; pass < 2; ++pass)
  {
    double2 c[AF_J][5];
#pragma unroll
    for (int j = 0; j < AF_J; ++j)
#pragma unroll
      for (int cc = 0; cc < 5; ++cc) c[j][cc] = *reinterpret_cast<const double2 *>(stage_col + j * 2560 + cc * 32 + pass * 16);
#pragma unroll
    for (int t = 0; t < 5; ++t)
    {
      double a0[AF_J], a1[AF_J];
#pragma unroll
      for (int j = 0; j < AF_J; ++j) a0[j] = a1[j] = 0.0;
#pragma unroll
      for (int cc = 0; cc < 5; ++cc)
      {
#pragma unroll
        for (int j = 0; j < AF_J; ++j)
        {
          a0[j] = a0[j] + p[cc].x * c[j][cc].x;
          a1[j] = a1[j] + p[cc].y * c[j][cc].y;
        }
        asm volatile("" : "+v"(a0[0]), "+v"(a1[0]), "+v"(a0[1]), "+v"(a1[1])); // (this entry is used up HERE)
        if (t < 4) p[cc] = row_entry(pass, t + 1, cc);
        else if (pass == 0) p[cc] = row_entry(1, 0, cc);
        asm volatile("" ::: "memory");
      }
#pragma unroll
      for (int j = 0; j < AF_J; ++j)
      {
        const double pair = a0[j] + a1[j];
        if (pass == 0) x[j][t] = pair;
        else
        {
          const double tipf = *reinterpret_cast<const double *>((t < 4 ? tip_b + 8 * t : tip_b5) + j * 2560);
          x[j][t] = tipf * (x[j][t] + pair);
        }
        asm volatile("" : "+v"(x[j][t])); // (wanted HERE: left alone the optimiser sinks the sums to where x is used)
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("" ::: "memory");
}

// a tile in the lane layout of the MFMA operands, out of a stage in LDS
__device__ __forceinline__ void af_read_tile(const char * stage_lane, const char * stage_lane5, double (&v)[AF_J][5])
{
#pragma unroll
  for (int j = 0; j < AF_J; ++j)
  {
    const double2 v0 = *reinterpret_cast<const double2 *>(stage_lane + j * 2560);
    const double2 v1 = *reinterpret_cast<const double2 *>(stage_lane + j * 2560 + 16);
    v[j][0] = v0.x; v[j][1] = v0.y; v[j][2] = v1.x; v[j][3] = v1.y;
    v[j][4] = *reinterpret_cast<const double *>(stage_lane5 + j * 2560);
  }
}

// The five slots live in the ACCUMULATION registers a0..a109 and are touched by inline assembly
// only: slot K, sub-tile j, word w (five doubles as lo/hi words, then the count) is a[22 K + 11 j + w].
// The compiler never sees a slot: it learns from the clobber lists that these registers are taken
// (it then splits the wave's 256 registers into 128 + 128) and keeps everything of its own in v0..v127.
//
// How it came to this (each step was built and looked at in the ISA):
//  - slots as structs selected by a C++ switch: the optimiser merged the arms into one access through
//    a pointer that depends on the slot number -- the slots became 448 bytes of scratch per lane;
//  - sixty scalar variables and switches: all sixty values threaded through every arm of every switch
//    as phi nodes, 1150 register moves, one slot in scratch;
//  - the switches hidden in inline asm, slot variables as "+v" operands, then pinned to v146..v255:
//    the register allocator treats a pinned operand as pinned AT the asm only, copied the slots to
//    other registers in between and spilled them wholesale (590 spills) once the op body was there
//    twice (the loop is unrolled by two so that no loaded value is ever copied).
// A slot number of 15 matches no slot: "no slot" needs no branch around the access.

#define AF_SLOT_READ_ASM_0 \
  "s_cmp_eq_u32 %11, 0\n\ts_cbranch_scc1 .Laf_r0_%=\n\t" \
  "s_cmp_eq_u32 %11, 1\n\ts_cbranch_scc1 .Laf_r1_%=\n\t" \
  "s_cmp_eq_u32 %11, 2\n\ts_cbranch_scc1 .Laf_r2_%=\n\t" \
  "s_cmp_eq_u32 %11, 3\n\ts_cbranch_scc1 .Laf_r3_%=\n\t" \
  ".Laf_r4_%=:\n\tv_accvgpr_read_b32 %0, a88\n\tv_accvgpr_read_b32 %1, a89\n\tv_accvgpr_read_b32 %2, a90\n\tv_accvgpr_read_b32 %3, a91\n\tv_accvgpr_read_b32 %4, a92\n\tv_accvgpr_read_b32 %5, a93\n\tv_accvgpr_read_b32 %6, a94\n\tv_accvgpr_read_b32 %7, a95\n\tv_accvgpr_read_b32 %8, a96\n\tv_accvgpr_read_b32 %9, a97\n\tv_accvgpr_read_b32 %10, a98\n\ts_branch .Laf_re_%=\n" \
  ".Laf_r3_%=:\n\tv_accvgpr_read_b32 %0, a66\n\tv_accvgpr_read_b32 %1, a67\n\tv_accvgpr_read_b32 %2, a68\n\tv_accvgpr_read_b32 %3, a69\n\tv_accvgpr_read_b32 %4, a70\n\tv_accvgpr_read_b32 %5, a71\n\tv_accvgpr_read_b32 %6, a72\n\tv_accvgpr_read_b32 %7, a73\n\tv_accvgpr_read_b32 %8, a74\n\tv_accvgpr_read_b32 %9, a75\n\tv_accvgpr_read_b32 %10, a76\n\ts_branch .Laf_re_%=\n" \
  ".Laf_r2_%=:\n\tv_accvgpr_read_b32 %0, a44\n\tv_accvgpr_read_b32 %1, a45\n\tv_accvgpr_read_b32 %2, a46\n\tv_accvgpr_read_b32 %3, a47\n\tv_accvgpr_read_b32 %4, a48\n\tv_accvgpr_read_b32 %5, a49\n\tv_accvgpr_read_b32 %6, a50\n\tv_accvgpr_read_b32 %7, a51\n\tv_accvgpr_read_b32 %8, a52\n\tv_accvgpr_read_b32 %9, a53\n\tv_accvgpr_read_b32 %10, a54\n\ts_branch .Laf_re_%=\n" \
  ".Laf_r1_%=:\n\tv_accvgpr_read_b32 %0, a22\n\tv_accvgpr_read_b32 %1, a23\n\tv_accvgpr_read_b32 %2, a24\n\tv_accvgpr_read_b32 %3, a25\n\tv_accvgpr_read_b32 %4, a26\n\tv_accvgpr_read_b32 %5, a27\n\tv_accvgpr_read_b32 %6, a28\n\tv_accvgpr_read_b32 %7, a29\n\tv_accvgpr_read_b32 %8, a30\n\tv_accvgpr_read_b32 %9, a31\n\tv_accvgpr_read_b32 %10, a32\n\ts_branch .Laf_re_%=\n" \
  ".Laf_r0_%=:\n\tv_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1\n\tv_accvgpr_read_b32 %2, a2\n\tv_accvgpr_read_b32 %3, a3\n\tv_accvgpr_read_b32 %4, a4\n\tv_accvgpr_read_b32 %5, a5\n\tv_accvgpr_read_b32 %6, a6\n\tv_accvgpr_read_b32 %7, a7\n\tv_accvgpr_read_b32 %8, a8\n\tv_accvgpr_read_b32 %9, a9\n\tv_accvgpr_read_b32 %10, a10\n\t" \
  ".Laf_re_%=:"
#define AF_SLOT_WRITE_ASM_0 \
  "s_cmp_lg_u32 %11, 0\n\ts_cbranch_scc1 .Laf_w0_%=\n\tv_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %1\n\tv_accvgpr_write_b32 a2, %2\n\tv_accvgpr_write_b32 a3, %3\n\tv_accvgpr_write_b32 a4, %4\n\tv_accvgpr_write_b32 a5, %5\n\tv_accvgpr_write_b32 a6, %6\n\tv_accvgpr_write_b32 a7, %7\n\tv_accvgpr_write_b32 a8, %8\n\tv_accvgpr_write_b32 a9, %9\n\tv_accvgpr_write_b32 a10, %10\n\t.Laf_w0_%=:\n\t" \
  "s_cmp_lg_u32 %11, 1\n\ts_cbranch_scc1 .Laf_w1_%=\n\tv_accvgpr_write_b32 a22, %0\n\tv_accvgpr_write_b32 a23, %1\n\tv_accvgpr_write_b32 a24, %2\n\tv_accvgpr_write_b32 a25, %3\n\tv_accvgpr_write_b32 a26, %4\n\tv_accvgpr_write_b32 a27, %5\n\tv_accvgpr_write_b32 a28, %6\n\tv_accvgpr_write_b32 a29, %7\n\tv_accvgpr_write_b32 a30, %8\n\tv_accvgpr_write_b32 a31, %9\n\tv_accvgpr_write_b32 a32, %10\n\t.Laf_w1_%=:\n\t" \
  "s_cmp_lg_u32 %11, 2\n\ts_cbranch_scc1 .Laf_w2_%=\n\tv_accvgpr_write_b32 a44, %0\n\tv_accvgpr_write_b32 a45, %1\n\tv_accvgpr_write_b32 a46, %2\n\tv_accvgpr_write_b32 a47, %3\n\tv_accvgpr_write_b32 a48, %4\n\tv_accvgpr_write_b32 a49, %5\n\tv_accvgpr_write_b32 a50, %6\n\tv_accvgpr_write_b32 a51, %7\n\tv_accvgpr_write_b32 a52, %8\n\tv_accvgpr_write_b32 a53, %9\n\tv_accvgpr_write_b32 a54, %10\n\t.Laf_w2_%=:\n\t" \
  "s_cmp_lg_u32 %11, 3\n\ts_cbranch_scc1 .Laf_w3_%=\n\tv_accvgpr_write_b32 a66, %0\n\tv_accvgpr_write_b32 a67, %1\n\tv_accvgpr_write_b32 a68, %2\n\tv_accvgpr_write_b32 a69, %3\n\tv_accvgpr_write_b32 a70, %4\n\tv_accvgpr_write_b32 a71, %5\n\tv_accvgpr_write_b32 a72, %6\n\tv_accvgpr_write_b32 a73, %7\n\tv_accvgpr_write_b32 a74, %8\n\tv_accvgpr_write_b32 a75, %9\n\tv_accvgpr_write_b32 a76, %10\n\t.Laf_w3_%=:\n\t" \
  "s_cmp_lg_u32 %11, 4\n\ts_cbranch_scc1 .Laf_w4_%=\n\tv_accvgpr_write_b32 a88, %0\n\tv_accvgpr_write_b32 a89, %1\n\tv_accvgpr_write_b32 a90, %2\n\tv_accvgpr_write_b32 a91, %3\n\tv_accvgpr_write_b32 a92, %4\n\tv_accvgpr_write_b32 a93, %5\n\tv_accvgpr_write_b32 a94, %6\n\tv_accvgpr_write_b32 a95, %7\n\tv_accvgpr_write_b32 a96, %8\n\tv_accvgpr_write_b32 a97, %9\n\tv_accvgpr_write_b32 a98, %10\n\t.Laf_w4_%=:\n\t"
#define AF_SLOT_CLOBBER_0 "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98"
#define AF_SLOT_READ_ASM_1 \
  "s_cmp_eq_u32 %11, 0\n\ts_cbranch_scc1 .Laf_r0_%=\n\t" \
  "s_cmp_eq_u32 %11, 1\n\ts_cbranch_scc1 .Laf_r1_%=\n\t" \
  "s_cmp_eq_u32 %11, 2\n\ts_cbranch_scc1 .Laf_r2_%=\n\t" \
  "s_cmp_eq_u32 %11, 3\n\ts_cbranch_scc1 .Laf_r3_%=\n\t" \
  ".Laf_r4_%=:\n\tv_accvgpr_read_b32 %0, a99\n\tv_accvgpr_read_b32 %1, a100\n\tv_accvgpr_read_b32 %2, a101\n\tv_accvgpr_read_b32 %3, a102\n\tv_accvgpr_read_b32 %4, a103\n\tv_accvgpr_read_b32 %5, a104\n\tv_accvgpr_read_b32 %6, a105\n\tv_accvgpr_read_b32 %7, a106\n\tv_accvgpr_read_b32 %8, a107\n\tv_accvgpr_read_b32 %9, a108\n\tv_accvgpr_read_b32 %10, a109\n\ts_branch .Laf_re_%=\n" \
  ".Laf_r3_%=:\n\tv_accvgpr_read_b32 %0, a77\n\tv_accvgpr_read_b32 %1, a78\n\tv_accvgpr_read_b32 %2, a79\n\tv_accvgpr_read_b32 %3, a80\n\tv_accvgpr_read_b32 %4, a81\n\tv_accvgpr_read_b32 %5, a82\n\tv_accvgpr_read_b32 %6, a83\n\tv_accvgpr_read_b32 %7, a84\n\tv_accvgpr_read_b32 %8, a85\n\tv_accvgpr_read_b32 %9, a86\n\tv_accvgpr_read_b32 %10, a87\n\ts_branch .Laf_re_%=\n" \
  ".Laf_r2_%=:\n\tv_accvgpr_read_b32 %0, a55\n\tv_accvgpr_read_b32 %1, a56\n\tv_accvgpr_read_b32 %2, a57\n\tv_accvgpr_read_b32 %3, a58\n\tv_accvgpr_read_b32 %4, a59\n\tv_accvgpr_read_b32 %5, a60\n\tv_accvgpr_read_b32 %6, a61\n\tv_accvgpr_read_b32 %7, a62\n\tv_accvgpr_read_b32 %8, a63\n\tv_accvgpr_read_b32 %9, a64\n\tv_accvgpr_read_b32 %10, a65\n\ts_branch .Laf_re_%=\n" \
  ".Laf_r1_%=:\n\tv_accvgpr_read_b32 %0, a33\n\tv_accvgpr_read_b32 %1, a34\n\tv_accvgpr_read_b32 %2, a35\n\tv_accvgpr_read_b32 %3, a36\n\tv_accvgpr_read_b32 %4, a37\n\tv_accvgpr_read_b32 %5, a38\n\tv_accvgpr_read_b32 %6, a39\n\tv_accvgpr_read_b32 %7, a40\n\tv_accvgpr_read_b32 %8, a41\n\tv_accvgpr_read_b32 %9, a42\n\tv_accvgpr_read_b32 %10, a43\n\ts_branch .Laf_re_%=\n" \
  ".Laf_r0_%=:\n\tv_accvgpr_read_b32 %0, a11\n\tv_accvgpr_read_b32 %1, a12\n\tv_accvgpr_read_b32 %2, a13\n\tv_accvgpr_read_b32 %3, a14\n\tv_accvgpr_read_b32 %4, a15\n\tv_accvgpr_read_b32 %5, a16\n\tv_accvgpr_read_b32 %6, a17\n\tv_accvgpr_read_b32 %7, a18\n\tv_accvgpr_read_b32 %8, a19\n\tv_accvgpr_read_b32 %9, a20\n\tv_accvgpr_read_b32 %10, a21\n\t" \
  ".Laf_re_%=:"
#define AF_SLOT_WRITE_ASM_1 \
  "s_cmp_lg_u32 %11, 0\n\ts_cbranch_scc1 .Laf_w0_%=\n\tv_accvgpr_write_b32 a11, %0\n\tv_accvgpr_write_b32 a12, %1\n\tv_accvgpr_write_b32 a13, %2\n\tv_accvgpr_write_b32 a14, %3\n\tv_accvgpr_write_b32 a15, %4\n\tv_accvgpr_write_b32 a16, %5\n\tv_accvgpr_write_b32 a17, %6\n\tv_accvgpr_write_b32 a18, %7\n\tv_accvgpr_write_b32 a19, %8\n\tv_accvgpr_write_b32 a20, %9\n\tv_accvgpr_write_b32 a21, %10\n\t.Laf_w0_%=:\n\t" \
  "s_cmp_lg_u32 %11, 1\n\ts_cbranch_scc1 .Laf_w1_%=\n\tv_accvgpr_write_b32 a33, %0\n\tv_accvgpr_write_b32 a34, %1\n\tv_accvgpr_write_b32 a35, %2\n\tv_accvgpr_write_b32 a36, %3\n\tv_accvgpr_write_b32 a37, %4\n\tv_accvgpr_write_b32 a38, %5\n\tv_accvgpr_write_b32 a39, %6\n\tv_accvgpr_write_b32 a40, %7\n\tv_accvgpr_write_b32 a41, %8\n\tv_accvgpr_write_b32 a42, %9\n\tv_accvgpr_write_b32 a43, %10\n\t.Laf_w1_%=:\n\t" \
  "s_cmp_lg_u32 %11, 2\n\ts_cbranch_scc1 .Laf_w2_%=\n\tv_accvgpr_write_b32 a55, %0\n\tv_accvgpr_write_b32 a56, %1\n\tv_accvgpr_write_b32 a57, %2\n\tv_accvgpr_write_b32 a58, %3\n\tv_accvgpr_write_b32 a59, %4\n\tv_accvgpr_write_b32 a60, %5\n\tv_accvgpr_write_b32 a61, %6\n\tv_accvgpr_write_b32 a62, %7\n\tv_accvgpr_write_b32 a63, %8\n\tv_accvgpr_write_b32 a64, %9\n\tv_accvgpr_write_b32 a65, %10\n\t.Laf_w2_%=:\n\t" \
  "s_cmp_lg_u32 %11, 3\n\ts_cbranch_scc1 .Laf_w3_%=\n\tv_accvgpr_write_b32 a77, %0\n\tv_accvgpr_write_b32 a78, %1\n\tv_accvgpr_write_b32 a79, %2\n\tv_accvgpr_write_b32 a80, %3\n\tv_accvgpr_write_b32 a81, %4\n\tv_accvgpr_write_b32 a82, %5\n\tv_accvgpr_write_b32 a83, %6\n\tv_accvgpr_write_b32 a84, %7\n\tv_accvgpr_write_b32 a85, %8\n\tv_accvgpr_write_b32 a86, %9\n\tv_accvgpr_write_b32 a87, %10\n\t.Laf_w3_%=:\n\t" \
  "s_cmp_lg_u32 %11, 4\n\ts_cbranch_scc1 .Laf_w4_%=\n\tv_accvgpr_write_b32 a99, %0\n\tv_accvgpr_write_b32 a100, %1\n\tv_accvgpr_write_b32 a101, %2\n\tv_accvgpr_write_b32 a102, %3\n\tv_accvgpr_write_b32 a103, %4\n\tv_accvgpr_write_b32 a104, %5\n\tv_accvgpr_write_b32 a105, %6\n\tv_accvgpr_write_b32 a106, %7\n\tv_accvgpr_write_b32 a107, %8\n\tv_accvgpr_write_b32 a108, %9\n\tv_accvgpr_write_b32 a109, %10\n\t.Laf_w4_%=:\n\t"
#define AF_SLOT_CLOBBER_1 "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109"

struct AfWords { unsigned int w[11]; };
#define AF_SLOT_READ_J(J, idx_, L)                                                                                    \
  {                                                                                                                   \
    AfWords u_;                                                                                                       \
    asm volatile(AF_SLOT_READ_ASM_##J                                                                                 \
                 : "=&v"(u_.w[0]), "=&v"(u_.w[1]), "=&v"(u_.w[2]), "=&v"(u_.w[3]), "=&v"(u_.w[4]), "=&v"(u_.w[5]),    \
                   "=&v"(u_.w[6]), "=&v"(u_.w[7]), "=&v"(u_.w[8]), "=&v"(u_.w[9]), "=&v"(u_.w[10])                    \
                 : "s"(idx_)                                                                                          \
                 : "scc");                                                                                            \
    _Pragma("unroll") for (int t_ = 0; t_ < 5; ++t_) L.v[J][t_] = __hiloint2double((int)u_.w[2 * t_ + 1], (int)u_.w[2 * t_]); \
    L.c[J] = u_.w[10];                                                                                                \
  }
#define AF_SLOT_WRITE_J(J, idx_, L)                                                                                   \
  {                                                                                                                   \
    AfWords u_;                                                                                                       \
    _Pragma("unroll") for (int t_ = 0; t_ < 5; ++t_)                                                                  \
    {                                                                                                                 \
      u_.w[2 * t_] = (unsigned int)__double2loint(L.v[J][t_]);                                                        \
      u_.w[2 * t_ + 1] = (unsigned int)__double2hiint(L.v[J][t_]);                                                    \
    }                                                                                                                 \
    u_.w[10] = L.c[J];                                                                                                \
    asm volatile(AF_SLOT_WRITE_ASM_##J                                                                                \
                 :                                                                                                    \
                 : "v"(u_.w[0]), "v"(u_.w[1]), "v"(u_.w[2]), "v"(u_.w[3]), "v"(u_.w[4]), "v"(u_.w[5]), "v"(u_.w[6]),  \
                   "v"(u_.w[7]), "v"(u_.w[8]), "v"(u_.w[9]), "v"(u_.w[10]), "s"(idx_)                                 \
                 : "scc", AF_SLOT_CLOBBER_##J);                                                                       \
  }
// (slot numbers above 4 read slot 4 / write nothing)
#define AF_SLOT_READ(idx_expr, L)                                                                                     \
  {                                                                                                                   \
    const unsigned int idx_ = __builtin_amdgcn_readfirstlane(idx_expr);                                               \
    AF_SLOT_READ_J(0, idx_, L)                                                                                        \
    AF_SLOT_READ_J(1, idx_, L)                                                                                        \
  }
#define AF_SLOT_WRITE(idx_expr, L)                                                                                    \
  {                                                                                                                   \
    const unsigned int idx_ = __builtin_amdgcn_readfirstlane(idx_expr);                                               \
    AF_SLOT_WRITE_J(0, idx_, L)                                                                                       \
    AF_SLOT_WRITE_J(1, idx_, L)                                                                                       \
  }
static_assert(AF_J == 2 && AF_NSLOT == 5, "the slot macros spell out two sub-tiles and five slots");

// Round 5: SEGMENTS (partials_fused.hpp).  The list may come as up to eight independent sub-lists, each a plan of its
// own (header, one record per op, the cyclic last record); the work items are (tile, segment) pairs, every tile of
// segment 0 first.  A workgroup that moves on to another segment starts cold -- everybody done with the blocks in
// LDS, the new segment's first blocks requested -- which the segment-major order makes a once-per-launch event.
// segtab: [segment] = {first record, ops}.
// NT: 0 plain stores, 1 the tiles non-temporal, 2 non-temporal except the values a later op of the list copies back
// (AF_KEEP).  An instance of its own: the branch around every store cost a list WITHOUT such values 3.5 % when all
// non-temporal lists ran through it (config 3: 1,710 -> 1,767 us per call, profiles/r6_aa_keep_instance_ab.txt).
template <int MODE, int NT>
__global__ __launch_bounds__(256, 2) void k_aa_fused(const AaRec * __restrict__ plan0, const unsigned int * __restrict__ segtab,
                                                     unsigned int nsegs, unsigned int sites,
                                                     const char * aorder, unsigned int ms, double2 * sink,
                                                     unsigned int * next_tile, unsigned int static_rounds)
{
  // (Round 6, measured and taken out again -- profiles/r6_aa_phase_offset_ab.txt: the second workgroup of every CU
  // started 20 / 40 / 75 / 150 us late, so that the two workgroups of a CU would not be at the same op of the list at
  // the same time.  C3 1.764 -> 1.784 / 1.802 / 1.834 / 1.853 ms, the 200-taxon random tree 3.401 -> 3.411 / 3.422 /
  // 3.439 / 3.464: the delay is paid in full and buys nothing -- the workgroups of a CU are not in step to begin with.)
  extern __shared__ double2 lds_af[];
  const unsigned int lane = threadIdx.x & 63u;
  const unsigned int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned int q = lane >> 4, rate = (lane >> 2) & 3u, n = lane & 3u;
  // LDS: [left block 0][left block 1][right block][per wave: stage 0, stage 1][tile word]
  char * lds = reinterpret_cast<char *>(lds_af);
  const unsigned int lds_b = __builtin_amdgcn_readfirstlane((unsigned int)(uintptr_t)(PLL_LDS char *)lds_af);
  const unsigned int x0buf_b = lds_b, x1buf_b = lds_b + AF_MAT_B, ybuf_b = lds_b + 2 * AF_MAT_B;
  const unsigned int st0_b = lds_b + 3 * AF_MAT_B + wave * 2 * AF_TILE_B, st1_b = st0_b + AF_TILE_B;
  char * st0 = lds + 3 * AF_MAT_B + wave * 2 * AF_TILE_B;
  char * st1 = st0 + AF_TILE_B;
  unsigned int * tile_word = reinterpret_cast<unsigned int *>(lds + 3 * AF_MAT_B + 8 * AF_TILE_B);
  // what never changes for a lane
  const unsigned int lane16 = lane * 16u;
  const unsigned int boff = n * 640u + rate * 160u + q * 32u;        // its four states 4q.. of sub-tile 0 in a stage
  const unsigned int boff5 = n * 640u + rate * 160u + 128u + q * 8u; // state 16 + q
  const char * x0lane = lds + lane * 8u, * x1lane = lds + AF_MAT_B + lane * 8u, * ylane = lds + 2 * AF_MAT_B + lane * 8u;
  // (round 5: the offsets that only the mat-vecs use are formed where they are used, from a copy of the lane number
  // the optimiser cannot see through -- kept in registers across the op loop they were what the allocator, at its
  // limit of 128, parked in a0: a slot; tools/check_agprs.py)
  auto lane_again = [&]() __attribute__((always_inline)) {
    unsigned int l = lane;
    asm volatile("" : "+v"(l));
    return l;
  };
  // the lane's (q, rate) class within a fifth-step block, x 32 bytes
  auto cls32_of = [](unsigned int l) __attribute__((always_inline)) { return ((l >> 4) * 4u + ((l >> 2) & 3u)) * 32u; };
  // the lane's column in a stage
  auto coloff_of = [](unsigned int l) __attribute__((always_inline)) { return (l & 3u) * 640u + ((l >> 2) & 3u) * 160u; };
  const unsigned long long sink_a = (unsigned long long)(uintptr_t)(sink + ((size_t)blockIdx.x * 4u + wave) * 8u); // (128 bytes per wave)
  const unsigned long long aorder_a = (unsigned long long)(uintptr_t)aorder;


#ifdef PLLHIP_AF_TIMING
  const unsigned int exp_mask = __builtin_amdgcn_readfirstlane(af_exp_mask);
#endif
  // a matrix block into LDS: 13 pieces of 1 KB over the four waves
  auto stage_matrix = [&](unsigned int buf_b, unsigned int off) __attribute__((always_inline)) {
    if (AF_EXP(8u)) return;
    // (waves 0..2 four consecutive pieces each, wave 3 the thirteenth)
    const unsigned int w4 = wave * 4096u;
    if (wave < 3u) af_dma_run<4>(buf_b + w4, aorder_a + off + w4, lane16);
    else af_dma_run<1>(buf_b + 12288u, aorder_a + off + 12288u, lane16);
  };

  const size_t tiles = ((size_t)sites + AF_WGS - 1) / AF_WGS;
  const size_t items = tiles * nsegs; // (tile, segment) pairs, segment-major
  unsigned int xpar = 0u; // which left buffer the current op multiplies by
  unsigned int cur_rec0 = ~0u; // the segment the workgroup is in: its header's record number
  const AaRec * const plan = plan0;
  unsigned int ch_p, ch_q; // tip characters of the next op / the op after next, in turn: lane l holds row (l >> 3) & 3, site l & 7
  for (unsigned int round = 0;; ++round) // (32 bits: as a size_t its bound lived in a register pair of every lane)
  {
    // a workgroup's first items are its own by a fixed stride, the last rounds' worth come from a
    // counter (the XCDs do not write at the same rate, partials_fused.hip)
    size_t tile;
    if (round < static_rounds || !next_tile)
      tile = (size_t)blockIdx.x + (size_t)round * gridDim.x;
    else
    {
      if (threadIdx.x == 0) *tile_word = atomicAdd(next_tile, 1u);
      __syncthreads();
      tile = (size_t)static_rounds * gridDim.x + (unsigned int)__builtin_amdgcn_readfirstlane((int)*tile_word);
    }
    if (tile >= items) break;
    unsigned int seg = 0u;
    while (tile >= tiles)
    {
      tile -= tiles;
      ++seg;
    }
    const unsigned int rec0 = ((af_words)(unsigned long long)(segtab + 2u * __builtin_amdgcn_readfirstlane(seg)))[0];
    if (rec0 != cur_rec0)
    {
      // a cold start: everybody is done with the blocks in LDS (and this wave's own requests have landed), then
      // the segment's first two ops' blocks (from then on every op requests those of the ops ahead, cyclically)
      if (cur_rec0 != ~0u)
      {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      cur_rec0 = rec0;
      xpar = 0u;
      const AfW<16, 3> h = af_load<16, 3>(plan, rec0), r1 = af_load<16, 3>(plan, rec0 + 1u);
      if (((h[18] >> 8) & AF_KIND_MASK) == 0u) stage_matrix(x0buf_b, h[16]);
      if ((r1[18] & AF_KIND_MASK) <= 1u) stage_matrix(ybuf_b, r1[17]);
      if (((r1[18] >> 8) & AF_KIND_MASK) == 0u) stage_matrix(x1buf_b, r1[16]);
    }
    const size_t site0 = tile * AF_WGS + (size_t)wave * AF_WS;
    const unsigned long long clv_off = (unsigned long long)site0 * 640u;
    const unsigned long long cnt_off = (unsigned long long)site0 * (MODE == SCALE_RATE ? 16u : 4u);

    // an operand without a slot: from HBM through stage 1 into a slot (AF_RELOAD_TAKE)
    auto reload_issue = [&](unsigned long long src) __attribute__((always_inline)) {
      if (AF_EXP(64u)) return; // (tool build, bit 64: nothing is copied back at all)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (the stage's last readers are done)
      af_dma_run<4>(st1_b, src + clv_off, lane16);
      af_dma_run<1>(st1_b + 4096u, src + clv_off + 4096u, lane16);
    };
    auto reload_counts = [&](unsigned long long cnt, unsigned int (&cj)[AF_J]) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < AF_J; ++j)
      {
        cj[j] = 0u;
        unsigned int o = MODE == SCALE_RATE ? ((4u * j + n) * 4u + rate) * 4u : (4u * j + n) * 4u;
        asm volatile("" : "+v"(o));
        if (MODE != SCALE_NONE && cnt && !AF_EXP(64u)) cj[j] = *(const unsigned int PLL_GLOBAL *)(af_base(cnt + cnt_off) + o);
      }
    };
    // (a macro, not a lambda: a lambda would capture the slot variables by reference, and
    // variables whose address is taken anywhere stay in memory)
#define AF_RELOAD_TAKE(slot, cj)                                   \
  {                                                                \
    if (!AF_EXP(32u | 64u)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* (tool build, bit 32: no wait for a reload) */ \
    AfSlot tmp;                                                    \
    af_read_tile(st1 + boff, st1 + boff5, tmp.v);                  \
    tmp.c[0] = cj[0];                                              \
    tmp.c[1] = cj[1];                                              \
    AF_SLOT_WRITE(slot, tmp)                                       \
  }
    // the tip characters of the op BEHIND record r (words 8..15: its four rows), ONE load (round 4: they were four,
    // and every vector-memory instruction of this kernel queues behind the other waves' on the texture path): lane l
    // takes the character of row (l >> 3) & 3 at site l & 7 of the tile; af_chars deals them out
    auto request_chars = [&](const AfW<0, 28> & r, unsigned int & c) __attribute__((always_inline)) {
      const unsigned int sel = (lane >> 3) & 3u;
      const unsigned long long base = sel == 0u ? r.quad(8) : sel == 1u ? r.quad(10) : sel == 2u ? r.quad(12) : r.quad(14);
      c = *(const unsigned char PLL_GLOBAL *)(base + site0 + (lane & 7u));
    };
    // The table row each of the tile's eight sites reads, as WAVE-UNIFORM values: the characters sit packed in one
    // register (lane 8 r + s: row r, site s), so v_readlane hands them to the scalar unit, which forms the pair
    // indices.  (Round 3 dealt the characters out with ds_bpermute and fetched every granule's row index with another:
    // fourteen trips through the LDS crossbar per gathered op, ten of them for a second table a tip-tip op does not have.)
    // byte offsets of the lane's five granules: granule it * 64 + lane of the tile is granule rr of site sl's row; the
    // 64 granules of a step belong to at most three consecutive sites, whose bounds are compile-time constants.  (The
    // rows of a step's sites are read when the step needs them: eight of them held at once cost scalar registers the
    // kernel does not have.)
    auto gather_offsets = [&](unsigned int chars, bool pairs, unsigned int first_row, unsigned int (&o)[5]) __attribute__((always_inline)) {
      const unsigned int cc = chars >= ms ? 0u : chars;
      auto row_of = [&](unsigned int sidx) __attribute__((always_inline)) {
        const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)cc, (int)(8u * first_row + sidx));
        if (!pairs) return hi;
        return hi * ms + (unsigned int)__builtin_amdgcn_readlane((int)cc, (int)(8u * first_row + 8u + sidx));
      };
      unsigned int lane_l = lane;
      asm volatile("" : "+v"(lane_l)); // (recomputed, not kept)
#pragma unroll
      for (unsigned int it = 0; it < 5; ++it)
      {
        const unsigned int a = (it * 64u) / 40u, b1 = 40u * (a + 1u) - it * 64u, b2 = 40u * (a + 2u) - it * 64u;
        unsigned int r = row_of(a + 1u), sl = a + 1u;
        if (b2 < 64u)
        {
          const unsigned int r2 = row_of(a + 2u);
          r = lane_l < b2 ? r : r2;
          sl = lane_l < b2 ? sl : a + 2u;
        }
        {
          const unsigned int r0 = row_of(a);
          r = lane_l < b1 ? r0 : r;
          sl = lane_l < b1 ? a : sl;
        }
        o[it] = r * 640u + (lane_l + it * 64u - 40u * sl) * 16u;
        asm volatile("" : "+v"(o[it])); // (one step's scalars at a time)
      }
    };
    // what an op of kind `nkind` gathers an op ahead, with its characters (in chars): a lookup its
    // table entries -- LDS-DMA with one address per lane, straight into the two stages in the
    // layout of the stores --, a tip-inner op the tip's factor (a row of its table per site, into stage 1)
    auto next_gathers = [&](unsigned int nkind, bool one_table, unsigned long long tab_l, unsigned long long tab_r, unsigned int chars) __attribute__((always_inline)) {
      if (AF_EXP(4u)) return;
      unsigned int o[5];
      if (nkind == 2u)
      {
        gather_offsets(chars, true, 0u, o);
        // (a single table goes to stage 1: stage 0 is where an op transposes its result, so an inner-inner op can
        // issue the NEXT op's one gather while it still multiplies -- partials_aa_fused_op.inc, "early")
        if (one_table) af_dma_gather5(st1_b, tab_l, o);
        else
        {
          af_dma_gather5(st0_b, tab_l, o);
          gather_offsets(chars, true, 2u, o);
          af_dma_gather5(st1_b, tab_r, o);
        }
      }
      else if (nkind == 1u)
      {
        // the tip's factor: row `code` of its table is 640 bytes laid out like a site of a CLV
        gather_offsets(chars, false, 0u, o);
        af_dma_gather5(st1_b, tab_l, o); // (stage 1: stage 0 takes the inner child's columns, af_matvec_plain)
      }
    };

    // ---- prologue: what the ops before op 0 would have done for it (reloads, characters, gathers)
    AfW<0, 28> rc_a = af_load<0, 28>(plan, rec0 + 1u), rc_b;
    {
      const AfW<0, 28> h = af_load<0, 28>(plan, rec0);
      const unsigned int hf = __builtin_amdgcn_readfirstlane(h[18]);
      // (the stages' last readers -- the tile before -- are done)
      if (hf & AF_RELOAD_A)
      {
        unsigned int cj[AF_J];
        reload_issue(h.quad(20));
        reload_counts(h.quad(22), cj);
        AF_RELOAD_TAKE((h[19] >> 12) & 15u, cj)
      }
      if (hf & AF_RELOAD_B)
      {
        unsigned int cj[AF_J];
        reload_issue(h.quad(24));
        reload_counts(h.quad(26), cj);
        AF_RELOAD_TAKE((h[19] >> 16) & 15u, cj)
      }
      request_chars(h, ch_p); // op 0's
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      {
        const unsigned int f0 = __builtin_amdgcn_readfirstlane(rc_a[18]);
        next_gathers(f0 & AF_KIND_MASK, (f0 & AF_ONE_TABLE) != 0u, rc_a.quad(4), rc_a.quad(6), ch_p);
      }
      // op 1's (used at the end of op 0): a record names the rows of the op after next, the last op's those of op 1
      const AfW<8, 8> r1 = af_load<8, 8>(plan, h[17]); // (the header's yoff: the segment's last record)
      AfW<0, 28> rows1 = rc_a;
#pragma unroll
      for (int t = 0; t < 8; ++t) rows1.w[8 + t] = r1[8 + t];
      request_chars(rows1, ch_q);
    }
    // (the first tile's blocks were requested without a barrier behind them; a tile's last stores
    // are waited for here too: once per tile)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

#ifdef PLLHIP_AF_TIMING
    // (tool build, tools/aa_fused_timing.sh: where a wave's cycles go, by op kind and phase)
    unsigned long long phase_cycles[3][11] = {};
    unsigned int nkind_ops[3] = {};
    unsigned long long t_last = __builtin_readcyclecounter();
#ifdef PLLHIP_AF_NOTICKS
#define AF_TICK(ph)
#else
#define AF_TICK(ph) { const unsigned long long t_now = __builtin_readcyclecounter(); phase_cycles[kind][ph] += t_now - t_last; t_last = t_now; }
#endif
#else
#define AF_TICK(ph)
#endif
    bool af_last;
    for (;;)
    {
#define AF_RC rc_a
#define AF_RN rc_b
#define AF_CH_REQ ch_p
#define AF_CH_USE ch_q
#include "partials_aa_fused_op.inc"
#undef AF_RC
#undef AF_RN
#undef AF_CH_REQ
#undef AF_CH_USE
      if (af_last) break;
#define AF_RC rc_b
#define AF_RN rc_a
#define AF_CH_REQ ch_q
#define AF_CH_USE ch_p
#include "partials_aa_fused_op.inc"
#undef AF_RC
#undef AF_RN
#undef AF_CH_REQ
#undef AF_CH_USE
      if (af_last) break;
    }
#if defined(PLLHIP_AF_TIMING) && !defined(PLLHIP_AF_NOTICKS)
    if (lane == 0 && round == 2 && wave == 1 && (blockIdx.x == 0 || blockIdx.x == 101 || blockIdx.x == 202 || blockIdx.x == 303))
      for (int kd = 0; kd < 3; ++kd)
        if (nkind_ops[kd])
          printf("kind %d: %u ops: first half %llu, wait A %llu, barrier A %llu, [slot read %llu, products %llu] rest of second half %llu, barrier B %llu, "
                 "[lds wait %llu, blocks %llu, gathers %llu] characters + stores %llu cycles per op\n",
                 kd, nkind_ops[kd], phase_cycles[kd][0] / nkind_ops[kd], phase_cycles[kd][1] / nkind_ops[kd], phase_cycles[kd][2] / nkind_ops[kd],
                 phase_cycles[kd][6] / nkind_ops[kd], phase_cycles[kd][7] / nkind_ops[kd], phase_cycles[kd][3] / nkind_ops[kd], phase_cycles[kd][4] / nkind_ops[kd],
                 phase_cycles[kd][8] / nkind_ops[kd], phase_cycles[kd][9] / nkind_ops[kd], phase_cycles[kd][10] / nkind_ops[kd], phase_cycles[kd][5] / nkind_ops[kd]);
#endif
  }
}
} // namespace

// ---------------------------------------------------------------- host

struct pllhip_aa_fused_cache
{
  std::vector<pllhip_op_t> last_ops;
  unsigned int epoch = 0, maxstates = 0;
  // what runs ahead of the list kernel
  std::vector<PartialsArgs> tt_ops;   // tip-tip ops, grouped by scaling mode
  std::vector<int> tt_modes;
  std::vector<PartialsArgs> lk_ops, lk_k1, lk_k2; // lookup ops and the tip-tip ops that made their children
  unsigned int nops = 0, nmat = 0, ntip = 0;
  int mode = SCALE_NONE;
  size_t off_mat = 0, off_tip = 0;    // job arrays within d_plan
  void * d_plan = nullptr;
  void * h_plan = nullptr;
  size_t plan_cap = 0;
  hipEvent_t done = nullptr;
  bool pending = false;
  char * d_aorder = nullptr;
  size_t aorder_cap = 0;
  char * d_titab = nullptr;
  size_t titab_cap = 0;
  char * d_pairtab = nullptr;          // pair tables of the list's tip-tip ops (AfPairJob)
  size_t pairtab_cap = 0, off_pair = 0;
  unsigned int npair = 0;
  size_t off_lk = 0;                   // lookup-table jobs (AaLookupJob; nlk == 0: the tables come from launches of their own)
  unsigned int nlk = 0;
  size_t off_seg = 0;                  // (round 5) the segment table: {first record, ops} per segment
  unsigned int nsegs = 1;
  // (round 6) the scaling certificate: the plan was made for these marks of the operands it reads from earlier calls
  // (index, mark) and in this mode; what it leaves marked; whether any op tests, and what a trip means (ctx.hpp)
  bool ti_mfma = false;
  std::vector<std::pair<unsigned int, double>> ext_marks, out_marks;
  int cert_kind = 0;
  bool cert_too_wide = false; // the bounds outgrew the widest window: every launch counts as uncertified
  // what the kept plan is made of (pllhip_aa_list_kinds: bench.py's flop count): ops, tip-tip ahead of the list,
  // tip-tip in the list, lookups, inner-inner on the matrix cores, tip-inner on the matrix cores, tip-inner on the
  // vector unit, operands reloaded
  unsigned int kinds_of_plan[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool any_keep = false; // some op's value is copied back by a later op of the list (AF_KEEP): the kernel instance that looks
};

void pllhip_aa_fused_free(pllhip_ctx * c)
{
  pllhip_aa_fused_cache * k = c->aa_fused;
  if (!k) return;
  if (k->d_plan) (void)hipFree(k->d_plan);
  if (k->h_plan) (void)hipHostFree(k->h_plan);
  if (k->done) (void)hipEventDestroy(k->done);
  if (k->d_aorder) (void)hipFree(k->d_aorder);
  if (k->d_titab) (void)hipFree(k->d_titab);
  if (k->d_pairtab) (void)hipFree(k->d_pairtab);
  delete k;
  c->aa_fused = nullptr;
}

// everything the kept plan describes, again: tip-tip ops, tables, matrices in operand order, the list
static int aa_fused_launch(pllhip_ctx * c, bool tables_built)
{
  pllhip_aa_fused_cache & k = *c->aa_fused;
  // tip-tip ops of one scaling mode per launch, PLLHIP_BATCH_MAX at a time
  for (size_t first = 0; first < k.tt_ops.size();)
  {
    PartialsBatch b;
    unsigned int nb = 0;
    const int mode = k.tt_modes[first];
    while (first < k.tt_ops.size() && k.tt_modes[first] == mode && nb < PLLHIP_BATCH_MAX) b.op[nb++] = k.tt_ops[first++];
    const int rc = pllhip_launch_aa_batch(c, b, nb, 2, mode);
    if (rc) return rc;
  }
  if (!k.lk_ops.empty() && !tables_built && !k.nlk)
  {
    // (same pool, same places: the records' addresses hold while the epoch does)
    std::vector<AaLookupTables> tabs(k.lk_ops.size());
    const int rc = pllhip_aa_lookup_tables(c, k.lk_ops.data(), k.lk_k1.data(), k.lk_k2.data(),
                                           (unsigned int)k.lk_ops.size(), tabs.data());
    if (rc) return rc;
  }
  const char * plan = static_cast<const char *>(k.d_plan);
  if (k.nmat + k.ntip + k.npair + k.nlk)
  {
    // (tool switch, wrong results: PLLHIP_AF_PREP_SKIP = 1 no matrix jobs | 2 no tip tables | 4 no pair tables |
    // 8 no lookup tables -- what each kind of job costs the launch, tools/aa_prepare_ab.sh)
    const unsigned int skip = pllhip_env("PLLHIP_AF_PREP_SKIP") ? (unsigned int)atoi(pllhip_env("PLLHIP_AF_PREP_SKIP")) : 0u;
    const unsigned int nmat = (skip & 1u) ? 0u : k.nmat, ntip = (skip & 2u) ? 0u : k.ntip;
    const unsigned int npair = (skip & 4u) ? 0u : k.npair, nlk = (skip & 8u) ? 0u : k.nlk;
    k_af_prepare<<<std::max(1u, nmat + ntip + npair * AF_PAIR_WGS + nlk * c->maxstates), 256, 0, c->stream>>>(
        (const AfMatJob *)(plan + k.off_mat), nmat, (const AfTipJob *)(plan + k.off_tip), ntip, k.d_aorder,
        k.d_titab, c->tipmap, c->maxstates, c->d_tile_counter, (const AfPairJob *)(plan + k.off_pair), npair,
        k.d_pairtab, (const AaLookupJob *)(plan + k.off_lk), nlk);
    HIP_TRY(hipGetLastError());
  }
  else HIP_TRY(hipMemsetAsync(c->d_tile_counter, 0, sizeof(unsigned int), c->stream));
  const size_t tiles = (((size_t)c->sh.sites + AF_WGS - 1) / AF_WGS) * k.nsegs; // (work items: (tile, segment) pairs)
  size_t grid = tiles;
  const size_t cap = pllhip_env("PLLHIP_AA_GRID_CAP") ? (size_t)atoi(pllhip_env("PLLHIP_AA_GRID_CAP")) : (size_t)c->num_cus * 2; // (tests: many tiles per workgroup)
  if (grid > cap) grid = cap;
  const size_t rounds = tiles / grid;
  const unsigned int dynamic_rounds = pllhip_env("PLLHIP_FUSED_DYNAMIC_ROUNDS") ? (unsigned int)atoi(pllhip_env("PLLHIP_FUSED_DYNAMIC_ROUNDS"))
                                                                            : (unsigned int)std::max<size_t>(2, rounds / 3);
  const unsigned int static_rounds = rounds > dynamic_rounds ? (unsigned int)(rounds - dynamic_rounds) : 1u;
  unsigned int * counter = pllhip_env("PLLHIP_FUSED_STATIC_TILES") ? nullptr : c->d_tile_counter;
#ifdef PLLHIP_AF_TIMING
  {
    const unsigned int m = pllhip_env("PLLHIP_AF_EXP") ? (unsigned int)atoi(pllhip_env("PLLHIP_AF_EXP")) : 0u;
    HIP_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(af_exp_mask), &m, sizeof(m), 0, hipMemcpyHostToDevice, c->stream));
  }
#endif
  const bool nt = pllhip_use_nt(c);
#define AF_LAUNCH(MODEV, NTV)                                                                                          \
  do {                                                                                                                 \
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_aa_fused<MODEV, NTV>),                               \
                                hipFuncAttributeMaxDynamicSharedMemorySize, AF_LDS_B));                                \
    k_aa_fused<MODEV, NTV><<<(unsigned int)grid, 256, AF_LDS_B, c->stream>>>(                                          \
        (const AaRec *)k.d_plan, (const unsigned int *)(plan + k.off_seg), k.nsegs, c->sh.sites, k.d_aorder,           \
        c->maxstates, (double2 *)c->d_sink, counter, static_rounds);                                                   \
  } while (0)
  // (tool switch: every non-temporal list through the instance that looks for AF_KEEP, profiles/r6_aa_keep_instance_ab.txt)
  const bool force_keep_instance = pllhip_env("PLLHIP_AA_KEEP_INSTANCE") && atoi(pllhip_env("PLLHIP_AA_KEEP_INSTANCE")) != 0;
#define AF_LAUNCH_MODE(MODEV)                                                                                          \
  do {                                                                                                                 \
    if (!nt) AF_LAUNCH(MODEV, 0);                                                                                      \
    else if (!k.any_keep && !force_keep_instance) AF_LAUNCH(MODEV, 1);                                                 \
    else AF_LAUNCH(MODEV, 2);                                                                                          \
  } while (0)
  if (k.mode == SCALE_NONE) AF_LAUNCH_MODE(SCALE_NONE);
  else if (k.mode == SCALE_RATE) AF_LAUNCH_MODE(SCALE_RATE);
  else AF_LAUNCH_MODE(SCALE_SITE);
#undef AF_LAUNCH_MODE
#undef AF_LAUNCH
  HIP_TRY(hipGetLastError());
  // the scaling certificate: what the list leaves marked, and whether its flag has to be looked at
  for (const auto & m : k.out_marks) pllhip_cert_mark_clv(c, m.first, m.second);
  if (k.cert_kind)
  {
    c->cert_pending = true;
    c->cert_kind = k.cert_kind;
    ++c->cert_stats[0];
    if (k.cert_too_wide) ++c->cert_stats[3];
  }
  return 0;
}

// what the last op list the 20-state whole-list kernel planned is made of (0: nothing planned on this context)
extern "C" int pllhip_aa_list_kinds(pllhip_ctx_t * c, unsigned int * out8)
{
  for (int t = 0; t < 8; ++t) out8[t] = 0;
  pllhip_ctx * s = c->shards.empty() ? c : c->shards[0];
  if (!s->aa_fused || s->aa_fused->last_ops.empty()) return 0;
  memcpy(out8, s->aa_fused->kinds_of_plan, 8 * sizeof(unsigned int));
  return 0;
}

// Returns 0 when the list has been enqueued, 1 when it is not one this path takes (the caller
// launches per level), < 0 on error.
static int aa_fused_update(pllhip_ctx * c, const pllhip_op_t * ops, unsigned int count, bool tt_wanted);

// Where a list's tip-tip ops run.  Round 3: AHEAD of the list kernel (a launch of k_aa_tt_rounds) for a list seen
// for the first time -- the cheaper plan then, 30 ops instead of 62 for BASELINE config 3 -- and INSIDE the list, as
// lookups over the two tip tables, once the very same list came again, unless that plan reloaded more than one
// operand.  Round 4: inside, always.  A tip-tip op of the list is ONE gather from its pair table now (AfPairJob) and
// the planner takes a third of the time (bump-allocated lists): measured with the tip-tip ops ahead / inside
// (profiles/r4_aa_tt_inside_ab.txt, one box): C3 2.10 / 1.80 ms, 64-taxon random tree 2.68 / 2.47, 200-taxon random
// tree (26 operands reloaded with the tip-tip ops ahead: every tip-tip result a matrix op consumes) 4.49 / 4.35.
// PLLHIP_AA_TT_INSIDE=0 puts them ahead again.
int pllhip_aa_fused_update(pllhip_ctx * c, const pllhip_op_t * ops, unsigned int count)
{
  return aa_fused_update(c, ops, count, true);
}

static int aa_fused_update(pllhip_ctx * c, const pllhip_op_t * ops, unsigned int count, bool tt_wanted)
{
  if (c->sh.states != 20 || c->sh.rate_cats != 4 || c->sh.asc_states || !c->rows.empty() ||
      c->aa_exact || count > PLLHIP_FUSED_MAX_OPS)
    return 1;
  if (c->sh.pattern_tip && (c->maxstates < 1 || c->maxstates > 32)) return 1;
  if (!c->aa_fused) c->aa_fused = new pllhip_aa_fused_cache();
  pllhip_aa_fused_cache & k = *c->aa_fused;
  // (PLLHIP_FUSED_DEBUG=3: where the host's time goes when a list is new)
  const bool host_times = c->fused_debug == 3;
  auto t_host = std::chrono::steady_clock::now();
  auto lap = [&](const char * what) {
    if (!host_times) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "pllhip 20-state list, host: %-28s %7.1f us\n", what, std::chrono::duration<double, std::micro>(now - t_host).count());
    t_host = now;
  };
  // Tip-inner mat-vecs on the matrix cores (the default, round 6) -- unless the list is being run again after its
  // certificate tripped, or could not be run again: a list that overwrites an operand it has read from an earlier
  // call (slot reuse across calls) is not idempotent, and keeps to the reference's order.
  const bool ti_mfma = c->aa_ti_mfma && !c->cert_force_exact && !c->no_batch;
  if (k.last_ops.size() == count && k.epoch == c->layout_epoch && k.maxstates == c->maxstates && k.ti_mfma == ti_mfma &&
      !c->fused_debug && memcmp(k.last_ops.data(), ops, (size_t)count * sizeof(pllhip_op_t)) == 0)
  {
    bool same_marks = true;
    for (const auto & m : k.ext_marks) same_marks = same_marks && pllhip_cert_err(c, m.first) == m.second;
    if (same_marks)
    {
      if (k.cert_kind == 1) c->cert_ops.assign(ops, ops + count);
      return aa_fused_launch(c, false);
    }
  }
  k.last_ops.clear();

  // ---- classify.  Tip-tip ops run ahead of the list: allowed only if nothing earlier in the list
  // wrote or read what they write (they read tips only).  An inner-inner op over two tip-tip
  // results of this list, or a tip-inner op over one, is a lookup.
  std::vector<PartialsArgs> args(count);
  std::vector<int> kinds(count), modes(count);
  const size_t nclv = c->clv.size(), nsc = c->sh.scale_buffers;
  std::vector<int> clv_touched(nclv, 0), sc_touched(nsc, 0), tt_writer(nclv, -1);
  std::vector<std::pair<int, int>> lk_kids(count, {-1, -1});
  bool any_scaler = false;
  const bool lookups_ok = pllhip_aa_cherry_covers(c, SCALE_SITE);
  const unsigned int lookups_max = lookups_ok ? pllhip_aa_lookup_budget(c) : 0u; // (their tables' pool is bounded)
  unsigned int lookups = 0;
  // Tip-tip ops INSIDE the list (kind 4 below): parent = tip table of the left matrix [character 1] (.) tip table of
  // the right matrix [character 2] is what a lookup op does with two other tables, so a tip-tip op is a lookup op
  // whose tables are the two tip tables k_af_prepare builds anyway and whose "pairs" are (0, character): no code of
  // its own in the kernel.  Its stores then interleave with the matrix ops of the other workgroup on the CU
  // instead of preceding the list as a launch of their own.  Measured (C3's partition, lists directed at five
  // edges): a list with no or one operand reloaded from HBM gains 4-5 % (2.09-2.11 against 2.19 ms), the two
  // lists with three reloads lose 10 % (2.47 against 2.25) -- so a list with more than one reload is planned again
  // with the tip-tip ops ahead of it (a rule from five lists, not a law).  PLLHIP_AA_TT_INSIDE=0 / 1: never / always.
  const char * tt_env = pllhip_env("PLLHIP_AA_TT_INSIDE");
  const bool tt_inside = c->sh.pattern_tip && (tt_env ? atoi(tt_env) != 0 : tt_wanted);
  for (unsigned int i = 0; i < count; ++i)
  {
    const int rc = pllhip_resolve_op(c, ops[i], args[i], kinds[i], modes[i]);
    if (rc) return rc;
    const pllhip_op_t & op = ops[i];
    if (kinds[i] == 2 && (clv_touched[op.parent_clv] || (op.parent_scaler >= 0 && sc_touched[op.parent_scaler]))) return 1;
    any_scaler = any_scaler || modes[i] != SCALE_NONE;
    const int plain = kinds[i];
    if (lookups < lookups_max && plain == 0 && tt_writer[op.child1_clv] >= 0 && tt_writer[op.child2_clv] >= 0)
    {
      lk_kids[i] = {tt_writer[op.child1_clv], tt_writer[op.child2_clv]};
      kinds[i] = 3;
      ++lookups;
    }
    else if (lookups < lookups_max && plain == 1)
    {
      const unsigned int inner = pllhip_is_tip(c, op.child1_clv) ? op.child2_clv : op.child1_clv;
      if (tt_writer[inner] >= 0)
      {
        lk_kids[i] = {-2, tt_writer[inner]};
        kinds[i] = 3;
        ++lookups;
      }
    }
    tt_writer[op.parent_clv] = plain == 2 ? (int)i : -1;
    clv_touched[op.parent_clv] = clv_touched[op.child1_clv] = clv_touched[op.child2_clv] = 1;
    if (op.parent_scaler >= 0) sc_touched[op.parent_scaler] = 1;
    if (op.child1_scaler >= 0) sc_touched[op.child1_scaler] = 1;
    if (op.child2_scaler >= 0) sc_touched[op.child2_scaler] = 1;
  }

  // ---- the scaling certificate (ctx.hpp): a bound on every op's relative difference from the reference's value --
  // its operands' bounds plus PLLHIP_CERT_OP_ERR when anything below it ran on the matrix cores' tip-inner path --, which
  // ops therefore test, and the window they test with
  std::vector<unsigned char> op_inexact(count, 0);
  bool list_ti_mfma = ti_mfma;
  double cert_window = PLLHIP_CERT_WINDOW_MIN;
  bool cert_too_wide = false;
  k.ext_marks.clear();
  k.out_marks.clear();
  {
    bool rerunnable = true;
    std::vector<unsigned char> written(nclv, 0), sc_written(nsc, 0), ext_read(nclv, 0), sc_ext_read(nsc, 0);
    for (unsigned int i = 0; i < count && ti_mfma; ++i)
    {
      const pllhip_op_t & op = ops[i];
      for (unsigned int ch : {op.child1_clv, op.child2_clv})
        if (!written[ch]) ext_read[ch] = 1;
      for (int sc : {op.child1_scaler, op.child2_scaler})
        if (sc >= 0 && !sc_written[sc]) sc_ext_read[sc] = 1;
      if (ext_read[op.parent_clv] || (op.parent_scaler >= 0 && sc_ext_read[op.parent_scaler])) rerunnable = false;
      written[op.parent_clv] = 1;
      if (op.parent_scaler >= 0) sc_written[op.parent_scaler] = 1;
    }
    list_ti_mfma = ti_mfma && rerunnable;
    std::vector<double> err(nclv, 0.0);
    std::vector<unsigned char> local(nclv, 0);
    bool any_ti = false;
    double worst = 0.0;
    for (int attempt = 0; attempt < 2; ++attempt)
    {
      // (first with the tip-inner mat-vecs on the matrix cores; if the bounds then outgrow every window -- operands
      // that earlier calls left with large bounds -- once more with every op in the reference's order)
      std::fill(local.begin(), local.end(), 0);
      k.ext_marks.clear();
      any_ti = false;
      worst = 0.0;
      for (unsigned int i = 0; i < count; ++i)
      {
        const pllhip_op_t & op = ops[i];
        const bool source = kinds[i] == 1 && list_ti_mfma;
        any_ti = any_ti || source;
        double in = 0.0;
        for (unsigned int ch : {op.child1_clv, op.child2_clv})
        {
          if (pllhip_is_tip(c, ch)) continue;
          if (!local[ch])
          {
            // an operand from an earlier call: the plan holds for THIS bound of it
            const double m = pllhip_cert_err(c, ch);
            bool seen = false;
            for (const auto & e : k.ext_marks) seen = seen || e.first == ch;
            if (!seen) k.ext_marks.push_back({ch, m});
            in += m;
          }
          else in += err[ch];
        }
        const double out = (source || in > 0.0) ? in + PLLHIP_CERT_OP_ERR : 0.0;
        err[op.parent_clv] = out;
        local[op.parent_clv] = 1;
        op_inexact[i] = out > 0.0 && op.parent_scaler >= 0;
        if (op_inexact[i] && out > worst) worst = out;
      }
      if (!(list_ti_mfma && 8.0 * worst > PLLHIP_CERT_WINDOW_MAX)) break;
      list_ti_mfma = false;
    }
    for (unsigned int i = 0; i < nclv; ++i)
      if (local[i]) k.out_marks.push_back({i, err[i]});
    k.cert_kind = !(worst > 0.0) ? 0 : (any_ti ? 1 : 2);
    cert_too_wide = 8.0 * worst > PLLHIP_CERT_WINDOW_MAX;
    cert_window = std::min(std::max(8.0 * worst, PLLHIP_CERT_WINDOW_MIN), PLLHIP_CERT_WINDOW_MAX);
    k.cert_too_wide = cert_too_wide;
    if (k.cert_kind == 1) c->cert_ops.assign(ops, ops + count);
  }
  k.ti_mfma = ti_mfma;
  lap("resolve + classify");
  // ---- the list the kernel walks: everything but the tip-tip ops, ordered and given slots by the
  // planner of the 4-state kernel (a lookup has no inner operands: a "tip-tip" op to the planner)
  std::vector<pllhip_op_t> rops;
  std::vector<PartialsArgs> rargs;
  std::vector<int> rkinds, orig;
  k.tt_ops.clear();
  k.any_keep = false;
  k.tt_modes.clear();
  for (int pass = 0; pass < 2; ++pass) // (tip-tip ops grouped by mode: without a scale buffer first)
    for (unsigned int i = 0; i < count; ++i)
      if (kinds[i] == 2 && !tt_inside && (modes[i] != SCALE_NONE) == (pass == 1))
      {
        k.tt_ops.push_back(args[i]);
        k.tt_modes.push_back(modes[i]);
      }
  for (unsigned int i = 0; i < count; ++i)
    if (kinds[i] != 2 || tt_inside)
    {
      if (kinds[i] == 2) kinds[i] = 4; // (a tip-tip op of the list)
      rops.push_back(ops[i]);
      rargs.push_back(args[i]);
      rkinds.push_back(kinds[i] >= 3 ? 2 : kinds[i]);
      orig.push_back((int)i);
    }
  const unsigned int n = (unsigned int)rops.size();
  if (n == 0)
  {
    // nothing but tip-tip ops
    k.nops = 0;
    return 1;
  }
  std::vector<FusedOp> fplan; // (the segments' plans one after the other; list_pos: position in rops)
  unsigned int reloads = 0;
  const FusedGeom geom = {nclv, nsc, c->sh.tips, c->sh.pattern_tip != 0};
  // Round 5: independent sub-lists as segments of the launch, (tile, segment) work items, while the tiles alone do
  // not fill the device's workgroup slots eight times over (partials_fused.hpp); PLLHIP_FUSED_SEGMENTS=0 / n: never /
  // up to n whatever the size.
  unsigned int max_segs = ((size_t)c->sh.sites + AF_WGS - 1) / AF_WGS < (size_t)c->num_cus * 2 * 8 ? PLLHIP_FUSED_MAX_SEGS : 1u;
  if (const char * e = pllhip_env("PLLHIP_FUSED_SEGMENTS")) max_segs = (unsigned int)std::max(1, atoi(e));
  std::vector<unsigned int> seg_of, seg_first, seg_n;
  const unsigned int nsegs = pllhip_fused_segments(geom, rops.data(), n, max_segs, seg_of);
  int rc = 0;
  if (nsegs == 1)
  {
    rc = pllhip_fused_plan(geom, rops.data(), rargs.data(), rkinds.data(), n, AF_NSLOT, fplan, &reloads);
    seg_first.push_back(0u);
    seg_n.push_back(n);
  }
  for (unsigned int sg = 0; nsegs > 1 && sg < nsegs && rc == 0; ++sg)
  {
    std::vector<pllhip_op_t> sops;
    std::vector<PartialsArgs> sargs;
    std::vector<int> skinds, where;
    for (unsigned int i = 0; i < n; ++i)
      if (seg_of[i] == sg)
      {
        sops.push_back(rops[i]);
        sargs.push_back(rargs[i]);
        skinds.push_back(rkinds[i]);
        where.push_back((int)i);
      }
    std::vector<FusedOp> part;
    unsigned int r = 0;
    rc = pllhip_fused_plan(geom, sops.data(), sargs.data(), skinds.data(), (unsigned int)sops.size(), AF_NSLOT, part, &r);
    reloads += r;
    seg_first.push_back((unsigned int)fplan.size());
    seg_n.push_back((unsigned int)part.size());
    for (FusedOp & f : part)
    {
      f.list_pos = where[f.list_pos];
      fplan.push_back(f);
    }
  }
  if (rc) return rc;
  lap("plan (order, slots)");

  // ---- encode
  if (!c->fused_zero_row)
  {
    const size_t bytes = (size_t)c->sh.sites + PLLHIP_TAIL_SITES + 256;
    HIP_TRY(hipMalloc((void **)&c->fused_zero_row, bytes));
    HIP_TRY(hipMemsetAsync(c->fused_zero_row, 0, bytes, c->stream));
  }
  if (!c->d_sink) HIP_TRY(hipMalloc(&c->d_sink, (size_t)c->num_cus * 16 * 80 * sizeof(double2)));
  if (!c->d_tile_counter) HIP_TRY(hipMalloc((void **)&c->d_tile_counter, 2 * PLLHIP_TILE_COUNTER_BYTES));
  k.lk_ops.clear();
  k.lk_k1.clear();
  k.lk_k2.clear();
  std::vector<int> lk_index(n, -1);
  for (unsigned int pos = 0; pos < n; ++pos)
  {
    const int ri = fplan[pos].list_pos, oi = orig[ri];
    if (kinds[oi] != 3) continue; // (a tip-tip op of the list needs no pair tables)
    lk_index[pos] = (int)k.lk_ops.size();
    k.lk_ops.push_back(args[oi]);
    if (lk_kids[oi].first >= 0) k.lk_k1.push_back(args[lk_kids[oi].first]);
    else
    {
      PartialsArgs none; // marks a tip-inner lookup op (no producing op on the tip's side)
      memset(&none, 0, sizeof(none));
      k.lk_k1.push_back(none);
    }
    k.lk_k2.push_back(args[lk_kids[oi].second]);
  }
  // the lookup tables' addresses are needed in the records: build them now (their pool may move)
  std::vector<AaLookupTables> tabs(k.lk_ops.size());
  // (round 4: the tables are made by k_af_prepare, AaLookupJob; PLLHIP_AA_LOOKUP_DIRECT=0: by launches of the
  // tabulating kernels ahead of it, as in round 3)
  const bool lk_direct = !(pllhip_env("PLLHIP_AA_LOOKUP_DIRECT") && atoi(pllhip_env("PLLHIP_AA_LOOKUP_DIRECT")) == 0);
  std::vector<AaLookupJob> lj(lk_direct ? 2 * k.lk_ops.size() : 0);
  if (!k.lk_ops.empty())
  {
    rc = pllhip_aa_lookup_tables(c, k.lk_ops.data(), k.lk_k1.data(), k.lk_k2.data(), (unsigned int)k.lk_ops.size(),
                                 tabs.data(), lk_direct ? lj.data() : nullptr);
    // (the pool could not be allocated: once more, now without lookup ops -- the context remembers)
    if (rc == 1 && c->cherry_pool_failed) return aa_fused_update(c, ops, count, tt_wanted);
    if (rc) return rc;
  }
  lap("lookup tables (launches)");
  std::vector<AaRec> recs; // per segment: the header, one record per op, the record that names op 0 again
  std::vector<unsigned int> segtab;
  std::vector<AfMatJob> mj;
  std::vector<AfTipJob> tj;
  std::vector<AfPairJob> pj;
  std::vector<size_t> tt_pair_rec, tt_inside_rec, op_rec; // (absolute record numbers)
  unsigned int synced = 0;
  const size_t tip_tab_b = (size_t)c->maxstates * 80 * sizeof(double);
  const size_t pair_tab_b = (size_t)c->maxstates * tip_tab_b;
  // (the pool of the pair tables shares the lookup tables' budget: what the lookups of this list left of it)
  const bool pairs_on = !(pllhip_env("PLLHIP_AA_TT_PAIRS") && atoi(pllhip_env("PLLHIP_AA_TT_PAIRS")) == 0);
  const size_t lookup_tab_b = 4 * ((size_t)c->maxstates * c->maxstates + PLLHIP_TAIL_SITES) * 80 * sizeof(double);
  const size_t pair_budget_b = !pairs_on ? 0 : (size_t)(lookups_max > lookups ? lookups_max - lookups : 0) * lookup_tab_b;
  const unsigned long long zero_row = (unsigned long long)(uintptr_t)c->fused_zero_row;
  k.mode = !any_scaler ? SCALE_NONE : (c->sh.rate_scalers ? SCALE_RATE : SCALE_SITE);
  auto slot4 = [](int s) { return (unsigned int)(s > 0 ? s : 0) & 15u; };
  auto reloads_of = [&](AaRec & r, const FusedOp & f) {
    // what is done during the op before `f` for it
    if (f.dma_flags & 1)
    {
      r.flags |= AF_RELOAD_A;
      r.ra_src = (unsigned long long)(uintptr_t)f.left_hbm;
      r.ra_cnt = (unsigned long long)(uintptr_t)f.lsc_hbm;
      r.slots |= slot4(f.lslot) << 12;
    }
    if (f.dma_flags & 2)
    {
      r.flags |= AF_RELOAD_B;
      r.rb_src = (unsigned long long)(uintptr_t)f.right_hbm;
      r.rb_cnt = (unsigned long long)(uintptr_t)f.rsc_hbm;
      r.slots |= slot4(f.rslot) << 16;
    }
  };
  for (unsigned int sg = 0; sg < nsegs; ++sg)
  {
  const unsigned int first = seg_first[sg], m = seg_n[sg];
  const size_t base = recs.size();
  segtab.push_back((unsigned int)base);
  segtab.push_back(m);
  recs.resize(base + m + 2);
  AaRec * const R = recs.data() + base;
  memset(R, 0, (m + 2) * sizeof(AaRec));
  reloads_of(R[0], fplan[first]);
  // what each op brings along itself: its rows, its left block, its kind -- recorded with the op BEFORE
  // it (the header for op 0; the last op names op 0 again: the segment is walked tile after tile)
  std::vector<AaRec> own(m);
  memset(own.data(), 0, m * sizeof(AaRec));
  for (unsigned int pos = 0; pos < m; ++pos)
  {
    const FusedOp & f = fplan[first + pos];
    const int oi = orig[f.list_pos];
    const bool tt_op = kinds[oi] == 4;
    const int kind = kinds[oi] >= 3 ? 2 : kinds[oi];
    AaRec & r = R[pos + 1];
    AaRec & o = own[pos];
    op_rec.push_back(base + pos + 1);
    r.parent = (unsigned long long)(uintptr_t)f.parent;
    r.pscaler = (unsigned long long)(uintptr_t)f.pscaler;
    r.flags = (unsigned int)kind;
    o.flags = (unsigned int)kind;
    if (f.pslot >= 0) r.flags |= AF_HAS_PSLOT;
    if (f.pscaler) r.flags |= tt_op ? AF_ZERO_COUNTS : AF_SCALING;
    if (f.pscaler && op_inexact[oi] && kind <= 1) r.flags |= AF_CERT;
    if (kind == 0 && f.lsc_slot >= 0) r.flags |= AF_LCNT;
    if (kind <= 1 && f.rsc_slot >= 0) r.flags |= AF_RCNT;
    r.slots = slot4(f.lslot) | slot4(f.rslot) << 4 | slot4(f.pslot) << 8;
    for (int t = 0; t < 4; ++t) o.row[t] = zero_row;
    if (kind == 0)
    {
      o.xoff = (unsigned int)(mj.size() * AF_MAT_B);
      mj.push_back(AfMatJob{f.lmat, (unsigned long long)o.xoff, 0ull});
    }
    if (kind <= 1)
    {
      r.yoff = (unsigned int)(mj.size() * AF_MAT_B);
      mj.push_back(AfMatJob{f.rmat, (unsigned long long)r.yoff, (kind == 1 && !list_ti_mfma) ? 1ull : 0ull});
      if (kind == 1 && list_ti_mfma) r.flags |= AF_TI_MFMA;
    }
    if (kind == 1)
    {
      o.row[0] = (unsigned long long)(uintptr_t)f.ltip;
      r.tab_l = (unsigned long long)(tj.size() * tip_tab_b); // (made absolute below)
      tj.push_back(AfTipJob{f.lmat, (unsigned long long)(tj.size() * tip_tab_b)});
    }
    if (tt_op && (pj.size() + 1) * pair_tab_b <= pair_budget_b)
    {
      // ONE table over the character pairs (AfPairJob; offset made absolute below): row c1 ms + c2
      r.tab_l = (unsigned long long)(pj.size() * pair_tab_b);
      pj.push_back(AfPairJob{args[oi].lmat, args[oi].rmat, (unsigned long long)(pj.size() * pair_tab_b)});
      r.flags |= AF_ONE_TABLE;
      o.row[0] = (unsigned long long)(uintptr_t)args[oi].ltip;
      o.row[1] = (unsigned long long)(uintptr_t)args[oi].rtip;
      tt_pair_rec.push_back(base + pos + 1);
    }
    else if (tt_op)
    {
      // (beyond the pool's budget: the two tip tables, multiplied per site.  Offsets into the tip tables, made
      // absolute below; "pair" (0, character) is row `character`)
      r.tab_l = (unsigned long long)(tj.size() * tip_tab_b);
      tj.push_back(AfTipJob{args[oi].lmat, (unsigned long long)(tj.size() * tip_tab_b)});
      r.tab_r = (unsigned long long)(tj.size() * tip_tab_b);
      tj.push_back(AfTipJob{args[oi].rmat, (unsigned long long)(tj.size() * tip_tab_b)});
      o.row[1] = (unsigned long long)(uintptr_t)args[oi].ltip;
      o.row[3] = (unsigned long long)(uintptr_t)args[oi].rtip;
      tt_inside_rec.push_back(base + pos + 1);
    }
    else if (kind == 2)
    {
      const AaLookupTables & t = tabs[lk_index[first + pos]];
      r.tab_l = (unsigned long long)(uintptr_t)t.tl;
      r.tab_r = (unsigned long long)(uintptr_t)t.tr;
      o.row[0] = (unsigned long long)(uintptr_t)t.t1;
      o.row[1] = (unsigned long long)(uintptr_t)t.t2;
      o.row[2] = (unsigned long long)(uintptr_t)t.t3;
      o.row[3] = (unsigned long long)(uintptr_t)t.t4;
    }
    if (pos + 1 < m) reloads_of(r, fplan[first + pos + 1]);
  }
  // (opt-in since the end of round 6, PLLHIP_AA_KEEP=1: with the planner's walk fixed hardly a list copies a value of its
  // own back any more -- one of 398 ops of a 400-taxon random tree -- and the kernel instance that looks for such values
  // costs a list 1.9 %: 3,727-3,751 us with it against 3,686-3,692 without, profiles/r6_aa_keep_instance_ab.txt)
  if (pllhip_env("PLLHIP_AA_KEEP") && atoi(pllhip_env("PLLHIP_AA_KEEP")) != 0)
    for (unsigned int pos = 0; pos < m; ++pos)
    {
      // (the writer of a value that op `pos` reloads, if it is an earlier op of this segment)
      const FusedOp & f = fplan[first + pos];
      for (const double * src : {(f.dma_flags & 1) ? f.left_hbm : nullptr, (f.dma_flags & 2) ? f.right_hbm : nullptr})
        for (unsigned int w = 0; src && w < pos; ++w)
          if (fplan[first + w].parent == src)
          {
            R[w + 1].flags |= AF_KEEP;
            k.any_keep = true;
          }
    }
  // The left block of op i is staged by the four waves, a part each, while they run op i - 2, and the barrier that
  // tells a wave that everybody's part has landed is barrier A of op i - 1 -- which a lookup does not have.  Until
  // the tip-tip ops joined the list (runs of tens of barrier-free ops, over which the waves drift apart by whole
  // ops) this went unnoticed: an inner-inner op behind a lookup then read its left block a few hundred cycles
  // after the waves had last met.  Such an op now begins with a barrier of its own.
  for (unsigned int pos = 0; pos < m; ++pos)
    if ((own[pos].flags & AF_KIND_MASK) == 0u && (own[(pos + m - 1) % m].flags & AF_KIND_MASK) == 2u)
    {
      R[pos + 1].flags |= AF_SYNC_LEFT;
      ++synced;
    }
  for (unsigned int pos = 0; pos <= m; ++pos)
  {
    // R[pos] is the record before op `pos` (R[0]: the header; R[m] names op 0 again)
    const AaRec & o = own[pos == m ? 0 : pos];
    AaRec & r = R[pos];
    r.xoff = o.xoff;
    r.flags |= (o.flags & AF_KIND_MASK) << 8;
    // (rows: those of the op after next -- the header names op 0's)
    const AaRec & o2 = pos == 0 ? own[0] : own[(pos + 1) % m];
    for (int t = 0; t < 4; ++t) r.row[t] = o2.row[t];
    // the link: the record of the op that runs after this record's (the header's: op 0's; the last op's: op 0's again)
    r.flags |= (unsigned int)(base + (pos == m ? 1u : pos + 1u)) << AF_NEXT_SHIFT;
    if (pos == m) r.flags |= AF_LAST;
  }
  R[0].yoff = (unsigned int)(base + m); // (the segment's last record: the prologue takes op 1's rows from it)
  // (the scaling certificate's rare path reads these two from the header: the flag's address, the window)
  R[0].parent = (unsigned long long)(uintptr_t)c->h_cert_dev;
  {
    const double w = PLLHIP_SCALE_THRESHOLD * cert_window;
    memcpy(&R[0].pscaler, &w, sizeof(w));
  }
  } // segments
  if (recs.size() >= (1u << (32 - AF_NEXT_SHIFT))) return 1;
  if ((mj.size() + 1) * (size_t)AF_MAT_B > 0xffffffffull) return 1;
  // buffers: matrices in operand order, tip tables
  if (k.aorder_cap < (mj.size() + 1) * (size_t)AF_MAT_B)
  {
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (k.d_aorder) HIP_TRY(hipFree(k.d_aorder));
    k.d_aorder = nullptr;
    k.aorder_cap = (mj.size() + 1) * (size_t)AF_MAT_B * 2;
    HIP_TRY(hipMalloc((void **)&k.d_aorder, k.aorder_cap));
  }
  if (k.titab_cap < (tj.size() + 1) * tip_tab_b)
  {
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (k.d_titab) HIP_TRY(hipFree(k.d_titab));
    k.d_titab = nullptr;
    k.titab_cap = (tj.size() + 1) * tip_tab_b * 2;
    HIP_TRY(hipMalloc((void **)&k.d_titab, k.titab_cap));
  }
  if (k.pairtab_cap < (pj.size() + 1) * pair_tab_b)
  {
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (k.d_pairtab) HIP_TRY(hipFree(k.d_pairtab));
    k.d_pairtab = nullptr;
    k.pairtab_cap = (pj.size() + 1) * pair_tab_b;
    HIP_TRY(hipMalloc((void **)&k.d_pairtab, k.pairtab_cap));
  }
  for (size_t ri : tt_pair_rec) recs[ri].tab_l += (unsigned long long)(uintptr_t)k.d_pairtab;
  for (size_t ri : op_rec)
    if ((recs[ri].flags & AF_KIND_MASK) == 1u) recs[ri].tab_l += (unsigned long long)(uintptr_t)k.d_titab;
  for (size_t ri : tt_inside_rec)
  {
    recs[ri].tab_l += (unsigned long long)(uintptr_t)k.d_titab;
    recs[ri].tab_r += (unsigned long long)(uintptr_t)k.d_titab;
  }

  const size_t rec_b = recs.size() * sizeof(AaRec), mat_b = (mj.size() + 1) * sizeof(AfMatJob),
               tip_b = (tj.size() + 1) * sizeof(AfTipJob), pair_b = (pj.size() + 1) * sizeof(AfPairJob),
               lk_b = (lj.size() + 1) * sizeof(AaLookupJob), seg_b = 2 * PLLHIP_FUSED_MAX_SEGS * sizeof(unsigned int);
  const size_t bytes = rec_b + mat_b + tip_b + pair_b + lk_b + seg_b;
  if (k.plan_cap < bytes)
  {
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (k.d_plan) HIP_TRY(hipFree(k.d_plan));
    if (k.h_plan) HIP_TRY(hipHostFree(k.h_plan));
    k.d_plan = k.h_plan = nullptr;
    k.plan_cap = bytes * 2;
    HIP_TRY(hipMalloc(&k.d_plan, k.plan_cap));
    HIP_TRY(hipHostMalloc(&k.h_plan, k.plan_cap, hipHostMallocDefault));
    if (!k.done) HIP_TRY(hipEventCreateWithFlags(&k.done, hipEventDisableTiming));
    k.pending = false;
  }
  if (k.pending) HIP_TRY(hipEventSynchronize(k.done));
  char * stage = static_cast<char *>(k.h_plan);
  memcpy(stage, recs.data(), rec_b);
  if (!mj.empty()) memcpy(stage + rec_b, mj.data(), mj.size() * sizeof(AfMatJob));
  if (!tj.empty()) memcpy(stage + rec_b + mat_b, tj.data(), tj.size() * sizeof(AfTipJob));
  if (!pj.empty()) memcpy(stage + rec_b + mat_b + tip_b, pj.data(), pj.size() * sizeof(AfPairJob));
  if (!lj.empty()) memcpy(stage + rec_b + mat_b + tip_b + pair_b, lj.data(), lj.size() * sizeof(AaLookupJob));
  memcpy(stage + rec_b + mat_b + tip_b + pair_b + lk_b, segtab.data(), segtab.size() * sizeof(unsigned int));
  HIP_TRY(hipMemcpyAsync(k.d_plan, k.h_plan, bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipEventRecord(k.done, c->stream));
  k.pending = true;
  k.off_mat = rec_b;
  k.off_tip = rec_b + mat_b;
  k.off_pair = rec_b + mat_b + tip_b;
  k.off_lk = rec_b + mat_b + tip_b + pair_b;
  k.off_seg = rec_b + mat_b + tip_b + pair_b + lk_b;
  k.nsegs = nsegs;
  k.nlk = (unsigned int)lj.size();
  k.npair = (unsigned int)pj.size();
  k.nmat = (unsigned int)mj.size();
  k.ntip = (unsigned int)tj.size();
  k.nops = n;
  {
    unsigned int n_ii = 0, n_ti = 0;
    for (unsigned int i = 0; i < count; ++i)
    {
      n_ii += kinds[i] == 0;
      n_ti += kinds[i] == 1;
    }
    const unsigned int v[8] = {count, (unsigned int)k.tt_ops.size(), (unsigned int)(tt_inside_rec.size() + tt_pair_rec.size()),
                               (unsigned int)k.lk_ops.size(), n_ii, list_ti_mfma ? n_ti : 0u, list_ti_mfma ? 0u : n_ti, reloads};
    memcpy(k.kinds_of_plan, v, sizeof(v));
  }
  if (c->fused_debug)
  {
    fprintf(stderr, "pllhip 20-state list kernel: %u ops = %zu tip-tip ahead + %zu tip-tip in the list + %zu lookups + %zu on the matrix cores "
                    "(%u of them behind a lookup: a barrier more), %u operands reloaded, %u segment(s)\n",
            count, k.tt_ops.size(), tt_inside_rec.size() + tt_pair_rec.size(), k.lk_ops.size(),
            (size_t)n - k.lk_ops.size() - tt_inside_rec.size() - tt_pair_rec.size(), synced, reloads, nsegs);
  }
  lap("encode + upload");
  rc = aa_fused_launch(c, true);
  if (rc) return rc;
  lap("launches");
  k.last_ops.assign(ops, ops + count);
  k.epoch = c->layout_epoch;
  k.maxstates = c->maxstates;
  return 0;
}
