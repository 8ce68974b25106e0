// partials_fused.hip -- a whole op list of 4-state CLV updates in ONE kernel, site-blocked.
//
// Replaces the per-level launches of partials.hip for pll_update_partials
// (partials.c:214-278 -> core_partials.c) when the list has more than one op.
//
// Observation.  A CLV update at site n needs the children's entries of site n only, so
// the order "all sites of op 1, all sites of op 2, ..." of the reference is a choice, not
// a dependency: a wave may just as well take a TILE of sites through the whole list.
// What it wrote for a child a moment ago is then still on chip -- and in the lane mapping
// of k_dna_partials (one lane per 16 bytes = two states of a (site, rate) element, lane
// pairs joined by DPP) every lane needs exactly the 16 bytes IT wrote for that child.
// So a parent's tile is stored to HBM (every CLV remains a result of the call) and kept
// in a wave-private LDS slot together with its scaler counts, from which the op that
// consumes it reads it back; no synchronisation of any kind.  HBM traffic per site-update
// drops from 396 B (read two children, write the parent) to 132 B + tip characters: the
// list becomes a WRITE stream.
//
//   tile      J sub-steps of 64 lanes x 16 B (J x 64 / (2 rate_cats) sites)
//   slots     tiles per wave in LDS (6 x 2.1 KB for 4 rate categories on 12 waves per CU, 7 on 8); the
//             host assigns them: a value keeps its slot until its last reader in the list has run
//   op order  any order that respects the list's read/write hazards on CLV and scale
//             buffer indices is equivalent; the host re-orders the list depth-first,
//             heavier subtree first (Sethi-Ullman), which bounds the number of live
//             values by the tree's Strahler number (5 for a balanced 64-taxon tree, 6 for 128
//             taxa, not the 32 / 64 of a level-by-level list)
//   reload    EVERY inner operand is read from a slot.  A value that had to give its slot up
//             (random trees: a handful per list), and every operand written by an earlier call
//             (partial traversals; tip CLVs), is copied from HBM into a slot by LDS-DMA
//             (global_load_lds: no registers) at the top of the op BEFORE its reader.
//   look-ahead  vector-memory loads and stores retire in ONE in-order queue, so a load
//             issued after a store cannot be consumed before that store is acknowledged.
//             Everything an op needs from memory (its two P-matrices, tip characters) is
//             therefore requested at the top of the op two before it, ahead of that op's stores,
//             with unconditional loads (absent operands read a zero block); the op itself
//             contains LDS traffic and stores only.
//   plan      32 bytes per op (indices, not pointers), read through the scalar data cache, a whole
//             record three ops ahead: see FusedRec in partials_fused.hpp for why it is that small.
//
// The arithmetic per site is that of k_dna_partials, statement for statement (dot4 /
// masksum4 order, scaling rule of core_partials_avx.c:486-527), so results are
// bit-identical to the per-level path; tests/test_gpu_parity.py and the random-op-sequence
// tests run through this kernel.
//
// Roofline: HBM writes.  132 B per site-update (128 B CLV + 4 B scaler count) + 1 B per
// tip character read; operands that are reloaded add 128 B each.
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "ctx.hpp"
#include "numerics.hpp"
#include "partials_fused.hpp"

#define PLL_LDS __attribute__((address_space(3)))

// ---- a plan record, word by word (all of this is scalar arithmetic on wave-uniform values) ----
typedef const unsigned int __attribute__((address_space(4))) * const_words;
typedef const unsigned long long __attribute__((address_space(4))) * const_quads;
struct RecAhead // words 0, 1 of a FusedRec: what request() needs
{
  unsigned int w0, w1;
};
struct RecOp // words 4..7: what the op and the op before it need
{
  unsigned int w4, w5, w6, w7;
};
// (constant address space: scalar loads; adjacent words, the compiler merges them)
__device__ __forceinline__ RecAhead rec_ahead(const FusedRec * plan, unsigned int i)
{
  const const_words w = (const_words)(unsigned long long)(plan + i);
  RecAhead r;
  r.w0 = w[0]; r.w1 = w[1];
  return r;
}
__device__ __forceinline__ RecOp rec_op(const FusedRec * plan, unsigned int i)
{
  const const_words w = (const_words)(unsigned long long)(plan + i);
  RecOp r;
  r.w4 = w[4]; r.w5 = w[5]; r.w6 = w[6]; r.w7 = w[7];
  return r;
}
__device__ __forceinline__ unsigned int rec_ltip(const RecAhead & r) { return r.w0 & 0xffffu; }
__device__ __forceinline__ unsigned int rec_rtip(const RecAhead & r) { return r.w0 >> 16; }
__device__ __forceinline__ unsigned int rec_lmat(const RecAhead & r) { return r.w1 & 0xffffu; }
__device__ __forceinline__ unsigned int rec_rmat(const RecAhead & r) { return r.w1 >> 16; }
__device__ __forceinline__ unsigned int rec_parent(const RecOp & r) { return r.w4 & 0xffffu; }
__device__ __forceinline__ unsigned int rec_pscaler(const RecOp & r) { return r.w4 >> 16; }
__device__ __forceinline__ unsigned int rec_pair(const RecOp & r) { return r.w5 & 0xffffu; }
__device__ __forceinline__ unsigned int rec_src(const RecOp & r) { return r.w5 >> 16; }
__device__ __forceinline__ int rec_lslot(const RecOp & r) { return (int)(r.w6 << 24) >> 24; }
__device__ __forceinline__ int rec_rslot(const RecOp & r) { return (int)(r.w6 << 16) >> 24; }
__device__ __forceinline__ int rec_pslot(const RecOp & r) { return (int)(r.w6 << 8) >> 24; }
__device__ __forceinline__ int rec_kind(const RecOp & r) { return (int)(r.w6 >> 24); }
__device__ __forceinline__ int rec_lsc(const RecOp & r) { return (int)(r.w7 << 24) >> 24; }
__device__ __forceinline__ int rec_rsc(const RecOp & r) { return (int)(r.w7 << 16) >> 24; }
__device__ __forceinline__ unsigned int rec_dma(const RecOp & r) { return (r.w7 >> 16) & 0xffu; }

// Tip operands: the parent entry of a tip-tip op depends on its two tip characters only, 16 x 16
// pairs.  One table per such op, [pair][rate][state] = masksum4(P_l row, code 1) *
// masksum4(P_r row, code 2) -- the very product the kernel would form per site (30 VALU
// instructions per tip operand and sub-step) -- built by one small launch ahead of the
// list and read back by the list kernel with one 16-byte gather per lane and sub-step
// (32 KB per op at 4 rate categories: L2-resident).
template <int RC>
__global__ __launch_bounds__(256) void k_dna_pair_tables(const FusedRec * __restrict__ plan, unsigned int nops,
                                                         FusedBases b)
{
  const unsigned int i = blockIdx.x;
  if (i >= nops || plan[i].pair == PLLHIP_FUSED_NONE) return;
  const double * lm = b.pmat + (size_t)plan[i].lmat * (RC * 16), * rm = b.pmat + (size_t)plan[i].rmat * (RC * 16);
  double * tab = b.pairtab + (size_t)plan[i].pair * (256 * RC * 4);
  const unsigned int pair = threadIdx.x, c1 = pair >> 4, c2 = pair & 15u;
  // (tip-inner ops: the tip's factor alone, in the entries [code 1][0] the kernel's index
  // (code 1 << 4 | character of an absent tip = 0) reaches; x * 1.0 is x)
  const bool tt = plan[i].kind == 2;
  for (unsigned int ki = 0; ki < RC * 4u; ++ki)
    tab[pair * RC * 4u + ki] = masksum4(lm + ki * 4u, c1) * (tt ? masksum4(rm + ki * 4u, c2) : 1.0);
}

// what a lane requests for an op ahead of its use
template <int PL, int J>
struct FusedFetch
{
  double2 pm[PL];                      // its 16 bytes of the two P-matrices (a coalesced block per wave)
  unsigned int codes_l[J], codes_r[J]; // tip characters of the lane's own site in each sub-step
  // element by element: a plain struct assignment of the arrays goes through scratch memory
  __device__ __forceinline__ void take(const FusedFetch & o)
  {
#pragma unroll
    for (int t = 0; t < PL; ++t) pm[t] = o.pm[t];
#pragma unroll
    for (int j = 0; j < J; ++j)
    {
      codes_l[j] = o.codes_l[j];
      codes_r[j] = o.codes_r[j];
    }
  }
};

// WPS: waves per SIMD the register budget is sized for (3 = 168 VGPRs: twelve waves per CU)
template <int RC, int J, int MODE, bool NT, int WPS>
__global__ __launch_bounds__(256, WPS) void k_dna_fused(const FusedRec * __restrict__ plan, FusedBases bases,
                                                        unsigned int nops, unsigned int sites, unsigned int nslots,
                                                        const unsigned int * __restrict__ zero, double2 * sink,
                                                        unsigned int * next_tile, unsigned int backwards, unsigned int dynamic_rounds)
{
  constexpr unsigned int W = 2 * RC, SPS = 64 / W, TS = J * SPS;
  constexpr unsigned int MG = RC * 8;                   // 16-byte granules of one P-matrix
  constexpr int PL = (2 * MG + 63) / 64;                // granules of [P_l | P_r] per lane
  extern __shared__ double2 lds_fused[];
  const unsigned int lane = threadIdx.x & 63u;
  const unsigned int h = lane & 1u;
  const unsigned int k = (lane >> 1) & (RC - 1);
  const unsigned int wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // per wave: [nslots][J][64] CLV granules | [MG] matrix granules (one matrix at a time) | counts
  // (counts: one word per site, or per (site, rate) with per-rate scalers, of a sub-step)
  constexpr unsigned int CW = (MODE == SCALE_RATE) ? 32u : (SPS < 4u ? 4u : SPS); // words per sub-step, 16-byte multiple
  const size_t wave_g = (size_t)nslots * J * 64 + MG + (size_t)nslots * J * CW / 4;
  double2 * clv = lds_fused + wave_in_wg * wave_g;
  double2 * pst = clv + (size_t)nslots * J * 64;
  unsigned int * cnt = reinterpret_cast<unsigned int *>(pst + MG);
  // the same places as LDS byte addresses (what the LDS-DMA of reload() takes in M0); computed
  // from the array itself: casting a derived generic pointer back to LDS makes the compiler
  // emit a null check it cannot always encode
  const unsigned int clv_lds_b = __builtin_amdgcn_readfirstlane(
      (unsigned int)(uintptr_t)(PLL_LDS char *)lds_fused + (unsigned int)(wave_in_wg * wave_g * 16));
  const unsigned int cnt_lds_b = clv_lds_b + (unsigned int)(((size_t)nslots * J * 64 + MG) * 16);
  const double2 * zero16 = reinterpret_cast<const double2 *>(zero);
  const unsigned char * zero8 = reinterpret_cast<const unsigned char *>(zero);

  // Every CLV, scale buffer and tip row has PLLHIP_TAIL_SITES sites of slack of its own
  // behind its last site (ctx.hip), so the last tile is loaded and STORED whole: no lane
  // predicates, and every address is a wave-uniform tile base plus a lane offset that
  // never changes.
  const size_t tiles = ((size_t)sites + TS - 1) / TS;
  // Tile = wave number: the four waves of a workgroup take four adjacent tiles, and neighbouring
  // workgroups -- which the dispatcher deals to the eight XCDs in turn -- the next four.  Giving
  // each XCD a region of its own instead (a contiguous eighth, or chunks of 16 ... 4096 tiles
  // dealt round-robin, so that a 2 MB page is written by one XCD only) was measured on the
  // 62-, 126- and 198-op lists and is slower the larger the chunk: 0.62 / 0.50 / 0.49 of the
  // HBM peak with this mapping, 0.51 / 0.50 / 0.48 with chunks of 16 tiles, 0.46 / 0.46 / 0.45
  // with 1024 (profiles/r2_xcd_tile_mapping.txt).
  const size_t wave = (size_t)blockIdx.x * 4u + wave_in_wg;
  const size_t nwaves = (size_t)gridDim.x * 4u;
  // every wave has 1280 bytes of sink of its own: thousands of waves storing to ONE block
  // serialise on its cache lines (measured: ~0.7 ms per launch at 1 M sites, hidden behind a
  // 62-op list but not behind a 5-op one)
  sink += wave * 80;

  // A wave's first tiles are its own by a fixed stride; the last `dynamic_rounds` rounds' worth come
  // from a counter.  Not only for the tail: the eight XCDs do not write at the same rate -- on
  // every box measured the odd-numbered ones take ~20 % longer for the same tiles
  // (tools/xcd_balance_bench.hip: last workgroup of XCDs 0/2/4/6 done at 0.92 ms, of 1/3/5/7
  // at 1.13 ms) -- so with equal shares the fast half of the chip idles at the end.  Seven
  // rounds of ~20 from the counter let it take the difference (two rounds, round 1's choice,
  // covered the tail only): 62 / 126 / 198-op lists 0.616 / 0.526 / 0.514 -> 0.635 / 0.580 / 0.546 of
  // the HBM peak (same box).  All tiles from the counter is slower again (atomics on one
  // address serialise at ~12 ns: 0.75 ms of them per launch at 1 M sites), and short lists keep
  // two rounds for that reason.
  size_t static_rounds = tiles / nwaves > dynamic_rounds ? tiles / nwaves - dynamic_rounds : 1; // (the last rounds from the counter)
  if (!next_tile) static_rounds = ~(size_t)0;
  size_t round = 0;
  for (size_t tile = wave; tile < tiles;)
  {
    // (every other launch of a context walks the tiles from the far end: see pllhip_relaunch_fused)
    const size_t site0 = (backwards ? tiles - 1 - tile : tile) * TS;

    // every load is unconditional (absent operands read the zero block): a load inside a
    // branch makes the compiler wait for everything in flight
    auto request = [&](FusedFetch<PL, J> & f, const RecAhead & r) {
      const double2 * lm = reinterpret_cast<const double2 *>(bases.pmat + (size_t)rec_lmat(r) * (RC * 16));
      const double2 * rm = reinterpret_cast<const double2 *>(bases.pmat + (size_t)rec_rmat(r) * (RC * 16));
#pragma unroll
      for (int t = 0; t < PL; ++t)
      {
        const unsigned int q = lane + 64u * t;
        const double2 * src = (q < MG) ? lm + q : (q < 2 * MG) ? rm + (q - MG) : zero16;
        f.pm[t] = *src;
      }
      const bool has_l = rec_ltip(r) != PLLHIP_FUSED_NONE, has_r = rec_rtip(r) != PLLHIP_FUSED_NONE;
      // (uniform: tile base or the zero block)
      const unsigned char * lt = has_l ? bases.tips + (size_t)rec_ltip(r) * bases.tip_stride + site0 : zero8;
      const unsigned char * rt = has_r ? bases.tips + (size_t)rec_rtip(r) * bases.tip_stride + site0 : zero8;
#pragma unroll
      for (unsigned int j = 0; j < J; ++j)
      {
        f.codes_l[j] = lt[has_l ? j * SPS + lane / W : 0u];
        f.codes_r[j] = rt[has_r ? j * SPS + lane / W : 0u];
      }
    };

    // the op's matrices: the wave's coalesced block goes through LDS (the previous op's rows are in registers by then),
    // each lane takes rows 2h, 2h+1 of category k (own column pair first, then the partner's).
    // One matrix at a time through ONE staging block (LDS operations of a wave execute in
    // order): the 512 bytes this saves per wave are what gives the 12-wave configuration
    // its sixth slot (a balanced 128-taxon tree needs six).
    auto stage_rows = [&](const FusedFetch<PL, J> & f, half_rows & pl, half_rows & pr) {
      double2 * p = pst;
#pragma unroll
      for (int t = 0; t < PL; ++t)
        if (lane + 64u * t < MG) p[lane + 64u * t] = f.pm[t];
      // (what a lane reads was written by OTHER lanes: the compiler, which reasons per
      // thread, must not carry a value over these lines -- it once served the lanes that do
      // not write in the second round with their first-round rows)
      asm volatile("" ::: "memory");
#pragma unroll
      for (int r = 0; r < 2; ++r)
      {
        const unsigned int row = k * 8 + (2 * h + r) * 2;
        const double2 lo = p[row + h], lp = p[row + 1 - h];
        pl.m[r][0] = lo.x; pl.m[r][1] = lo.y; pl.m[r][2] = lp.x; pl.m[r][3] = lp.y;
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int t = 0; t < PL; ++t)
        if (lane + 64u * t >= MG && lane + 64u * t < 2 * MG) p[lane + 64u * t - MG] = f.pm[t];
      asm volatile("" ::: "memory");
#pragma unroll
      for (int r = 0; r < 2; ++r)
      {
        const unsigned int row = k * 8 + (2 * h + r) * 2;
        const double2 ro = p[row + h], rp = p[row + 1 - h];
        pr.m[r][0] = ro.x; pr.m[r][1] = ro.y; pr.m[r][2] = rp.x; pr.m[r][3] = rp.y;
      }
    };

    // Reload: an operand that has no slot -- a value that gave its slot up, or one written by
    // an earlier call -- is copied from HBM straight into the slot the plan names for it, by
    // LDS-DMA (global_load_lds: no registers), at the top of the op BEFORE its reader.  The
    // instructions are inline assembly on purpose: the compiler does not count them, so they
    // cost no wait of their own -- they are issued ahead of that iteration's look-ahead
    // loads, memory operations return in order, and the next iteration's first statement
    // waits for those loads before any slot is read.  (Per-lane 64-bit addresses and `off`:
    // the form the compiler itself emits for the builtin.)
    auto reload = [&](const RecOp & r) {
      // (the sources follow the records and their three look-ahead copies)
      const const_quads q = (const_quads)(unsigned long long)(reinterpret_cast<const FusedSrc *>(plan + nops + 3) + rec_src(r));
      const double * src[2] = {(const double *)q[0], (const double *)q[1]};
      const unsigned int * csrc[2] = {(const unsigned int *)q[2], (const unsigned int *)q[3]};
      const int slot[2] = {rec_lslot(r), rec_rslot(r)};
#pragma unroll
      for (int o = 0; o < 2; ++o)
      {
        if (src[o])
        {
          const double2 * base = reinterpret_cast<const double2 *>(src[o]) + site0 * W;
#pragma unroll
          for (unsigned int j = 0; j < J; ++j)
          {
            const unsigned int lds_b = clv_lds_b + ((unsigned int)slot[o] * J + j) * 1024u;
            const double2 * gsrc = base + j * 64u + lane;
            unsigned int m0_saved;
            if (NT)
              asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                           "global_load_lds_dwordx4 %2, off nt\n\ts_mov_b32 m0, %0"
                           : "=&s"(m0_saved) : "s"(lds_b), "v"(gsrc) : "memory");
            else
              asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                           "global_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                           : "=&s"(m0_saved) : "s"(lds_b), "v"(gsrc) : "memory");
          }
        }
        if (MODE != SCALE_NONE && csrc[o])
        {
          // the tile's counts are J * CW consecutive words: the first so many lanes move one each
          const unsigned int * cbase = csrc[o] + ((MODE == SCALE_RATE) ? site0 * RC : site0);
          const unsigned int lds_b = cnt_lds_b + (unsigned int)slot[o] * (J * CW * 4u);
          const unsigned long long mask = (J * CW >= 64u) ? ~0ull : ((1ull << (J * CW)) - 1ull);
          const unsigned int * gsrc = cbase + lane;
          unsigned long long exec_saved;
          unsigned int m0_saved;
          asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 %1, m0\n\ts_mov_b64 exec, %2\n\ts_mov_b32 m0, %3\n\t"
                       "s_nop 0\n\tglobal_load_lds_dword %4, off\n\ts_mov_b32 m0, %1\n\ts_mov_b64 exec, %0"
                       : "=&s"(exec_saved), "=&s"(m0_saved) : "s"(mask), "s"(lds_b), "v"(gsrc) : "memory");
        }
      }
    };
    auto pair_table = [&](const RecOp & r) -> const double2 * {
      return rec_pair(r) != PLLHIP_FUSED_NONE
                 ? reinterpret_cast<const double2 *>(bases.pairtab + (size_t)rec_pair(r) * (256 * RC * 4))
                 : nullptr;
    };

    // Two ops of look-ahead: at the top of op i the block of op i+2 is requested, the block
    // of op i+1 (requested one op ago) goes through LDS into the registers op i+1 will use,
    // and op i runs on registers filled one op ago.  The plan records of ops i .. i+2 are in
    // SGPRs when they are needed: the op's own words for ops i and i+1 (r0, r1; those of op i+2
    // are requested at the top of op i), the look-ahead words of op i+2 (a2; op i+3's requested
    // at the top of op i).  (Scalar loads return out of order and share a counter with LDS, so
    // a plan field consumed right after its load would drain the LDS reads in flight.)  The
    // plan carries three copies of the last op behind it.
    RecOp r0 = rec_op(plan, 0), r1 = rec_op(plan, 1);
    RecAhead a2 = rec_ahead(plan, 2);
    FusedFetch<PL, J> cur, fa;
    half_rows pl, pr;
    if (rec_dma(r0)) reload(r0);
    request(cur, rec_ahead(plan, 0));
    request(fa, rec_ahead(plan, 1));
    // The compiler counts the memory operations issued after a load to know how many may
    // stay in flight when the load is consumed, and takes the minimum over all paths into
    // the loop.  On the path through the loop an op's stores follow the look-ahead loads;
    // these stores to the sink give the entry path the same shape, so that the wait at the
    // top of an op leaves the previous op's stores in flight.
#pragma unroll
    for (unsigned int j = 0; j < J; ++j)
    {
      st16<NT>(sink + lane, 0.0, 0.0);
      if (MODE != SCALE_NONE) reinterpret_cast<unsigned int *>(sink + 64)[lane] = 0u;
    }
    stage_rows(cur, pl, pr);
    // tip operands' pair tables: the gather of op i+1 is issued at the top of op i, from the
    // characters that arrived for it; what op i uses was gathered during op i-1
    double2 pt_use[J], pt_next[J];
    {
      const double2 * t0 = pair_table(r0);
#pragma unroll
      for (unsigned int j = 0; j < J; ++j)
      {
        const unsigned int pair = ((cur.codes_l[j] & 15u) << 4) | (cur.codes_r[j] & 15u);
        pt_next[j] = t0 ? t0[pair * W + (lane & (W - 1))] : zero16[0];
      }
    }
    for (unsigned int i = 0; i < nops; ++i)
    {
      const RecOp r2 = rec_op(plan, i + 2);
      const RecAhead a3 = rec_ahead(plan, i + 3);
      const int kind = rec_kind(r0), lslot = rec_lslot(r0), rslot = rec_rslot(r0), pslot = rec_pslot(r0);
      const int lsc_slot = rec_lsc(r0), rsc_slot = rec_rsc(r0);
      double2 * out = reinterpret_cast<double2 *>(bases.clv + (size_t)rec_parent(r0) * bases.site_stride * (RC * 4));
      const bool scaling = MODE != SCALE_NONE && rec_pscaler(r0) != PLLHIP_FUSED_NONE;
      unsigned int * pscaler = bases.scaler + (size_t)rec_pscaler(r0) * bases.site_stride * (MODE == SCALE_RATE ? RC : 1); // (used if scaling)
      const bool have_pairs = rec_pair(r0) != PLLHIP_FUSED_NONE; // this op takes its entries from its pair table
      const double2 * pair_next = pair_table(r1);                // the table of op i+1
      // (rare, wave-uniform: the sources of the reload are read on the spot)
      if (rec_dma(r1)) reload(r1);
      FusedFetch<PL, J> fb;
      request(fb, a2);
      unsigned int pairs[J];
#pragma unroll
      for (unsigned int j = 0; j < J; ++j) pairs[j] = ((fa.codes_l[j] & 15u) << 4) | (fa.codes_r[j] & 15u);
      // Everything requested one op ago has arrived once these characters are used -- and with
      // it what that iteration's reload() copied into this op's slots (issued ahead of those
      // requests; memory operations return in order).  No slot is read above this line.
      asm volatile("" ::"v"(pairs[J - 1]) : "memory");
#pragma unroll
      for (unsigned int j = 0; j < J; ++j)
      {
        pt_use[j] = pt_next[j];
        pt_next[j] = pair_next ? pair_next[pairs[j] * W + (lane & (W - 1))] : zero16[0];
      }

      double2 * out_tile = out + site0 * W;
      // (the counts of an op without a scale buffer go to a sink, so that every op issues the
      // same number of stores, see below)
      unsigned int * cnt_tile = scaling ? pscaler + ((MODE == SCALE_RATE) ? site0 * RC : site0)
                                        : reinterpret_cast<unsigned int *>(sink + 64);
      unsigned long long scaled[J];
#pragma unroll
      for (unsigned int j = 0; j < J; ++j)
      {
        const unsigned int g = j * 64u + lane; // granule within the tile
        // operands and inherited counts: LDS slots (slot 0 is read when there is none)
        const double2 lo = clv[((lslot >= 0 ? lslot : 0) * J + j) * 64 + lane];
        const double2 ro = clv[((rslot >= 0 ? rslot : 0) * J + j) * 64 + lane];
        double x0, x1, y0 = 1.0, y1 = 1.0;
        if (have_pairs)
        {
          // tip-tip with a pair table: the finished entries; tip-inner: the tip's factor
          x0 = pt_use[j].x;
          x1 = pt_use[j].y;
        }
        else if (kind == 0)
        {
          const double2 lp = make_double2(dpp_pair_swap(lo.x), dpp_pair_swap(lo.y));
          x0 = pl.dot(0, lo, lp);
          x1 = pl.dot(1, lo, lp);
        }
        else
        {
          // tip: the entries of rows 2h, 2h+1 that the character's state mask selects, in
          // the pairwise order of masksum4 (own pair + partner pair; commutative)
          const unsigned int code = cur.codes_l[j] & 15u;
          const unsigned int b0 = (code >> (2 * h)) & 1u, b1 = (code >> (2 * h + 1)) & 1u;
          const unsigned int b2 = (code >> (2 - 2 * h)) & 1u, b3 = (code >> (3 - 2 * h)) & 1u;
          x0 = ((b0 ? pl.m[0][0] : 0.0) + (b1 ? pl.m[0][1] : 0.0)) + ((b2 ? pl.m[0][2] : 0.0) + (b3 ? pl.m[0][3] : 0.0));
          x1 = ((b0 ? pl.m[1][0] : 0.0) + (b1 ? pl.m[1][1] : 0.0)) + ((b2 ? pl.m[1][2] : 0.0) + (b3 ? pl.m[1][3] : 0.0));
        }
        if (have_pairs && kind == 2)
          ;
        else if (kind != 2)
        {
          const double2 rp = make_double2(dpp_pair_swap(ro.x), dpp_pair_swap(ro.y));
          y0 = pr.dot(0, ro, rp);
          y1 = pr.dot(1, ro, rp);
        }
        else
        {
          const unsigned int code = cur.codes_r[j] & 15u;
          const unsigned int b0 = (code >> (2 * h)) & 1u, b1 = (code >> (2 * h + 1)) & 1u;
          const unsigned int b2 = (code >> (2 - 2 * h)) & 1u, b3 = (code >> (3 - 2 * h)) & 1u;
          y0 = ((b0 ? pr.m[0][0] : 0.0) + (b1 ? pr.m[0][1] : 0.0)) + ((b2 ? pr.m[0][2] : 0.0) + (b3 ? pr.m[0][3] : 0.0));
          y1 = ((b0 ? pr.m[1][0] : 0.0) + (b1 ? pr.m[1][1] : 0.0)) + ((b2 ? pr.m[1][2] : 0.0) + (b3 ? pr.m[1][3] : 0.0));
        }
        // (tip-tip with a pair table: y is exactly 1.0, the product is the table entry itself)
        double p0 = x0 * y0, p1 = x1 * y1;

        // scaling rule of core_partials_avx.c:486-527; tip-tip never scales and clears
        // its scaler (core_partials_avx.c:598-599)
        bool scale = false;
        if (scaling && kind != 2)
        {
          const bool small = (p0 < PLLHIP_SCALE_THRESHOLD) & (p1 < PLLHIP_SCALE_THRESHOLD);
          scale = (MODE == SCALE_RATE) ? group_all<2>(small) : group_all<W>(small);
          if (scale)
          {
            p0 *= PLLHIP_SCALE_FACTOR;
            p1 *= PLLHIP_SCALE_FACTOR;
          }
        }
        scaled[j] = __ballot(scale); // (wave-uniform: which groups of this sub-step were scaled)
        // The NUMBER of stores per op is fixed (a store under a branch forces a full drain
        // of the memory queue): the compiler can then wait for the look-ahead loads by count
        // and leave this op's stores in flight.
        st16<NT>(out_tile + g, p0, p1);
        if (pslot >= 0) clv[(pslot * J + j) * 64 + lane] = make_double2(p0, p1);
      }
      // The tile's counts, once per op: entry t (a site, or a (site, rate) with per-rate
      // scalers) is handled by lane t -- inherited counts from the operands' slots, plus
      // one if the sub-step that held the entry scaled its group -- and all of them leave in ONE
      // store (64 contiguous bytes per tile with per-site counts).  (Round 1 did this per
      // sub-step with all 64 lanes: two LDS reads, an LDS write and a 32-byte store each.  Measured:
      // the same speed either way.  A list without scale buffers runs 10-14 % faster than one
      // with them -- 500 k sites x 64 taxa 712 against 832 us -- but no single piece explains it:
      // counts stored to a sink instead 820, not stored at all 806, no scaling test 816.)
      if (MODE != SCALE_NONE)
      {
        constexpr unsigned int GW = (MODE == SCALE_RATE) ? 2u : W; // lanes that share a count
        constexpr unsigned int EPS = 64u / GW, E = J * EPS;        // entries per sub-step / per tile
        static_assert(CW == EPS, "one count word per entry of a sub-step");
        const unsigned int t = lane < E ? lane : 0u;
        unsigned long long mine = scaled[0];
#pragma unroll
        for (unsigned int j = 1; j < J; ++j) mine = (t / EPS == j) ? scaled[j] : mine;
        const unsigned int bit = (unsigned int)(mine >> ((t % EPS) * GW)) & 1u;
        // (tip operands and tip-tip ops inherit nothing; tip-tip never scales and clears its counts)
        unsigned int lc = cnt[(lsc_slot >= 0 ? lsc_slot : 0) * (J * CW) + t];
        unsigned int rc = cnt[(rsc_slot >= 0 ? rsc_slot : 0) * (J * CW) + t];
        if (lsc_slot < 0 || have_pairs || kind != 0) lc = 0u;
        if (rsc_slot < 0 || kind == 2) rc = 0u;
        const unsigned int count = lc + rc + ((scaling && kind != 2) ? bit : 0u);
        if (pslot >= 0 && lane < E) cnt[pslot * (J * CW) + t] = scaling ? count : 0u;
        unsigned int * dst = (scaling && lane < E) ? cnt_tile + lane : reinterpret_cast<unsigned int *>(sink + 64) + lane;
        *dst = count;
      }
      // the next op's matrix rows replace this op's in the same registers: the block was
      // requested an op ago, the LDS round trip overlaps the next op's scalar phase
      stage_rows(fa, pl, pr);
      cur.take(fa);
      fa.take(fb);
      r0 = r1;
      r1 = r2;
      a2 = a3;
    }
    if (++round < static_rounds)
    {
      tile += nwaves;
      continue;
    }
    unsigned int nt = 0;
    if (lane == 0) nt = atomicAdd(next_tile, 1u);
    tile = static_rounds * nwaves + (unsigned int)__builtin_amdgcn_readfirstlane((int)nt);
  }
}

// ---------------------------------------------------------------- host: order, slots, launch

namespace
{
struct Node
{
  std::vector<unsigned int> hard; // WAW / WAR predecessors (and scaler hazards): must run before
  int raw[2] = {-1, -1};          // producers of the two children within the list (-1: outside)
  int sraw[2] = {-1, -1};         // writers of the two child scale buffers within the list
  unsigned int need = 1;          // Sethi-Ullman number of the subtree rooted here
};
}

// slots per wave when `wgs` workgroups of four waves share a CU's LDS: 64 KB per workgroup
// for two (8 waves per CU), 52 KB for three (12 waves: better latency hiding, one slot less)
unsigned int pllhip_fused_slots(const pllhip_ctx * c, unsigned int wgs)
{
  // 64 KB per workgroup of four waves: 16 KB per wave minus the matrix block
  const unsigned int R = c->sh.rate_cats;
  const unsigned int sps = 64 / (2 * R);
  const size_t cw = c->sh.rate_scalers ? 32 : (sps < 4 ? 4 : sps);
  const size_t per_slot = (size_t)PLLHIP_FUSED_J * (64 * 16 + cw * 4);
  const size_t pmat = (size_t)R * 16 * sizeof(double); // one matrix at a time (stage_rows)
  const size_t budget = PLLHIP_FUSED_J == 1 ? 9472 : (wgs >= 3 ? 13312 : 16384); // J = 1: four workgroups per CU
  return (unsigned int)((budget - pmat) / per_slot);
}

// Slot assignment: every inner operand is read from a slot.  A value whose
// slot was taken away (or that an earlier call wrote) is copied back from HBM into a slot by
// the kernel's reload() at the top of the op BEFORE its reader; that slot must be free from
// then on (not read by that op, not its parent's).  Belady's rule decides who gives a slot up:
// the live value whose next reader is farthest away.
static int assign_slots_reload(const FusedGeom & geom, const pllhip_op_t * ops, const PartialsArgs * args,
                               const int * kinds, unsigned int count, unsigned int nslots,
                               const std::vector<unsigned int> & order, const std::vector<unsigned int> & pos_of,
                               const std::vector<Node> & node, std::vector<FusedOp> & plan,
                               unsigned int * reloads_out)
{
  // inner operands of the op at each position: producing list op (-1: written by an earlier
  // call), its HBM address, the HBM address of the counts the reader passes with it, and the
  // list op that wrote those counts
  struct Operand { int w; const double * hbm; const unsigned int * sc; int sw; };
  auto operands = [&](unsigned int i, Operand (&o)[2]) {
    o[0] = o[1] = Operand{-2, nullptr, nullptr, -1}; // -2: no such operand (a tip)
    if (kinds[i] == 0)
    {
      o[0] = Operand{node[i].raw[0], args[i].left, args[i].lscaler, node[i].sraw[0]};
      o[1] = Operand{node[i].raw[1], args[i].right, args[i].rscaler, node[i].sraw[1]};
    }
    else if (kinds[i] == 1)
    {
      const int inner = geom.is_tip(ops[i].child1_clv) ? 1 : 0;
      o[1] = Operand{node[i].raw[inner], args[i].right, args[i].rscaler, node[i].sraw[inner]};
    }
  };
  std::vector<std::vector<unsigned int>> uses(count); // positions at which each list value is read
  for (unsigned int pos = 0; pos < count; ++pos)
  {
    Operand o[2];
    operands(order[pos], o);
    if (o[0].w >= 0) uses[o[0].w].push_back(pos);
    if (o[1].w >= 0 && o[1].w != o[0].w) uses[o[1].w].push_back(pos);
  }
  std::vector<unsigned int> next_use(count, 0);
  std::vector<int> slot_of(count, -1);
  std::vector<int> free_slots;
  for (int s = (int)nslots - 1; s >= 0; --s) free_slots.push_back(s);
  std::vector<unsigned int> live;          // list values that hold a slot
  std::vector<int> oneshot, oneshot_next;  // slots of operands from earlier calls (of this op / the next): free after their one reader
  unsigned int reloads = 0;
  const unsigned int NEVER = ~0u;
  auto next_read = [&](unsigned int v) { return next_use[v] < uses[v].size() ? uses[v][next_use[v]] : NEVER; };
  // a slot that may be written from position `pos` on: a free one, else that of the live
  // value read farthest in the future -- but not before pos + 2 (its own reload is issued at
  // the top of the op before its reader and needs a slot free by then)
  auto take_slot = [&](unsigned int pos) -> int {
    if (!free_slots.empty())
    {
      const int s = free_slots.back();
      free_slots.pop_back();
      return s;
    }
    int victim = -1;
    unsigned int far = 0;
    for (unsigned int v : live)
    {
      const unsigned int u = next_read(v);
      if (u != NEVER && u >= pos + 2 && u >= far)
      {
        far = u;
        victim = (int)v;
      }
    }
    if (victim < 0) return -1;
    const int s = slot_of[victim];
    slot_of[victim] = -1;
    live.erase(std::find(live.begin(), live.end(), (unsigned int)victim));
    return s;
  };
  // operands of the op at position `pos` that are not in a slot: reloaded at the top of
  // position pos - 1 (`at`; the kernel's prologue for pos 0)
  auto place_reloads = [&](unsigned int pos, unsigned int at) -> int {
    const unsigned int i = order[pos];
    FusedOp & f = plan[pos];
    Operand o[2];
    operands(i, o);
    for (int side = 0; side < 2; ++side)
    {
      const Operand & x = o[side];
      if (x.w == -2) continue;
      if (side == 1 && x.w >= 0 && x.w == o[0].w) continue; // the same value twice: one slot
      if (x.w >= 0 && (pos_of[x.w] >= at || slot_of[x.w] >= 0)) continue; // still to come, or in a slot
      if (x.w >= 0 && pos_of[x.w] + 2 >= pos) return 1; // (cannot happen: evicted values are read later)
      const int s = take_slot(at);
      if (s < 0) return 1;
      ++reloads;
      const unsigned int * counts = nullptr;
      if (x.w >= 0)
      {
        // a value of this list: its counts are those its producer wrote, whoever reads it
        counts = args[x.w].pscaler;
        slot_of[x.w] = s;
        live.push_back((unsigned int)x.w);
      }
      else
      {
        // written by an earlier call: the counts the reader passes, which no op of this list
        // may have rewritten shortly before
        if (x.sc && x.sw >= 0 && pos_of[x.sw] + 2 >= pos) return 1;
        counts = x.sc;
        oneshot_next.push_back(s);
      }
      if (side == 0) { f.left_hbm = x.hbm; f.lsc_hbm = counts; f.lslot = s; f.dma_flags |= 1; }
      else { f.right_hbm = x.hbm; f.rsc_hbm = counts; f.rslot = s; f.dma_flags |= 2; }
    }
    return 0;
  };

  plan.resize(count);
  for (unsigned int pos = 0; pos < count; ++pos)
  {
    const unsigned int i = order[pos];
    const PartialsArgs & a = args[i];
    FusedOp & f = plan[pos];
    memset(&f, 0, sizeof(f));
    f.parent = a.parent;
    f.ltip = a.ltip;
    f.rtip = a.rtip;
    f.lmat = a.lmat;
    f.rmat = a.rmat;
    f.pscaler = a.pscaler;
    f.kind = kinds[i];
    f.list_pos = (int)i;
    f.lslot = f.rslot = f.pslot = f.lsc_slot = f.rsc_slot = -1;
  }
  if (place_reloads(0, 0)) return 1;
  oneshot.swap(oneshot_next);
  for (unsigned int pos = 0; pos < count; ++pos)
  {
    const unsigned int i = order[pos];
    FusedOp & f = plan[pos];
    // top of the op: the next op's missing operands are requested into slots free NOW
    if (pos + 1 < count && place_reloads(pos + 1, pos)) return 1;
    Operand o[2];
    operands(i, o);
    for (int side = 0; side < 2; ++side)
    {
      const Operand & x = o[side];
      if (x.w == -2) continue;
      int & slot = side == 0 ? f.lslot : f.rslot;
      int & sc_slot = side == 0 ? f.lsc_slot : f.rsc_slot;
      if (x.w >= 0)
      {
        if (slot < 0) slot = slot_of[x.w];
        if (slot < 0) return 1;
        // counts: only those written together with the value live in its slot
        if (x.sc)
        {
          if (x.sw != x.w || x.sc != args[x.w].pscaler) return 1;
          sc_slot = slot;
        }
      }
      else
      {
        if (slot < 0) return 1; // (placed by place_reloads)
        if (x.sc) sc_slot = slot;
      }
    }
    // operands read for the last time give their slots back
    for (int side = 0; side < 2; ++side)
    {
      const int w = o[side].w;
      if (w < 0 || (side == 1 && w == o[0].w)) continue;
      if (next_use[w] < uses[w].size() && uses[w][next_use[w]] == pos) next_use[w]++;
      if (next_use[w] >= uses[w].size() && slot_of[w] >= 0)
      {
        free_slots.push_back(slot_of[w]);
        slot_of[w] = -1;
        live.erase(std::find(live.begin(), live.end(), (unsigned int)w));
      }
    }
    for (int s : oneshot) free_slots.push_back(s);
    oneshot.clear();
    oneshot.swap(oneshot_next);
    // the parent: a slot if it has readers -- unless its first reader is far enough away for a
    // reload (three ops: its stores must have left) and farther than every live value's next
    if (!uses[i].empty())
    {
      const unsigned int first = uses[i][0];
      bool wants = true;
      if (free_slots.empty() && first >= pos + 3)
      {
        unsigned int far = 0;
        for (unsigned int v : live)
        {
          const unsigned int u = next_read(v);
          if (u != NEVER && u >= pos + 2 && u > far) far = u;
        }
        if (first >= far) wants = false;
      }
      if (wants)
      {
        const int s = take_slot(pos);
        if (s < 0)
        {
          if (first < pos + 3) return 1;
        }
        else
        {
          f.pslot = s;
          slot_of[i] = s;
          live.push_back(i);
        }
      }
    }
  }
  *reloads_out = reloads;
  return 0;
}

int pllhip_fused_plan(const FusedGeom & geom, const pllhip_op_t * ops, const PartialsArgs * args,
                      const int * kinds, unsigned int count, unsigned int nslots,
                      std::vector<FusedOp> & plan, unsigned int * reloads_out)
{
  std::vector<Node> node(count);
  const size_t nclv = geom.nclv, nsc = geom.nsc;
  // last writer and readers-since of every CLV / scale buffer, in list order
  std::vector<int> clv_w(nclv, -1), sc_w(nsc, -1);
  std::vector<std::vector<unsigned int>> clv_r(nclv), sc_r(nsc);
  for (unsigned int i = 0; i < count; ++i)
  {
    const pllhip_op_t & op = ops[i];
    Node & nd = node[i];
    auto hard = [&](int p) { if (p >= 0 && (unsigned int)p != i) nd.hard.push_back((unsigned int)p); };
    nd.raw[0] = clv_w[op.child1_clv];
    nd.raw[1] = clv_w[op.child2_clv];
    nd.sraw[0] = op.child1_scaler >= 0 ? sc_w[op.child1_scaler] : -1;
    nd.sraw[1] = op.child2_scaler >= 0 ? sc_w[op.child2_scaler] : -1;
    hard(clv_w[op.parent_clv]);
    for (unsigned int r : clv_r[op.parent_clv]) hard((int)r);
    hard(nd.sraw[0]);
    hard(nd.sraw[1]);
    if (op.parent_scaler >= 0)
    {
      hard(sc_w[op.parent_scaler]);
      for (unsigned int r : sc_r[op.parent_scaler]) hard((int)r);
    }
    clv_w[op.parent_clv] = (int)i;
    clv_r[op.parent_clv].clear();
    clv_r[op.child1_clv].push_back(i);
    clv_r[op.child2_clv].push_back(i);
    if (op.parent_scaler >= 0)
    {
      sc_w[op.parent_scaler] = (int)i;
      sc_r[op.parent_scaler].clear();
    }
    if (op.child1_scaler >= 0) sc_r[op.child1_scaler].push_back(i);
    if (op.child2_scaler >= 0) sc_r[op.child2_scaler].push_back(i);
    const unsigned int a = nd.raw[0] >= 0 ? node[nd.raw[0]].need : 0;
    const unsigned int b = nd.raw[1] >= 0 ? node[nd.raw[1]].need : 0;
    nd.need = std::max(1u, a == b ? a + (a ? 1u : 0u) : std::max(a, b));
  }

  // depth-first order from the end of the list: hazards first, then the heavier child
  std::vector<unsigned int> order;
  order.reserve(count);
  {
    std::vector<unsigned char> state(count, 0); // 0 new, 1 open, 2 emitted
    std::vector<std::pair<unsigned int, unsigned int>> stack; // (op, next predecessor to look at)
    std::vector<std::vector<unsigned int>> preds(count);
    for (unsigned int i = 0; i < count; ++i)
    {
      preds[i] = node[i].hard;
      int r0 = node[i].raw[0], r1 = node[i].raw[1];
      if (r0 >= 0 && r1 >= 0 && node[r1].need > node[r0].need) std::swap(r0, r1);
      if (r0 >= 0) preds[i].push_back((unsigned int)r0);
      if (r1 >= 0 && r1 != r0) preds[i].push_back((unsigned int)r1);
    }
    for (unsigned int root = count; root-- > 0;)
    {
      if (state[root]) continue;
      stack.push_back({root, 0});
      state[root] = 1;
      while (!stack.empty())
      {
        auto & top = stack.back();
        if (top.second < preds[top.first].size())
        {
          const unsigned int p = preds[top.first][top.second++];
          if (!state[p])
          {
            state[p] = 1;
            stack.push_back({p, 0});
          }
        }
        else
        {
          state[top.first] = 2;
          order.push_back(top.first);
          stack.pop_back();
        }
      }
    }
  }
  std::vector<unsigned int> pos_of(count);
  for (unsigned int pos = 0; pos < count; ++pos) pos_of[order[pos]] = pos;

  const int rc = assign_slots_reload(geom, ops, args, kinds, count, nslots, order, pos_of, node, plan, reloads_out);
  if (rc) return rc;
  if (getenv("PLLHIP_FUSED_DEBUG"))
  {
    fprintf(stderr, "pllhip fused plan: %u ops, %u slots, %u operands reloaded from HBM\n", count, nslots, *reloads_out);
    if (atoi(getenv("PLLHIP_FUSED_DEBUG")) > 1)
      for (unsigned int pos = 0; pos < count; ++pos)
      {
        const FusedOp & f = plan[pos];
        fprintf(stderr, "  %3u: op %3d kind %d  l %2d r %2d p %2d  lsc %2d rsc %2d  dma %d  hbm %p %p counts %p %p\n", pos,
                f.list_pos, f.kind, f.lslot, f.rslot, f.pslot, f.lsc_slot, f.rsc_slot, f.dma_flags,
                (const void *)f.left_hbm, (const void *)f.right_hbm, (const void *)f.lsc_hbm, (const void *)f.rsc_hbm);
      }
  }
  return 0;
}

// The planner without a device (tests/test_host.py, tools): which order and how many
// operands without a slot a list gets with `nslots` slots per wave.
extern "C" int pllhip_fused_plan_dry(unsigned int tips, unsigned int clv_buffers, unsigned int scale_buffers,
                                     int pattern_tip, const pllhip_op_t * ops, unsigned int count,
                                     unsigned int nslots, unsigned int * order_out,
                                     unsigned int * reloads_out, int * slots_out)
{
  const FusedGeom geom = {(size_t)tips + clv_buffers, scale_buffers, tips, pattern_tip != 0};
  std::vector<PartialsArgs> args(count);
  std::vector<int> kinds(count);
  for (unsigned int i = 0; i < count; ++i)
  {
    const pllhip_op_t & op = ops[i];
    if (op.parent_clv >= geom.nclv || op.child1_clv >= geom.nclv || op.child2_clv >= geom.nclv ||
        op.parent_scaler >= (int)scale_buffers || op.child1_scaler >= (int)scale_buffers ||
        op.child2_scaler >= (int)scale_buffers)
    {
      pllhip_set_error("pllhip_fused_plan_dry: index out of range in op %u", i);
      return -1;
    }
    const bool t1 = geom.is_tip(op.child1_clv), t2 = geom.is_tip(op.child2_clv);
    memset(&args[i], 0, sizeof(PartialsArgs));
    kinds[i] = (t1 && t2) ? 2 : (t1 || t2) ? 1 : 0;
    // (distinct fake addresses per scale buffer: the reload plan compares them)
    auto sc = [&](int idx) { return idx >= 0 ? reinterpret_cast<unsigned int *>((uintptr_t)4096 * (idx + 1)) : (unsigned int *)nullptr; };
    args[i].pscaler = sc(op.parent_scaler);
    auto clv = [&](unsigned int idx) { return reinterpret_cast<const double *>((uintptr_t)4096 * (idx + 1)); };
    if (kinds[i] == 0)
    {
      args[i].left = clv(op.child1_clv);
      args[i].right = clv(op.child2_clv);
      args[i].lscaler = sc(op.child1_scaler);
      args[i].rscaler = sc(op.child2_scaler);
    }
    else if (kinds[i] == 1)
    {
      args[i].right = clv(t1 ? op.child2_clv : op.child1_clv);
      args[i].rscaler = sc(t1 ? op.child2_scaler : op.child1_scaler);
    }
  }
  std::vector<FusedOp> plan;
  unsigned int reloads = 0;
  const int rc = pllhip_fused_plan(geom, ops, args.data(), kinds.data(), count, nslots, plan, &reloads);
  if (rc) return rc;
  for (unsigned int pos = 0; pos < count; ++pos)
  {
    if (order_out) order_out[pos] = (unsigned int)plan[pos].list_pos;
    if (slots_out)
    {
      const FusedOp & f = plan[pos];
      const int v[6] = {f.lslot, f.rslot, f.pslot, f.lsc_slot, f.rsc_slot, f.dma_flags};
      for (int t = 0; t < 6; ++t) slots_out[pos * 6 + t] = v[t];
    }
  }
  if (reloads_out) *reloads_out = reloads;
  return 0;
}

template <int RC>
static int launch_fused_rc(pllhip_ctx * c, const FusedRec * d_plan, const FusedBases & bases, unsigned int count,
                           unsigned int nslots, int mode)
{
  constexpr int J = PLLHIP_FUSED_J;
  const unsigned int sites = c->sh.sites;
  const size_t tile_sites = (size_t)J * (64 / (2 * RC));
  const size_t tiles = (sites + tile_sites - 1) / tile_sites;
  const size_t cw = c->sh.rate_scalers ? 32 : ((64 / (2 * RC)) < 4 ? 4 : (64 / (2 * RC)));
  const size_t lds = 4 * ((size_t)nslots * J * (64 * 16 + cw * 4) + (size_t)RC * 16 * sizeof(double));
  // three workgroups (12 waves) per CU when the slots leave room for them, else two; each wave
  // walks its share of the tiles
  size_t grid = (tiles + 3) / 4;
  const size_t cap = (size_t)c->num_cus * (J == 1 ? 4 : (nslots <= pllhip_fused_slots(c, 3) ? 3 : 2));
  if (grid > cap) grid = cap;
  const bool nt = pllhip_use_nt(c);
  unsigned int * tile_counter = getenv("PLLHIP_FUSED_STATIC_TILES") ? nullptr : c->d_tile_counter;
  // Address translations: the translation caches reach about 8 GB (4096 pages of 2 MB).  A
  // partition whose CLVs exceed that is swept from end to end by every launch, so a launch
  // that starts where the previous one started finds none of its pages cached (measured:
  // every shape runs at 0.59-0.64 of the HBM peak up to 8 GB of CLVs and at 0.46-0.50 beyond,
  // whatever the tree -- profiles/r2_footprint.txt).  Every other launch therefore walks the
  // tiles backwards: it starts in the pages the previous launch touched last.
  const unsigned int backwards = (c->fused_pingpong && c->clv_arena_bytes > ((size_t)6 << 30)) ? (c->fused_launches++ & 1u) : 0u;
  const unsigned int dynamic_rounds = getenv("PLLHIP_FUSED_DYNAMIC_ROUNDS") ? (unsigned int)atoi(getenv("PLLHIP_FUSED_DYNAMIC_ROUNDS"))
                                      : (count >= 32 ? 7u : 2u);
#define LAUNCH_FUSED(MODEV, NTV)                                                                                  \
  k_dna_fused<RC, J, MODEV, NTV, (J == 1 ? 4 : 3)><<<(unsigned int)grid, 256, lds, c->stream>>>(                 \
      d_plan, bases, count, sites, nslots, c->d_zero, (double2 *)c->d_sink, tile_counter, backwards, dynamic_rounds)
#define LAUNCH_FUSED_MODE(NTV)                         \
  do {                                                  \
    if (mode == SCALE_NONE) LAUNCH_FUSED(0, NTV);       \
    else if (mode == SCALE_SITE) LAUNCH_FUSED(1, NTV);  \
    else LAUNCH_FUSED(2, NTV);                          \
  } while (0)
  if (nt) LAUNCH_FUSED_MODE(true);
  else LAUNCH_FUSED_MODE(false);
#undef LAUNCH_FUSED_MODE
#undef LAUNCH_FUSED
  HIP_TRY(hipGetLastError());
  return 0;
}

// what the indices of the plan records are relative to
static FusedBases fused_bases(const pllhip_ctx * c)
{
  FusedBases b;
  b.clv = c->clv_arena;
  b.scaler = c->scaler_arena;
  b.tips = c->tipchars;
  b.pmat = c->pmatrix;
  b.pairtab = c->d_pairtab;
  b.site_stride = c->sh.sites + PLLHIP_TAIL_SITES;
  b.tip_stride = (unsigned int)c->tip_stride;
  return b;
}

int pllhip_launch_fused(pllhip_ctx * c, const std::vector<FusedOp> & plan, unsigned int nslots)
{
  const unsigned int count = (unsigned int)plan.size();
  // pair tables of the tip-tip and tip-inner ops (k_dna_pair_tables), carved from one device buffer
  const int level = c->fused_pairs; // env PLLHIP_FUSED_PAIRS: 0 off, 1 tip-tip only, 2 (default) tip-inner too
  const int min_kind = level == 0 ? 3 : (level == 1 ? 2 : 1);
  const size_t per = (size_t)256 * c->sh.rate_cats * 4;
  size_t ntab = 0;
  for (const FusedOp & f : plan) ntab += (f.kind >= min_kind);
  if (ntab > PLLHIP_FUSED_MAX_INDEX) return 1;
  if (ntab && c->pairtab_elems < ntab * per)
  {
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->d_pairtab) HIP_TRY(hipFree(c->d_pairtab));
    c->d_pairtab = nullptr;
    HIP_TRY(hipMalloc((void **)&c->d_pairtab, ntab * per * sizeof(double)));
    c->pairtab_elems = ntab * per;
    ++c->layout_epoch;
  }
  // the records: indices relative to the arenas (a list whose indices do not fit 16 bits
  // runs per level), then three copies of the last op (what the kernel's look-ahead
  // requests beyond the end), then the sources of the reloads
  const unsigned int first_clv = c->sh.pattern_tip ? c->sh.tips : 0;
  if (c->clv.size() - first_clv > PLLHIP_FUSED_MAX_INDEX || c->sh.scale_buffers > PLLHIP_FUSED_MAX_INDEX ||
      c->sh.tips > PLLHIP_FUSED_MAX_INDEX || c->sh.prob_matrices > PLLHIP_FUSED_MAX_INDEX)
    return 1;
  std::vector<FusedRec> recs(count + 3);
  std::vector<FusedSrc> srcs;
  int mode = SCALE_NONE;
  size_t tab = 0;
  for (unsigned int pos = 0; pos < count; ++pos)
  {
    const FusedOp & f = plan[pos];
    FusedRec & r = recs[pos];
    memset(&r, 0, sizeof(r));
    r.parent = (unsigned short)((f.parent - c->clv_arena) / c->clv_stride);
    r.pscaler = f.pscaler ? (unsigned short)((f.pscaler - c->scaler_arena) / c->scaler_stride) : PLLHIP_FUSED_NONE;
    r.ltip = f.ltip ? (unsigned short)((f.ltip - c->tipchars) / c->tip_stride) : PLLHIP_FUSED_NONE;
    r.rtip = f.rtip ? (unsigned short)((f.rtip - c->tipchars) / c->tip_stride) : PLLHIP_FUSED_NONE;
    r.lmat = (unsigned short)((f.lmat - c->pmatrix) / c->pmat_elems);
    r.rmat = (unsigned short)((f.rmat - c->pmatrix) / c->pmat_elems);
    r.pair = f.kind >= min_kind ? (unsigned short)(tab++) : PLLHIP_FUSED_NONE;
    r.src = PLLHIP_FUSED_NONE;
    if (f.dma_flags)
    {
      if (srcs.size() >= PLLHIP_FUSED_MAX_INDEX) return 1;
      r.src = (unsigned short)srcs.size();
      srcs.push_back(FusedSrc{f.left_hbm, f.right_hbm, f.lsc_hbm, f.rsc_hbm});
    }
    r.lslot = (signed char)f.lslot;
    r.rslot = (signed char)f.rslot;
    r.pslot = (signed char)f.pslot;
    r.kind = (signed char)f.kind;
    r.lsc_slot = (signed char)f.lsc_slot;
    r.rsc_slot = (signed char)f.rsc_slot;
    r.dma_flags = (unsigned char)f.dma_flags;
    r.list_pos = (unsigned int)f.list_pos;
    // every op with a parent scaler scales the partition's way
    if (f.pscaler) mode = c->sh.rate_scalers ? SCALE_RATE : SCALE_SITE;
  }
  for (unsigned int t = 0; t < 3; ++t)
  {
    recs[count + t] = recs[count - 1];
    recs[count + t].dma_flags = 0;
    recs[count + t].src = PLLHIP_FUSED_NONE;
  }
  const size_t rec_bytes = recs.size() * sizeof(FusedRec);
  const size_t bytes = rec_bytes + (srcs.size() + 1) * sizeof(FusedSrc);
  if (c->plan_cap < bytes)
  {
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int b = 0; b < 2; ++b)
    {
      if (c->h_plan[b]) HIP_TRY(hipHostFree(c->h_plan[b]));
      c->h_plan[b] = nullptr;
      HIP_TRY(hipHostMalloc(&c->h_plan[b], bytes * 2, hipHostMallocDefault));
      if (!c->plan_done[b]) HIP_TRY(hipEventCreateWithFlags(&c->plan_done[b], hipEventDisableTiming));
      c->plan_pending[b] = false;
    }
    if (c->d_plan) HIP_TRY(hipFree(c->d_plan));
    c->d_plan = nullptr;
    HIP_TRY(hipMalloc(&c->d_plan, bytes * 2));
    c->plan_cap = bytes * 2;
    ++c->layout_epoch;
  }
  // two pinned staging buffers in turn: the copy of the call before last has long finished
  const int b = c->plan_next;
  c->plan_next ^= 1;
  if (c->plan_pending[b]) HIP_TRY(hipEventSynchronize(c->plan_done[b]));
  memcpy(c->h_plan[b], recs.data(), rec_bytes);
  if (!srcs.empty()) memcpy(static_cast<char *>(c->h_plan[b]) + rec_bytes, srcs.data(), srcs.size() * sizeof(FusedSrc));
  HIP_TRY(hipMemcpyAsync(c->d_plan, c->h_plan[b], bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipEventRecord(c->plan_done[b], c->stream));
  c->plan_pending[b] = true;
  if (!c->d_sink) HIP_TRY(hipMalloc(&c->d_sink, (size_t)c->num_cus * 16 * 80 * sizeof(double2))); // 1280 B per wave
  if (!c->d_tile_counter) HIP_TRY(hipMalloc((void **)&c->d_tile_counter, sizeof(unsigned int)));
  // what a repeated call with the same op list needs (pllhip_relaunch_fused)
  c->fused_last_entries = (unsigned int)recs.size();
  c->fused_last_count = count;
  c->fused_last_nslots = nslots;
  c->fused_last_mode = mode;
  c->fused_last_epoch = c->layout_epoch;
  return pllhip_relaunch_fused(c);
}

// The device copy of the plan is still that of the previous call (same op list: the plan
// holds buffer indices, not values -- P-matrices, tip characters and CLVs are read when the
// kernels run): tip tables and the list kernel again, no planning, no upload.
int pllhip_relaunch_fused(pllhip_ctx * c)
{
  const unsigned int entries = c->fused_last_entries, count = c->fused_last_count, nslots = c->fused_last_nslots;
  const int mode = c->fused_last_mode;
  HIP_TRY(hipMemsetAsync(c->d_tile_counter, 0, sizeof(unsigned int), c->stream));
  const FusedRec * d_plan = (const FusedRec *)c->d_plan;
  const FusedBases bases = fused_bases(c);
  switch (c->sh.rate_cats)
  {
    case 1: k_dna_pair_tables<1><<<entries, 256, 0, c->stream>>>(d_plan, entries, bases); break;
    case 2: k_dna_pair_tables<2><<<entries, 256, 0, c->stream>>>(d_plan, entries, bases); break;
    default: k_dna_pair_tables<4><<<entries, 256, 0, c->stream>>>(d_plan, entries, bases); break;
  }
  HIP_TRY(hipGetLastError());
  switch (c->sh.rate_cats)
  {
    case 1: return launch_fused_rc<1>(c, d_plan, bases, count, nslots, mode);
    case 2: return launch_fused_rc<2>(c, d_plan, bases, count, nslots, mode);
    default: return launch_fused_rc<4>(c, d_plan, bases, count, nslots, mode);
  }
}
