// partials_fused.hip -- a whole op list of 4-state CLV updates in ONE kernel, site-blocked.
//
// Replaces the per-level launches of partials.hip for pll_update_partials
// (partials.c:214-278 -> core_partials.c) when the list has more than one op and the partition at
// least one tile per SIMD.
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
//             Everything an op needs from memory (its two P-matrices) is therefore requested
//             during the op two before it, ahead of that op's stores, with unconditional loads,
//             and its pair-table entries are gathered during the op before it.
//   characters  the tile's characters of up to 64 tip rows are fetched with ONE load at the top of
//             the tile and kept in four registers; an op takes its own with v_readlane (round 3:
//             per-op requests of two rows each were what held partitions beyond 33 GB at 0.5-0.57)
//   tiles     a wave's first rounds by fixed stride, the last third from eight counters (the XCDs
//             do not write at the same rate), the ticket requested two ops before it is needed
//   plan      one 64-byte record per op, read through the scalar data cache one op ahead: absolute
//             addresses and LDS offsets, decoded by the host (FusedRec in partials_fused.hpp)
//   limit     a wave's own serial path, not HBM: twelve waves per CU is all the slots allow, so
//             the order within an op overlaps the wave's latencies with its own work (see step())
//
// The arithmetic per site is that of k_dna_partials, statement for statement (dot4 /
// masksum4 order, scaling rule of core_partials_avx.c:486-527), so results are
// bit-identical to the per-level path; tests/test_gpu_parity.py and the random-op-sequence
// tests run through this kernel.
//
// Roofline: HBM writes.  132 B per site-update (128 B CLV + 4 B scaler count) + 1 B per
// tip character read; operands that are reloaded add 128 B each.  Round 3: 0.69-0.75 of the
// 8 TB/s peak on every shape from 8 to 133 GB, rate categories 1, 2, 4, 8, lists from two ops on
// (DESIGN.md 2.0).
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <memory_resource>
#include <vector>

#include "ctx.hpp"
#include "numerics.hpp"
#include "partials_fused.hpp"

#define PLL_LDS __attribute__((address_space(3)))
// The plan holds addresses as integers; a pointer made from one must say that it points to global
// memory, or its accesses become FLAT ones (which count as LDS operations too and drain both queues).
#define PLL_GLOBAL __attribute__((address_space(1)))
typedef pll_v2d PLL_GLOBAL * global_v2d;
template <bool NT>
__device__ __forceinline__ void st16g(unsigned long long base, unsigned int byte_offset, double a, double b)
{
  const global_v2d p = (global_v2d)(base + byte_offset);
  const pll_v2d v = {a, b};
  if (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// ---- a plan record: sixteen words, one scalar load (all of this is wave-uniform) ----
typedef const unsigned int __attribute__((address_space(4))) * const_words;
typedef const unsigned long long __attribute__((address_space(4))) * const_quads;
struct Rec
{
  unsigned int w[16];
};
// (constant address space: a scalar load; adjacent words, the compiler merges them into one)
__device__ __forceinline__ Rec rec_load(const FusedRec * plan, unsigned int i)
{
  const const_words p = (const_words)(unsigned long long)(plan + i);
  Rec r;
#pragma unroll
  for (int t = 0; t < 16; ++t) r.w[t] = p[t];
  return r;
}
__device__ __forceinline__ unsigned long long rec_quad(const Rec & r, int t) { return (unsigned long long)r.w[t] | ((unsigned long long)r.w[t + 1] << 32); }
__device__ __forceinline__ unsigned int rec_chars(const Rec & r) { return r.w[0]; }
__device__ __forceinline__ unsigned int rec_req_lmat(const Rec & r) { return r.w[4]; }
__device__ __forceinline__ unsigned int rec_req_rmat(const Rec & r) { return r.w[5]; }
__device__ __forceinline__ unsigned int rec_gather(const Rec & r) { return r.w[6]; }
__device__ __forceinline__ unsigned int rec_flags(const Rec & r) { return r.w[7]; }
__device__ __forceinline__ unsigned long long rec_parent(const Rec & r) { return rec_quad(r, 8); }
__device__ __forceinline__ unsigned long long rec_pscaler(const Rec & r) { return rec_quad(r, 10); }
__device__ __forceinline__ unsigned int rec_lslot(const Rec & r) { return r.w[12] & 0xffffu; }
__device__ __forceinline__ unsigned int rec_rslot(const Rec & r) { return r.w[12] >> 16; }
__device__ __forceinline__ unsigned int rec_pslot(const Rec & r) { return r.w[13] & 0xffffu; }
__device__ __forceinline__ unsigned int rec_src(const Rec & r) { return r.w[13] >> 16; }
__device__ __forceinline__ unsigned int rec_lcnt(const Rec & r) { return r.w[14] & 0xffffu; }
__device__ __forceinline__ unsigned int rec_rcnt(const Rec & r) { return r.w[14] >> 16; }
__device__ __forceinline__ unsigned int rec_pcnt(const Rec & r) { return r.w[15] & 0xffffu; }

// Tip operands: the parent entry of a tip-tip op depends on its two tip characters only, 16 x 16
// pairs.  One table per op with a tip, [pair][rate][state] = masksum4(P_l row, code 1) *
// masksum4(P_r row, code 2) -- the very product the kernel would form per site (30 VALU
// instructions per tip operand and sub-step) -- built by one small launch ahead of the
// list and read back by the list kernel with one 16-byte gather per lane and sub-step
// (32 KB per op at 4 rate categories: L2-resident).  Table 0 is all zeros: what ops without a
// tip gather from (every op issues the same loads).
// (round 4: the launch also reset the tile counters of the list kernel behind it -- counter g is word 32 g, one
// 128-byte line each -- which used to be a fill kernel of its own per call, 4.3 us; round 5: the list kernel resets
// the set of counters of the launch BEHIND it, two sets in turn, and this launch no longer does)
template <int RC>
__global__ __launch_bounds__(256) void k_dna_pair_tables(const FusedPairJob * __restrict__ jobs, unsigned int njobs,
                                                         unsigned int * __restrict__ tile_counters)
{
  // (round 6: four workgroups per job, a quarter of the 256 pairs each -- the launch is a chain of latencies, and
  // thirty-odd workgroups of sixteen rounds each left the device empty for 10 us ahead of every list; a thread reads
  // ITS row of each matrix -- 32 bytes -- straight into registers: every pair it makes has the same (rate, state))
  const unsigned int i = blockIdx.x >> 2, quarter = blockIdx.x & 3u;
  if (blockIdx.x == 0 && tile_counters) tile_counters[threadIdx.x * 32u] = 0u;
  if (i >= njobs) return;
  // Round 5: the table written in address order.  (A thread per character pair that
  // read its 2 x 16 rows of 4 straight from memory and wrote 128 contiguous bytes of its own took 10.6 us per launch
  // -- sixteen dependent trips to L1 per thread --, a fifth of pll_update_partials at 20,000 sites.)  Same products,
  // same order: masksum4 of a row, times masksum4 of a row.
  const PLL_GLOBAL double * lm = (const PLL_GLOBAL double *)jobs[i].lmat, * rm = (const PLL_GLOBAL double *)jobs[i].rmat;
  double * tab = jobs[i].tab;
  // (tip-inner ops: the tip's factor alone, in the entries [code 1][0] the kernel's index
  // (code 1 << 4 | character of an absent tip = 0) reaches; x * 1.0 is x)
  const bool tt = jobs[i].tip_tip != 0;
  constexpr unsigned int ROW = RC * 4u;                 // entries of a pair: (rate, state)
  constexpr unsigned int PER = 256u / ROW;              // pairs a workgroup makes per round
  constexpr unsigned int ROUNDS = 64u / PER;            // rounds of its quarter (64 pairs)
  const unsigned int ki = threadIdx.x % ROW, sub = threadIdx.x / ROW;
  double l[4], r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
  {
    l[k] = lm[ki * 4u + k];
    r[k] = tt ? rm[ki * 4u + k] : 0.0;
  }
#pragma unroll
  for (unsigned int round = 0; round < ROUNDS; ++round)
  {
    const unsigned int pair = quarter * 64u + round * PER + sub, c1 = pair >> 4, c2 = pair & 15u;
    tab[pair * ROW + ki] = masksum4(l, c1) * (tt ? masksum4(r, c2) : 1.0);
  }
}

// what a lane requests for an op ahead of its use
template <int J>
struct FusedFetch
{
  double2 pm;  // its 16 bytes of [P_l | P_r] (a coalesced block per wave)
  double2 pm2; // 8 rate categories: a matrix is a whole 1 KB block -- pm of the left, pm2 of the right one
};

typedef unsigned int pll_v4u __attribute__((ext_vector_type(4)));

// all lanes of a group of W (2, 4 or 8: a site's lanes, or a rate's) hold x != 0?  DPP, no ballot
template <int W>
__device__ __forceinline__ unsigned int group_and(unsigned int x)
{
  x &= (unsigned int)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);                // lane ^ 1
  if (W >= 4) x &= (unsigned int)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, true);    // lane ^ 2
  if (W >= 8) x &= (unsigned int)__builtin_amdgcn_mov_dpp((int)x, 0x141, 0xF, 0xF, true);   // lane <-> 7 - lane
  if (W >= 16) x &= (unsigned int)__builtin_amdgcn_mov_dpp((int)x, 0x128, 0xF, 0xF, true);  // row_ror:8 = lane ^ 8
  return x;
}

// WPS: waves per SIMD the register budget is sized for (3 = 168 VGPRs: twelve waves per CU)
// NTP: cache policy -- 0 plain stores, 1 the tiles non-temporal, 2 the counts as well
template <int RC, int J, int MODE, int NTP, int WPS>
__global__ __launch_bounds__(256, WPS) void k_dna_fused(const FusedRec * __restrict__ plan0, FusedBases bases,
                                                        unsigned int nops0, unsigned int sites, unsigned int nslots,
                                                        double2 * sink, unsigned int * next_tile, unsigned int dynamic_rounds,
                                                        unsigned int site_base, unsigned int tile_groups,
                                                        unsigned int * reset_tiles)
{
  // (round 5) Two sets of tile counters in turn: this launch hands out tiles from `next_tile` and zeroes the OTHER set
  // for the launch behind it (stream order separates the two) -- until then a list without tip operands, which has no
  // table launch to do it, was preceded by a 32 KB hipMemsetAsync: two fill kernels in front of every short
  // all-inner list, the path to the root after a branch-length change (tools/step_floor.c).
  if (reset_tiles && blockIdx.x == 0) reset_tiles[threadIdx.x * 32u] = 0u;
  static_assert(RC == 1 || RC == 2 || RC == 4 || RC == 8, "lane groups of 2, 4, 8 or 16");
  constexpr bool NT = NTP != 0;
  constexpr unsigned int W = 2 * RC, SPS = 64 / W, TS = J * SPS;
  constexpr unsigned int MG = RC * 8;                   // 16-byte granules of one P-matrix
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
  // what never changes for a lane: its byte offsets into a tile, a tip row, a pair-table entry, a matrix block
  const unsigned int lane16 = lane * 16u;
  const unsigned int tip_lane = lane / W;
  const unsigned int gat_lane = (lane & (W - 1)) * 16u;
  const unsigned int pm_lane = (lane & (MG - 1)) * 16u;
  const bool pm_left = lane < MG;

  // Every CLV, scale buffer and tip row has PLLHIP_TAIL_SITES sites of slack of its own
  // behind its last site (ctx.hip), so the last tile is loaded and STORED whole: no lane
  // predicates, and every address is a wave-uniform tile base plus a lane offset that
  // never changes.
  const size_t tiles = ((size_t)sites + TS - 1) / TS;
  // (round 5) work items: (tile, segment) pairs, segment-major -- every tile of the longest segment first
  const unsigned int nsegs = bases.nsegs;
  const size_t items = tiles * nsegs;
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
  sink += wave * 96;
  unsigned int * sink_cnt = reinterpret_cast<unsigned int *>(sink + 64);
  // Tiles of the last rounds come from `tile_groups` counters (eight): groups of eight consecutive workgroups --
  // one on each XCD -- take the counters in turn, counter g hands out tiles g, g + 8, ... of that region.  (One
  // counter for everybody serialised its atomics at ~12 ns: 37 us per round of 3072 waves, which a 62-op list
  // hides and a 3-op list does not; a counter per group of eight workgroups balances too little -- the 62-op list
  // lost 6 % -- eight counters cost nothing and balance like one: 2 / 3 / 5 ops at 1 M sites 225 / 261 / 311 us
  // with one counter, 134 / 180 / 239 with eight; per-level launches 134 / 200 / 282.)  The tile after this one is
  // asked for during the last-but-one op, so that nobody waits for the answer (no earlier: a wave would sit on a
  // tile it only begins a whole list later); static rounds ask a word of the wave's own sink instead.
  const unsigned int tile_group = (blockIdx.x / 8u) % tile_groups;
  const unsigned long long my_counter = (unsigned long long)(uintptr_t)(next_tile + (size_t)tile_group * 32u);
  const unsigned long long no_counter = (unsigned long long)(uintptr_t)(sink + 80);
  // where this lane's 16 bytes of tip characters of the first batch of rows begin (+ the tile's first site)
  const unsigned long long row0 = bases.rowtab[lane];
  // (8 rate categories: a tile is 8 sites -- a lane fetches 16 bytes all the same and uses the first 8)
  static_assert((TS >= 16 || TS == 8) && 1024 % TS == 0, "a tip row of a tile is a whole number of 16-byte lanes, or half of one");

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
  size_t static_rounds = items / nwaves > dynamic_rounds ? items / nwaves - dynamic_rounds : 1; // (the last rounds from the counter)
  if (!next_tile) static_rounds = ~(size_t)0;
  size_t round = 0;
  for (size_t item = wave; item < items;)
  {
    // the item's segment: its records, its length, where its first character rows are
    size_t tile = item;
    const FusedRec * plan = plan0;
    unsigned int nops = nops0;
    unsigned long long row_first = row0;
    if (nsegs > 1u)
    {
      unsigned int seg = 0;
      while (tile >= tiles)
      {
        tile -= tiles;
        ++seg;
      }
      const const_words sg = (const_words)(unsigned long long)(bases.segs + __builtin_amdgcn_readfirstlane(seg));
      plan = plan0 + sg[0];
      nops = sg[1];
      row_first = bases.rowtab[(size_t)sg[2] * 64u + lane];
    }
    const size_t site0 = (size_t)site_base + tile * TS; // (site_base: this launch's block of the alignment)
#ifdef PLLHIP_FUSED_TIMING
    const unsigned long long t_tile = __builtin_readcyclecounter();
#endif
    unsigned int next_ticket = 0;
    const unsigned long long ticket_counter = (next_tile && round + 1 >= static_rounds) ? my_counter : no_counter;
    const size_t clv_off = site0 * (W * 16u);                             // bytes into a CLV
    const size_t cnt_off = site0 * ((MODE == SCALE_RATE) ? RC * 4u : 4u); // bytes into a scale buffer

    // The tip characters.  Lane l holds 16 bytes -- this tile's sites -- of one tip row (of a TS / 16-th of
    // one): the rows of up to 1024 / TS tip operands in the order the list uses them, ALL fetched by the one
    // load below at the top of the tile.  (Until round 3 every op requested its own two rows two ops ahead:
    // a few bytes from two pages of their own per op, whose address translations the write streams keep
    // evicting -- with 8 M sites per row that cost the 133 GB partition a quarter of its speed; tool builds
    // that read every row from the first row's pages: 0.56 -> 0.76 of the HBM peak, profiles/r3_footprint.txt.
    // Sixty-four rows in one instruction wait for their translations together, long before they are used.)
    pll_v4u cs = *(const pll_v4u PLL_GLOBAL *)(row_first + site0);
    // What an op needs from memory besides is requested TWO ops ahead of it: its 16 bytes of [P_l | P_r].
    auto request = [&](FusedFetch<J> & f, const Rec & r) {
      const unsigned int off = (pm_left ? rec_req_lmat(r) : rec_req_rmat(r)) + pm_lane;
      f.pm = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(bases.pmat) + off);
      if (RC == 8) f.pm2 = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(bases.pmat) + rec_req_rmat(r) + pm_lane);
    };
    // A lane's characters of sub-step j of a row: byte j * SPS + tip_lane of the row's TS bytes, i.e. one of
    // the SPS / 4 words that `readlane` brings from the lane(s) holding them (wave-uniform positions).
    auto row_code = [&](unsigned int pos, unsigned int j) -> unsigned int {
      unsigned int word = 0;
#pragma unroll
      for (unsigned int m = 0; m < SPS / 4; ++m)
      {
        const unsigned int w = j * (SPS / 4) + m; // word of the row
        const unsigned int v = (w & 3u) == 0 ? cs.x : (w & 3u) == 1 ? cs.y : (w & 3u) == 2 ? cs.z : cs.w;
        const unsigned int sw = (unsigned int)__builtin_amdgcn_readlane((int)v, (int)(pos + w / 4));
        word = ((tip_lane >> 2) == m) ? sw : word;
      }
      return (word >> ((tip_lane & 3u) * 8u)) & 255u;
    };
    // ... and ONE op ahead its entries of the pair table are gathered (table 0, all zeros, for an op
    // without a tip; an absent tip's character is 0)
    auto gather = [&](double2 (&pt)[J], const Rec & r) {
      const unsigned int ch = rec_chars(r);
      const unsigned int lmask = (ch & PLLHIP_FUSED_CH_LTIP) ? 15u : 0u, rmask = (ch & PLLHIP_FUSED_CH_RTIP) ? 15u : 0u;
#pragma unroll
      for (unsigned int j = 0; j < J; ++j)
      {
        const unsigned int pair = ((row_code(PLLHIP_FUSED_CH_LPOS(ch), j) & lmask) << 4) | (row_code(PLLHIP_FUSED_CH_RPOS(ch), j) & rmask);
        const unsigned int off = pair * (W * 16u) + gat_lane + rec_gather(r);
        pt[j] = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(bases.pairtab) + off);
      }
      // the list moves on to rows this batch does not hold (rare: every 1024 / TS tip operands): the lanes'
      // addresses of the next batch, then its rows.  Assembly, like reload(): the compiler counts no load
      // here (a load inside a branch makes it wait for everything in flight on every path), the next op's
      // first wait -- for a request issued after this -- covers it.
      if (ch & PLLHIP_FUSED_CH_LOAD)
      {
        const unsigned long long tab = (unsigned long long)(bases.rowtab + (size_t)(ch >> 24) * 64u);
        unsigned long long a;
        asm volatile("global_load_dwordx2 %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=v"(a) : "v"(lane * 8u), "s"(tab) : "memory");
        a += site0;
        asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(cs) : "v"(a) : "memory");
      }
    };

    // The next op's matrices: the wave's coalesced block goes through LDS, each lane takes rows
    // 2h, 2h+1 of category k (own column pair first, then the partner's).  One matrix at a time
    // through ONE staging block (LDS operations of a wave execute in order): the 512 bytes
    // this saves per wave are what gives the 12-wave configuration its sixth slot (a balanced
    // 128-taxon tree needs six).  `need`: 2 both matrices, 1 the right one only (the tip of a
    // tip-inner op comes from its pair table), 0 none (tip-tip).
    auto stage_rows = [&](const FusedFetch<J> & f, unsigned int need, half_rows & pl, half_rows & pr) {
      double2 * p = pst;
      if (need >= 2)
      {
        if (lane < MG) p[lane] = f.pm;
        // (what a lane reads was written by OTHER lanes: the compiler, which reasons per
        // thread, must not carry a value over these lines)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 2; ++r)
        {
          const unsigned int row = k * 8 + (2 * h + r) * 2;
          const double2 lo = p[row + h], lp = p[row + 1 - h];
          pl.m[r][0] = lo.x; pl.m[r][1] = lo.y; pl.m[r][2] = lp.x; pl.m[r][3] = lp.y;
        }
        asm volatile("" ::: "memory");
      }
      if (need >= 1)
      {
        if (RC == 8) p[lane] = f.pm2;
        else if (lane >= MG && lane < 2 * MG) p[lane - MG] = f.pm;
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 2; ++r)
        {
          const unsigned int row = k * 8 + (2 * h + r) * 2;
          const double2 ro = p[row + h], rp = p[row + 1 - h];
          pr.m[r][0] = ro.x; pr.m[r][1] = ro.y; pr.m[r][2] = rp.x; pr.m[r][3] = rp.y;
        }
        asm volatile("" ::: "memory");
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
    auto reload = [&](unsigned int src_index) {
      const const_quads q = (const_quads)(unsigned long long)(bases.srcs + src_index);
      const unsigned long long src[2] = {q[0], q[1]}, csrc[2] = {q[2], q[3]};
      const unsigned long long where = q[4]; // lslot_b | rslot_b << 16 | lcnt_b << 32 | rcnt_b << 48
      const unsigned int slot_b[2] = {(unsigned int)where & 0xffffu, (unsigned int)(where >> 16) & 0xffffu};
      const unsigned int count_b[2] = {(unsigned int)(where >> 32) & 0xffffu, (unsigned int)(where >> 48)};
#pragma unroll
      for (int o = 0; o < 2; ++o)
      {
        if (src[o])
        {
          const char * base = reinterpret_cast<const char *>(src[o]) + clv_off;
#pragma unroll
          for (unsigned int j = 0; j < J; ++j)
          {
            const unsigned int lds_b = clv_lds_b + slot_b[o] + j * 1024u;
            const char * gsrc = base + j * 1024u + lane16;
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
          const char * cbase = reinterpret_cast<const char *>(csrc[o]) + cnt_off;
          const unsigned int lds_b = cnt_lds_b + count_b[o];
          const unsigned long long mask = (J * CW >= 64u) ? ~0ull : ((1ull << (J * CW)) - 1ull);
          const char * gsrc = cbase + lane * 4u;
          unsigned long long exec_saved;
          unsigned int m0_saved;
          asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 %1, m0\n\ts_mov_b64 exec, %2\n\ts_mov_b32 m0, %3\n\t"
                       "s_nop 0\n\tglobal_load_lds_dword %4, off\n\ts_mov_b32 m0, %1\n\ts_mov_b64 exec, %0"
                       : "=&s"(exec_saved), "=&s"(m0_saved) : "s"(mask), "s"(lds_b), "v"(gsrc) : "memory");
        }
      }
    };

    // The pipeline.  While op i runs, the wave has in registers: the matrix rows of op i (pl, pr)
    // and its pair-table entries (pt_use), both fetched during op i-1; the block and characters
    // of op i+1 (fa), requested during op i-1; and it requests those of op i+2 (fb).  Record i
    // says all of that -- what to request for op i+2, which table to gather from for op i+1,
    // what op i+1 reloads and which matrices it needs, and op i itself -- so ONE record is live
    // per op and the next one is in flight (scalar loads return out of order and share a
    // counter with LDS, so a field consumed right after its load would drain the LDS reads in
    // flight).  Two header records ahead of the list do for ops 0 and 1 what the records of
    // ops -2 and -1 would.
    const Rec h0 = rec_load(plan, 0), h1 = rec_load(plan, 1);
    Rec ra = rec_load(plan, 2), rb;
    FusedFetch<J> fa, fb;
    half_rows pl, pr;
    double2 pta[J], ptb[J];
    // The prologue issues its memory operations in the order two ops would -- requests, gather,
    // stores (to the sink) -- because the compiler counts the operations issued after a load to
    // know how many may stay in flight when the load is consumed, and takes the minimum over
    // all paths into the loop: with the same shape on the entry path, the wait at the top of
    // an op leaves the previous op's stores in flight.
    auto sink_stores = [&]() {
#pragma unroll
      for (unsigned int j = 0; j < J; ++j)
      {
        st16<NT>(sink + lane, 0.0, 0.0);
        asm volatile("" ::: "memory"); // (J stores, not one)
      }
      if (MODE != SCALE_NONE) sink_cnt[lane] = 0u;
    };
    request(fb, h0);
    sink_stores();
    if (rec_flags(h1) & PLLHIP_FUSED_RELOAD_NEXT) reload(rec_src(h1));
    gather(pta, h1);
    request(fa, h1);
    asm volatile("" ::"v"(fb.pm.x), "v"(fb.pm.y) : "memory");
    if (RC == 8) asm volatile("" ::"v"(fb.pm2.x), "v"(fb.pm2.y) : "memory");
    sink_stores();
    stage_rows(fb, (rec_flags(h1) >> PLLHIP_FUSED_STAGE_SHIFT) & 3u, pl, pr);

#ifdef PLLHIP_FUSED_TIMING
    // (tool build, tools/fused_timing.sh: where a wave's cycles go, printed by a few waves)
    unsigned long long seg[6] = {0, 0, 0, 0, 0, 0};
#define PLLHIP_TICK(n) { const unsigned long long t_now = __builtin_readcyclecounter(); seg[n] += t_now - t_last; t_last = t_now; }
    unsigned long long t_last = __builtin_readcyclecounter();
    const unsigned long long t_prologue = t_last - t_tile;
#else
#define PLLHIP_TICK(n)
#endif
    // One op.  r0: its record, r1: where the next one is loaded to; fu: the fetch of op i+1,
    // ff: where that of op i+2 goes; pu: its pair-table entries, pf: where those of op i+1 go.
    // The caller alternates the two of each, so that nothing loaded is ever copied (a copy
    // would have to wait for the load).
    auto step = [&](const Rec & r0, Rec & r1, const FusedFetch<J> & fu, FusedFetch<J> & ff, const double2 (&pu)[J],
                    double2 (&pf)[J], unsigned int i) __attribute__((always_inline)) {
      // A wave is the limit of this kernel, not HBM: with twelve waves per CU nothing hides
      // what a wave waits for itself (counters: ~2300 cycles per op of which ~500 issue
      // vector and ~270 scalar instructions).  So the order below overlaps the wave's own
      // latencies: operands are read from LDS FIRST and the look-ahead requests are
      // computed and issued while they arrive; the next op's matrix rows go through LDS right
      // after the arithmetic and arrive during the stores and the counts.
      const unsigned int fl = rec_flags(r0);
      const unsigned int kind = fl & PLLHIP_FUSED_KIND_MASK;
      const bool has_slot = fl & PLLHIP_FUSED_HAS_PSLOT;
      const bool scaling = MODE != SCALE_NONE && (fl & PLLHIP_FUSED_SCALING);
      const unsigned long long out = rec_parent(r0) + clv_off; // (uniform)
      char * lds_l = reinterpret_cast<char *>(clv) + rec_lslot(r0) + lane16;
      char * lds_r = reinterpret_cast<char *>(clv) + rec_rslot(r0) + lane16;
      constexpr unsigned int GW = (MODE == SCALE_RATE) ? 2u : W; // lanes that share a count
      constexpr unsigned int EPS = 64u / GW, E = J * EPS;        // entries per sub-step / per tile
      static_assert(MODE == SCALE_NONE || CW == EPS, "one count word per entry of a sub-step");
      const unsigned int t = lane % E; // (lanes beyond the entries repeat them: same values to the same addresses)

      // Everything requested one op ago has arrived once fu's matrix block is used -- and with
      // it what that iteration's reload() copied into this op's slots (issued ahead of those
      // requests; memory operations return in order).  No slot is read above this line.
      asm volatile("" ::"v"(fu.pm.x), "v"(fu.pm.y) : "memory");
      if (RC == 8) asm volatile("" ::"v"(fu.pm2.x), "v"(fu.pm2.y) : "memory");
      // operands and the counts they bring along (entry t of the tile -- a site, or a (site, rate)
      // with per-rate scalers -- is lane t's); every kind reads both operands (a tip-tip op reads
      // slot 0 for nothing: no branch)
      double2 lo[J], ro[J];
#pragma unroll
      for (unsigned int j = 0; j < J; ++j)
      {
        lo[j] = *reinterpret_cast<const double2 *>(lds_l + j * 1024u);
        ro[j] = *reinterpret_cast<const double2 *>(lds_r + j * 1024u);
      }
      unsigned int lc = 0u, rc = 0u;
      if (MODE != SCALE_NONE)
      {
        lc = *reinterpret_cast<unsigned int *>(reinterpret_cast<char *>(cnt) + rec_lcnt(r0) + t * 4u);
        rc = *reinterpret_cast<unsigned int *>(reinterpret_cast<char *>(cnt) + rec_rcnt(r0) + t * 4u);
      }
      asm volatile("" ::: "memory"); // (the reads above are issued before what follows)
      // (rare, wave-uniform: the sources of the reload are read on the spot)
      if (fl & PLLHIP_FUSED_RELOAD_NEXT) reload(rec_src(r0));
      // the ticket for the next tile (lane 0 only, through EXEC; assembly: the compiler sees no memory
      // operation, and operations return in order -- the next op's first wait covers it)
      if (i + 2 == nops)
        asm volatile("s_mov_b64 exec, 1\n\tglobal_atomic_add %0, %1, %2, off sc0\n\ts_mov_b64 exec, -1"
                     : "=&v"(next_ticket) : "v"(ticket_counter), "v"(1u) : "memory");
      // (the request last: what the next op waits for first is the youngest operation in flight)
      gather(pf, r0);
      request(ff, r0);
      PLLHIP_TICK(0)
      // (every load is consumed on every path, needed or not: the registers of a load that
      // nobody waited for stay "pending" for the compiler, and it drains the queue -- this op's
      // predecessor's stores included -- when it next reuses them)
#pragma unroll
      for (unsigned int j = 0; j < J; ++j) asm volatile("" ::"v"(pu[j].x), "v"(pu[j].y));
      PLLHIP_TICK(1)
      unsigned long long scaled[J];
      double p0[J], p1[J];
      if (kind == 2)
      {
        // tip-tip: the finished entries come from the pair table; never scales, clears its counts
        // (core_partials_avx.c:598-599)
#pragma unroll
        for (unsigned int j = 0; j < J; ++j)
        {
          scaled[j] = 0;
          p0[j] = pu[j].x;
          p1[j] = pu[j].y;
        }
      }
      else
      {
#pragma unroll
        for (unsigned int j = 0; j < J; ++j)
        {
          const double2 rp = make_double2(dpp_pair_swap(ro[j].x), dpp_pair_swap(ro[j].y));
          double x0, x1;
          if (kind == 1)
          {
            // tip-inner: the tip's factor is its pair-table entry
            x0 = pu[j].x;
            x1 = pu[j].y;
          }
          else
          {
            const double2 lp = make_double2(dpp_pair_swap(lo[j].x), dpp_pair_swap(lo[j].y));
            x0 = pl.dot(0, lo[j], lp);
            x1 = pl.dot(1, lo[j], lp);
          }
          double q0 = x0 * pr.dot(0, ro[j], rp), q1 = x1 * pr.dot(1, ro[j], rp);
          // scaling rule of core_partials_avx.c:486-527: all entries of the site (of the rate,
          // with per-rate scalers) below the threshold; x * 1.0 is x
          scaled[j] = 0;
          if (MODE != SCALE_NONE)
          {
            const unsigned int small = ((q0 < PLLHIP_SCALE_THRESHOLD) & (q1 < PLLHIP_SCALE_THRESHOLD)) ? 1u : 0u;
            const bool scale = scaling && group_and<(int)GW>(small) != 0u;
            const double factor = __hiloint2double(scale ? 0x4ff00000 : 0x3ff00000, 0); // 2^256 : 1.0
            q0 *= factor;
            q1 *= factor;
            scaled[j] = __ballot(scale); // (wave-uniform: which groups of this sub-step were scaled)
          }
          p0[j] = q0;
          p1[j] = q1;
        }
      }
      // (the arithmetic is done with pl, pr: their registers take the next op's rows, whose
      // LDS round trip runs while the stores below are issued)
      asm volatile("" ::"v"(p0[J - 1]), "v"(p1[J - 1]), "v"(fu.pm.x), "v"(fu.pm.y) : "memory");
      PLLHIP_TICK(2)
      // the next record: a scalar load, and scalar loads return out of order -- while one is in
      // flight every wait for an LDS read or an earlier record field becomes a wait for
      // everything.  Here nothing of that kind is waited for until the next op begins.
      r1 = rec_load(plan, i + 3);
      stage_rows(fu, (fl >> PLLHIP_FUSED_STAGE_SHIFT) & 3u, pl, pr);
      PLLHIP_TICK(5)
      // The stores are common to all kinds and under no branch: the compiler counts the memory
      // operations of the path with the FEWEST of them to decide how many may stay in flight at a
      // wait, and a path without this op's stores would make the next op wait for the stores of
      // the op before.
      // (That goes for the parent's LDS slot too: `if (has_slot)` around its write made the
      // compiler duplicate the store into both arms, and the structurizer's path through
      // neither arm had one store less.  The write is masked through EXEC instead -- all lanes or
      // none -- in assembly, so that there is no branch to reason about.)
      const unsigned long long slot_mask = has_slot ? ~0ull : 0ull;
      const unsigned int lds_p_b = clv_lds_b + rec_pslot(r0) + lane16;
#pragma unroll
      for (unsigned int j = 0; j < J; ++j)
      {
        st16g<NT>(out, j * 1024u + lane16, p0[j], p1[j]);
        const pll_v2d v = {p0[j], p1[j]};
        asm volatile("s_mov_b64 exec, %0\n\tds_write_b128 %1, %2 offset:%3\n\ts_mov_b64 exec, -1"
                     :: "s"(slot_mask), "v"(lds_p_b), "v"(v), "n"(j * 1024u) : "memory");
      }
      PLLHIP_TICK(3)
      // The tile's counts, once per op: inherited counts plus one if the sub-step that held the
      // entry scaled its group; all of them leave in ONE store (64 contiguous bytes per tile
      // with per-site counts).  (An op without a scale buffer stores to the sink.)
      if (MODE != SCALE_NONE)
      {
        unsigned long long mine = scaled[0];
#pragma unroll
        for (unsigned int j = 1; j < J; ++j) mine = (t / EPS == j) ? scaled[j] : mine;
        const unsigned int bit = (unsigned int)(mine >> ((t % EPS) * GW)) & 1u;
        // (tip operands inherit nothing)
        if (!(fl & PLLHIP_FUSED_LCNT)) lc = 0u;
        if (!(fl & PLLHIP_FUSED_RCNT)) rc = 0u;
        const unsigned int count = scaling ? lc + rc + bit : 0u;
        if (has_slot && lane < E) *reinterpret_cast<unsigned int *>(reinterpret_cast<char *>(cnt) + rec_pcnt(r0) + t * 4u) = count;
        const unsigned long long cdst = scaling ? rec_pscaler(r0) + cnt_off : (unsigned long long)(uintptr_t)sink_cnt; // (uniform)
        // (non-temporal like the tiles once the partition exceeds the translation caches' reach: as
        // plain stores -- half a cache line each, which L2 completes by reading the other half --
        // the counts cost the 126-op list of a 16 GB partition a fifth of its speed, 0.56 against
        // 0.685 of the HBM peak on one box, the 198-op list 0.53 against 0.62; below the reach plain
        // stores are the better ones by 0-3 %: neighbouring tiles' halves meet in L2)
        if (NTP == 2) __builtin_nontemporal_store(count, (unsigned int PLL_GLOBAL *)(cdst + t * 4u));
        else *(unsigned int PLL_GLOBAL *)(cdst + t * 4u) = count;
      }
      PLLHIP_TICK(4)
    };
    for (unsigned int i = 0;;)
    {
      step(ra, rb, fa, fb, pta, ptb, i);
      if (++i == nops) break;
      step(rb, ra, fb, fa, ptb, pta, i);
      if (++i == nops) break;
    }
#ifdef PLLHIP_FUSED_TIMING
    if (lane == 0 && round == 3 && (wave == 0 || wave == 1001 || wave == 2002 || wave == 3003))
      printf("wave %u ops %u: tile %llu cycles, of which prologue %llu; per op: top (records, requests arrive, gather) %llu pair entries arrive %llu arithmetic %llu stores %llu counts %llu stage %llu\n",
             (unsigned int)wave, nops, (unsigned long long)__builtin_readcyclecounter() - t_tile, t_prologue,
             seg[0] / nops, seg[1] / nops, seg[2] / nops, seg[3] / nops, seg[4] / nops, seg[5] / nops);
#endif
    if (++round < static_rounds)
    {
      item += nwaves;
      continue;
    }
    asm volatile("" : "+v"(next_ticket));
    item = static_rounds * nwaves + tile_group + (size_t)tile_groups * (unsigned int)__builtin_amdgcn_readfirstlane((int)next_ticket);
  }
}

// ---------------------------------------------------------------- host: order, slots, launch

namespace
{
// (round 4) The planner's many small lists -- hazard predecessors, readers per buffer, uses per value -- come from a
// bump allocator over a per-thread buffer: as std::vectors on the heap they were most of its time (0.32 us per op,
// 20 us for BASELINE config 2's 62-op list, 75 us for a 198-op list; VERDICT r3 Weak 6: tree search hands over a new
// list almost every call).
typedef std::pmr::vector<unsigned int> PlanList;
typedef std::pmr::vector<PlanList> PlanLists;
struct Node
{
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
  // (four workgroups -- 16 waves of 128 registers, four slots -- were measured for short lists in round 3 and are
  // slower than three at every list length: 2 / 3 / 5 / 7 / 15 ops 257 / 286 / 335 / 282 / 421 us against 226 / 260 /
  // 310 / 246 / 411; the variant spills seven registers)
  const size_t budget = PLLHIP_FUSED_J == 1 ? 9472 : (wgs >= 4 ? 9984 : wgs >= 3 ? 13312 : 16384); // J = 1: four workgroups per CU
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
                               unsigned int * reloads_out, std::pmr::memory_resource * pool)
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
  PlanLists uses(count, pool); // positions at which each list value is read
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
  static thread_local std::vector<char> arena(256 * 1024);
  std::pmr::monotonic_buffer_resource pool(arena.data(), arena.size()); // (beyond the buffer: the heap)
  std::vector<Node> node(count);
  PlanLists hard_of(count, &pool); // WAW / WAR predecessors (and scaler hazards) of each op: must run before it
  const size_t nclv = geom.nclv, nsc = geom.nsc;
  // last writer and readers-since of every CLV / scale buffer, in list order
  std::pmr::vector<int> clv_w(nclv, -1, &pool), sc_w(nsc, -1, &pool);
  PlanLists clv_r(nclv, &pool), sc_r(nsc, &pool);
  for (unsigned int i = 0; i < count; ++i)
  {
    const pllhip_op_t & op = ops[i];
    Node & nd = node[i];
    nd.raw[0] = clv_w[op.child1_clv];
    nd.raw[1] = clv_w[op.child2_clv];
    // (round 6: a predecessor that is one of the op's two PRODUCERS is no hazard of its own -- the usual case: a
    // child's scale buffer was written by the op that wrote the child -- and must not be walked ahead of the
    // "heavier child first" rule below.  Until then every list with scale buffers was walked child 1 first, whatever the
    // subtrees' sizes: a traversal directed at a deep edge of a balanced 64-taxon tree needed 4 operands copied back
    // from HBM with five slots and ended in a run of seven matrix ops; now none, and at most three in a row.)
    // (PLLHIP_FUSED_ORDER=0, a developer's switch: the old walk, for A/B measurements)
    static const bool producers_are_no_hazards = !(pllhip_env("PLLHIP_FUSED_ORDER") && atoi(pllhip_env("PLLHIP_FUSED_ORDER")) == 0);
    auto hard = [&](int p) {
      if (p >= 0 && (unsigned int)p != i && !(producers_are_no_hazards && (p == nd.raw[0] || p == nd.raw[1])))
        hard_of[i].push_back((unsigned int)p);
    };
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
    PlanLists & preds = hard_of; // (the two producers are appended: hazards first, then the heavier child)
    for (unsigned int i = 0; i < count; ++i)
    {
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

  const int rc = assign_slots_reload(geom, ops, args, kinds, count, nslots, order, pos_of, node, plan, reloads_out, &pool);
  if (rc) return rc;
  if (pllhip_env("PLLHIP_FUSED_DEBUG"))
  {
    fprintf(stderr, "pllhip fused plan: %u ops, %u slots, %u operands reloaded from HBM\n", count, nslots, *reloads_out);
    if (atoi(pllhip_env("PLLHIP_FUSED_DEBUG")) > 1)
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
  const size_t nsegs = bases.nsegs; // (work items: (tile, segment) pairs)
  size_t grid = (tiles * nsegs + 3) / 4;
#ifdef PLLHIP_FUSED_WPS4
  // (tool build, tools/fused_wps4.sh: four workgroups per CU -- sixteen waves of 128 registers, four slots each)
  const bool four = J != 1 && nslots <= pllhip_fused_slots(c, 4);
#else
  const bool four = false;
#endif
  const size_t cap = (size_t)c->num_cus * (J == 1 || four ? 4 : (nslots <= pllhip_fused_slots(c, 3) ? 3 : 2));
  if (grid > cap) grid = cap;
  const bool nt = pllhip_use_nt(c);
  const bool static_tiles = pllhip_env("PLLHIP_FUSED_STATIC_TILES") != nullptr;
  // Partitions beyond 8 GB (CLVs + scale buffers): the counts are stored non-temporally like the
  // tiles (see the kernel).  8 GB is what the address-translation caches reach (4096 pages of
  // 2 MB): every shape ran at 0.59-0.64 of the HBM peak up to there and at 0.46-0.50 beyond
  // (profiles/r2_footprint.txt), which looked like the price of the page walks themselves -- and
  // walking every other launch backwards, so that it began in the pages still cached, did recover
  // part of it (0.51 -> 0.56-0.62).  The walks only hurt READS, though, and the one read this
  // write stream had was L2 completing the half cache lines of the count stores.  With those
  // non-temporal the same lists run at 0.69-0.75 and the direction makes no difference any more
  // (0.747 / 0.746, 0.688 / 0.689): the alternating walk is gone.
  const size_t footprint = c->clv_arena_bytes + (size_t)c->sh.scale_buffers * c->scaler_stride * sizeof(unsigned int);
  const bool beyond_reach = footprint > (size_t)4096 * ((size_t)2 << 20);
  // a third of a wave's rounds of tiles come from the counter (see the kernel: the XCDs' unequal
  // write rates): seven of the ~21 rounds of 1 M sites (the measured optimum there), and in proportion
  // for longer alignments (8 M sites x 128 taxa: 7 rounds 0.565, 30 0.584, 54 0.583 of the HBM peak;
  // 2 M sites: 7 rounds 0.652, 14 0.672); short lists two, see the kernel
  // PLLHIP_FUSED_BLOCK_SITES (an experiment kept as a switch; default one launch): the alignment walked in
  // BLOCKS of sites, one launch each, every launch taking its block through the whole list.  Measured on the
  // 133 GB partition (8 M sites x 128 taxa) with blocks of 4 M ... 500 k sites: 0.557 / 0.558 / 0.554 / 0.547 of
  // the HBM peak against 0.555 in one launch -- what that partition lost it lost on the reads of the tip
  // characters, not on the footprint of a launch (profiles/r3_footprint.txt).
  size_t block_sites = sites;
  {
    const char * e = pllhip_env("PLLHIP_FUSED_BLOCK_SITES");
    if (e) block_sites = atoi(e) > 0 ? (size_t)atoi(e) : sites;
    block_sites = (block_sites + 255) / 256 * 256; // (whole tiles, whole rounds)
    if (block_sites > sites) block_sites = sites;
  }
  for (size_t base = 0; base < sites; base += block_sites)
  {
  const unsigned int bsites = (unsigned int)(sites - base < block_sites ? sites - base : block_sites);
  const size_t btiles = (bsites + tile_sites - 1) / tile_sites;
  size_t bgrid = (btiles * nsegs + 3) / 4;
  if (bgrid > cap) bgrid = cap;
  // (the two sets of tile counters in turn: see the kernel)
  unsigned int * const tile_counter = static_tiles ? nullptr : c->d_tile_counter + (size_t)c->tile_counter_phase * (PLLHIP_TILE_COUNTER_BYTES / 4);
  unsigned int * const reset_tiles = static_tiles ? nullptr : c->d_tile_counter + (size_t)(c->tile_counter_phase ^ 1u) * (PLLHIP_TILE_COUNTER_BYTES / 4);
  c->tile_counter_phase ^= 1u;
  const size_t rounds = btiles * nsegs / (bgrid * 4);
  const unsigned int dynamic_rounds = pllhip_env("PLLHIP_FUSED_DYNAMIC_ROUNDS") ? (unsigned int)atoi(pllhip_env("PLLHIP_FUSED_DYNAMIC_ROUNDS"))
                                      : (c->fused_last_longest >= 32 ? (unsigned int)std::max<size_t>(7, rounds / 3) : 2u);
  // (eight counters, or what PLLHIP_FUSED_TILE_GROUPS says -- but never more than there are groups of eight
  // workgroups, or a counter's tiles would have no takers; and no more than the counter buffer holds)
  const size_t want_groups = pllhip_env("PLLHIP_FUSED_TILE_GROUPS") ? (size_t)std::max(1, atoi(pllhip_env("PLLHIP_FUSED_TILE_GROUPS"))) : 8;
  const unsigned int tile_groups = (unsigned int)std::min<size_t>(std::min<size_t>(want_groups, PLLHIP_TILE_COUNTER_BYTES / 128),
                                                                  std::max<size_t>(1, bgrid / 8));
#define LAUNCH_FUSED_ARGS (unsigned int)bgrid, 256, lds, c->stream>>>( \
      d_plan, bases, count, bsites, nslots, (double2 *)c->d_sink, tile_counter, dynamic_rounds, (unsigned int)base, tile_groups, \
      reset_tiles)
#ifdef PLLHIP_FUSED_WPS4
#define LAUNCH_FUSED(MODEV, NTV)                                                      \
  do {                                                                                \
    if (four) k_dna_fused<RC, J, MODEV, NTV, 4><<<LAUNCH_FUSED_ARGS;                  \
    else k_dna_fused<RC, J, MODEV, NTV, (J == 1 ? 4 : 3)><<<LAUNCH_FUSED_ARGS;        \
  } while (0)
#else
#define LAUNCH_FUSED(MODEV, NTV) k_dna_fused<RC, J, MODEV, NTV, (J == 1 ? 4 : 3)><<<LAUNCH_FUSED_ARGS
#endif
#define LAUNCH_FUSED_MODE(NTV)                         \
  do {                                                  \
    if (mode == SCALE_NONE) LAUNCH_FUSED(0, NTV);       \
    else if (mode == SCALE_SITE) LAUNCH_FUSED(1, NTV);  \
    else LAUNCH_FUSED(2, NTV);                          \
  } while (0)
  if (!nt) LAUNCH_FUSED_MODE(0);
  else if (!beyond_reach && c->nt_override != 2) LAUNCH_FUSED_MODE(1);
  else LAUNCH_FUSED_MODE(2);
#undef LAUNCH_FUSED_MODE
#undef LAUNCH_FUSED
#undef LAUNCH_FUSED_ARGS
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// Where each op's tip characters sit in a wave's character registers.  A wave holds the characters of 64 / lpr tip
// rows of its tile at a time (lpr lanes of 16 bytes per row); the rows are dealt in the order the list uses them,
// both rows of an op into the same batch.  tips[pos]: bit 0 / 1 = the op has a left / right tip.  Out, per op:
// chars (PLLHIP_FUSED_CH_LPOS / RPOS = first lane of the row, CH_LTIP / CH_RTIP) and the batch its rows are in
// (an op without tips: the batch current at that point).  Returns the number of batches.  Pure host logic.
unsigned int pllhip_fused_char_batches(const unsigned int * tips, unsigned int count, unsigned int lpr,
                                       unsigned int * chars_out, unsigned int * batch_out)
{
  const unsigned int rpb = 64 / lpr; // rows per batch
  unsigned int batch = 0, q = 0;
  for (unsigned int pos = 0; pos < count; ++pos)
  {
    const unsigned int n = (tips[pos] & 1u) + ((tips[pos] >> 1) & 1u);
    if (q + n > rpb)
    {
      ++batch;
      q = 0;
    }
    batch_out[pos] = batch;
    chars_out[pos] = 0;
    if (tips[pos] & 1u) chars_out[pos] |= PLLHIP_FUSED_CH_LTIP | (q++ * lpr);
    if (tips[pos] & 2u) chars_out[pos] |= PLLHIP_FUSED_CH_RTIP | (q++ * lpr) << 8;
  }
  return batch + 1;
}

extern "C" unsigned int pllhip_fused_char_batches_dry(const unsigned int * tips, unsigned int count, unsigned int rate_cats,
                                                      unsigned int * chars_out, unsigned int * batch_out)
{
  const unsigned int ts = PLLHIP_FUSED_J * (64 / (2 * rate_cats));
  return pllhip_fused_char_batches(tips, count, ts >= 16 ? ts / 16 : 1, chars_out, batch_out);
}

// Independent sub-lists (FusedSeg in partials_fused.hpp).  Two ops belong together when one of them writes a
// buffer -- CLV or scale buffer -- the other reads or writes; buffers nobody in the list writes (tips, operands of
// earlier calls) tie nothing.  Union-find over the ops, then the components dealt longest first.
unsigned int pllhip_fused_segments(const FusedGeom & geom, const pllhip_op_t * ops, unsigned int count,
                                   unsigned int max_segments, std::vector<unsigned int> & seg_of)
{
  seg_of.assign(count, 0u);
  if (max_segments > PLLHIP_FUSED_MAX_SEGS) max_segments = PLLHIP_FUSED_MAX_SEGS;
  if (count < 4 || max_segments < 2) return 1;
  // (every new list passes here: the work arrays are the thread's, not the heap's)
  static thread_local std::vector<unsigned int> parent, size, roots, load, seg_of_root;
  static thread_local std::vector<int> clv_first, sc_first;
  parent.resize(count);
  for (unsigned int i = 0; i < count; ++i) parent[i] = i;
  auto find = [&](unsigned int x) {
    while (parent[x] != x) x = parent[x] = parent[parent[x]];
    return x;
  };
  auto join = [&](unsigned int a, unsigned int b) {
    a = find(a);
    b = find(b);
    if (a != b) parent[a > b ? a : b] = a < b ? a : b;
  };
  // first op that touches each WRITTEN buffer (-1: not written in this list, -2: written, nobody met yet)
  clv_first.assign(geom.nclv, -1);
  sc_first.assign(geom.nsc, -1);
  for (unsigned int i = 0; i < count; ++i)
  {
    if (ops[i].parent_clv >= geom.nclv || ops[i].child1_clv >= geom.nclv || ops[i].child2_clv >= geom.nclv ||
        ops[i].parent_scaler >= (int)geom.nsc || ops[i].child1_scaler >= (int)geom.nsc || ops[i].child2_scaler >= (int)geom.nsc)
      return 1; // (the path taken reports it)
    clv_first[ops[i].parent_clv] = -2;
    if (ops[i].parent_scaler >= 0) sc_first[ops[i].parent_scaler] = -2;
  }
  for (unsigned int i = 0; i < count; ++i)
  {
    auto touch = [&](std::vector<int> & first, int idx) {
      if (idx < 0 || first[idx] == -1) return;
      if (first[idx] == -2) first[idx] = (int)i;
      else join((unsigned int)first[idx], i);
    };
    touch(clv_first, (int)ops[i].parent_clv);
    touch(clv_first, (int)ops[i].child1_clv);
    touch(clv_first, (int)ops[i].child2_clv);
    touch(sc_first, ops[i].parent_scaler);
    touch(sc_first, ops[i].child1_scaler);
    touch(sc_first, ops[i].child2_scaler);
  }
  // components by size, longest first (ties: the one that begins first)
  size.assign(count, 0u);
  for (unsigned int i = 0; i < count; ++i) ++size[find(i)];
  roots.clear();
  for (unsigned int i = 0; i < count; ++i)
    if (size[i]) roots.push_back(i);
  if (roots.size() < 2) return 1;
  std::stable_sort(roots.begin(), roots.end(), [&](unsigned int a, unsigned int b) { return size[a] > size[b]; });
  // as many segments as have two ops each at least, and no more than shorten the longest
  unsigned int nsegs = (unsigned int)std::min<size_t>(max_segments, roots.size());
  seg_of_root.assign(count, 0u);
  for (;; --nsegs)
  {
    load.assign(nsegs, 0u);
    for (unsigned int r : roots)
    {
      const unsigned int k = (unsigned int)(std::min_element(load.begin(), load.end()) - load.begin());
      seg_of_root[r] = k;
      load[k] += size[r];
    }
    if (nsegs == 1 || *std::min_element(load.begin(), load.end()) >= 2) break;
  }
  if (nsegs == 1) return 1;
  // segment 0 the longest, as the header says and the kernels' segment-major order wants (every tile of the longest
  // segment first): the greedy dealing above does not give that by itself once components outnumber segments --
  // components of 5, 4 and 4 ops over two segments are loads of 5 and 8 -- so the segments are numbered by
  // descending load afterwards (tests/test_host.py)
  std::vector<unsigned int> rank_of(nsegs);
  {
    std::vector<unsigned int> by_load(nsegs);
    for (unsigned int k = 0; k < nsegs; ++k) by_load[k] = k;
    std::stable_sort(by_load.begin(), by_load.end(), [&](unsigned int a, unsigned int b) { return load[a] > load[b]; });
    for (unsigned int r = 0; r < nsegs; ++r) rank_of[by_load[r]] = r;
  }
  for (unsigned int i = 0; i < count; ++i) seg_of[i] = rank_of[seg_of_root[find(i)]];
  return nsegs;
}

extern "C" unsigned int pllhip_fused_segments_dry(unsigned int tips, unsigned int clv_buffers, unsigned int scale_buffers,
                                                  int pattern_tip, const pllhip_op_t * ops, unsigned int count,
                                                  unsigned int max_segments, unsigned int * seg_out)
{
  const FusedGeom geom = {(size_t)tips + clv_buffers, scale_buffers, tips, pattern_tip != 0};
  std::vector<unsigned int> seg_of;
  const unsigned int n = pllhip_fused_segments(geom, ops, count, max_segments, seg_of);
  for (unsigned int i = 0; i < count && seg_out; ++i) seg_out[i] = seg_of[i];
  return n;
}

// Encode the plans for the device (FusedRec in partials_fused.hpp) -- one per segment: two header records that
// stand for ops -2 and -1, one record per op, one of padding (the last op's look-ahead load) -- then the segment
// table, the reload sources, the pair-table jobs, the character rows' addresses.  Returns 1 if the list is not one
// the kernel takes.
int pllhip_launch_fused(pllhip_ctx * c, const std::vector<std::vector<FusedOp>> & plans, unsigned int nslots)
{
  const unsigned int nsegs = (unsigned int)plans.size();
  unsigned int count = 0, longest = 0;
  for (const auto & plan : plans)
  {
    count += (unsigned int)plan.size();
    longest = std::max(longest, (unsigned int)plan.size());
  }
  const unsigned int R = c->sh.rate_cats;
  if (count > PLLHIP_FUSED_MAX_OPS || nsegs < 1 || nsegs > PLLHIP_FUSED_MAX_SEGS) return 1;
  // pair tables of the ops with a tip (k_dna_pair_tables), carved from one device buffer behind a
  // table of zeros
  const size_t per = (size_t)256 * R * 4; // doubles per table
  size_t ntab = 1;
  for (const auto & plan : plans)
    for (const FusedOp & f : plan) ntab += (f.kind >= 1);
  if (ntab * per * sizeof(double) > 0xffffffffull ||
      (size_t)c->sh.prob_matrices * c->pmat_elems * sizeof(double) > 0xffffffffull)
    return 1;
  if (c->pairtab_elems < ntab * per)
  {
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->d_pairtab) HIP_TRY(hipFree(c->d_pairtab));
    c->d_pairtab = nullptr;
    HIP_TRY(hipMalloc((void **)&c->d_pairtab, ntab * per * sizeof(double)));
    HIP_TRY(hipMemsetAsync(c->d_pairtab, 0, per * sizeof(double), c->stream));
    c->pairtab_elems = ntab * per;
    ++c->layout_epoch;
  }
  // the row of zeros that lanes without a tip row fetch their "characters" from
  if (!c->fused_zero_row)
  {
    const size_t bytes = (size_t)c->sh.sites + PLLHIP_TAIL_SITES + 256;
    HIP_TRY(hipMalloc((void **)&c->fused_zero_row, bytes));
    HIP_TRY(hipMemsetAsync(c->fused_zero_row, 0, bytes, c->stream));
  }
  const unsigned int J = PLLHIP_FUSED_J;
  const unsigned int sps = 64 / (2 * R);
  const unsigned int cw = c->sh.rate_scalers ? 32 : (sps < 4 ? 4 : sps);
  const unsigned int slot_bytes = J * 1024u, count_bytes = J * cw * 4u;
  if ((size_t)nslots * slot_bytes > 0xffffu) return 1;

  std::vector<FusedRec> recs;
  std::vector<FusedSeg> segs(nsegs);
  std::vector<FusedSrc> srcs;
  std::vector<FusedPairJob> jobs;
  std::vector<unsigned long long> rowtab;
  int mode = SCALE_NONE;
  const unsigned int lpr = J * sps >= 16 ? J * sps / 16 : 1; // lanes per row
  for (unsigned int sg = 0; sg < nsegs; ++sg)
  {
  const std::vector<FusedOp> & plan = plans[sg];
  const unsigned int n = (unsigned int)plan.size();
  // Tip rows in the order the segment uses them, in batches of what a wave's 64 x 16 bytes hold of a tile
  // (pllhip_fused_char_batches).  rowtab[batch][lane]: the address the lane fetches from (+ the tile's first
  // site); lanes without a row fetch zeros.  Batches are numbered across the segments.
  std::vector<unsigned int> chars_of(n, 0), batch_of(n, 0);
  const unsigned int batch0 = (unsigned int)(rowtab.size() / 64);
  {
    std::vector<unsigned int> ntips(n);
    for (unsigned int pos = 0; pos < n; ++pos) ntips[pos] = (plan[pos].ltip ? 1u : 0u) | (plan[pos].rtip ? 2u : 0u);
    const unsigned int nbatches = pllhip_fused_char_batches(ntips.data(), n, lpr, chars_of.data(), batch_of.data());
    if (batch0 + nbatches > 255) return 1;
    rowtab.resize((size_t)(batch0 + nbatches) * 64, (unsigned long long)(uintptr_t)c->fused_zero_row);
    for (unsigned int pos = 0; pos < n; ++pos)
    {
      batch_of[pos] += batch0;
      const unsigned char * rows[2] = {plan[pos].ltip, plan[pos].rtip};
      const unsigned int lane0[2] = {PLLHIP_FUSED_CH_LPOS(chars_of[pos]), PLLHIP_FUSED_CH_RPOS(chars_of[pos])};
      for (int o = 0; o < 2; ++o)
        for (unsigned int l = 0; rows[o] && l < lpr; ++l)
          rowtab[(size_t)batch_of[pos] * 64 + lane0[o] + l] = (unsigned long long)(uintptr_t)rows[o] + l * 16u;
    }
  }
  std::vector<unsigned int> table_of(n, 0); // byte offset of each op's pair table (0: zeros)
  for (unsigned int pos = 0; pos < n; ++pos)
  {
    const FusedOp & f = plan[pos];
    if (f.kind >= 1)
    {
      const size_t index = jobs.size() + 1;
      table_of[pos] = (unsigned int)(index * per * sizeof(double));
      jobs.push_back(FusedPairJob{f.lmat, f.rmat, c->d_pairtab + index * per, f.kind == 2 ? 1ull : 0ull});
    }
    // every op with a parent scaler scales the partition's way
    if (f.pscaler) mode = c->sh.rate_scalers ? SCALE_RATE : SCALE_SITE;
  }
  // what record `r` says about the ops ahead of position `pos` (pos may be -2, -1: the headers)
  auto look_ahead = [&](FusedRec & r, long pos) -> int {
    r.chars = 0;
    r.req_lmat = r.req_rmat = 0;
    if (pos + 1 >= 0 && pos + 1 < (long)n) r.chars = chars_of[pos + 1];
    if (pos + 2 < (long)n)
    {
      const FusedOp & f = plan[pos + 2];
      // (the rows of op + 2 are fetched once op + 1's characters have been taken from the registers)
      const unsigned int now = pos + 1 >= 0 ? batch_of[pos + 1] : batch0;
      if (batch_of[pos + 2] != now) r.chars |= PLLHIP_FUSED_CH_LOAD | batch_of[pos + 2] << 24;
      r.req_lmat = (unsigned int)((f.lmat - c->pmatrix) * sizeof(double));
      r.req_rmat = (unsigned int)((f.rmat - c->pmatrix) * sizeof(double));
    }
    r.src = 0;
    if (pos + 1 >= 0 && pos + 1 < (long)n)
    {
      const FusedOp & f = plan[pos + 1];
      r.gather_off = table_of[pos + 1];
      r.flags |= (f.kind == 0 ? 2u : f.kind == 1 ? 1u : 0u) << PLLHIP_FUSED_STAGE_SHIFT;
      if (f.dma_flags)
      {
        if (srcs.size() >= 0xffffu) return 1;
        r.flags |= PLLHIP_FUSED_RELOAD_NEXT;
        r.src = (unsigned short)srcs.size();
        FusedSrc s;
        memset(&s, 0, sizeof(s));
        if (f.dma_flags & 1) { s.left_hbm = f.left_hbm; s.lsc_hbm = f.lsc_hbm; }
        if (f.dma_flags & 2) { s.right_hbm = f.right_hbm; s.rsc_hbm = f.rsc_hbm; }
        s.where = (unsigned long long)((f.lslot > 0 ? f.lslot : 0) * slot_bytes) |
                  (unsigned long long)((f.rslot > 0 ? f.rslot : 0) * slot_bytes) << 16 |
                  (unsigned long long)((f.lslot > 0 ? f.lslot : 0) * count_bytes) << 32 |
                  (unsigned long long)((f.rslot > 0 ? f.rslot : 0) * count_bytes) << 48;
        srcs.push_back(s);
      }
    }
    return 0;
  };
  const size_t first = recs.size();
  segs[sg] = FusedSeg{(unsigned int)first, n, batch0, 0u};
  recs.resize(first + n + 3);
  FusedRec * rs = recs.data() + first;
  memset(rs, 0, (n + 3) * sizeof(FusedRec));
  if (look_ahead(rs[0], -2) || look_ahead(rs[1], -1)) return 1;
  for (unsigned int pos = 0; pos < n; ++pos)
  {
    const FusedOp & f = plan[pos];
    FusedRec & r = rs[pos + 2];
    r.flags = (unsigned int)f.kind & PLLHIP_FUSED_KIND_MASK;
    if (f.pslot >= 0) r.flags |= PLLHIP_FUSED_HAS_PSLOT;
    if (f.pscaler) r.flags |= PLLHIP_FUSED_SCALING;
    if (f.kind == 0 && f.lsc_slot >= 0) r.flags |= PLLHIP_FUSED_LCNT;
    if (f.kind != 2 && f.rsc_slot >= 0) r.flags |= PLLHIP_FUSED_RCNT;
    r.parent = (unsigned long long)(uintptr_t)f.parent;
    r.pscaler = (unsigned long long)(uintptr_t)f.pscaler;
    r.lslot_b = (unsigned short)((f.lslot > 0 ? f.lslot : 0) * slot_bytes);
    r.rslot_b = (unsigned short)((f.rslot > 0 ? f.rslot : 0) * slot_bytes);
    r.pslot_b = (unsigned short)((f.pslot > 0 ? f.pslot : 0) * slot_bytes);
    r.lcnt_b = (unsigned short)((f.lsc_slot > 0 ? f.lsc_slot : 0) * count_bytes);
    r.rcnt_b = (unsigned short)((f.rsc_slot > 0 ? f.rsc_slot : 0) * count_bytes);
    r.pcnt_b = (unsigned short)((f.pslot > 0 ? f.pslot : 0) * count_bytes);
    r.list_pos = (unsigned short)f.list_pos;
    if (look_ahead(r, (long)pos)) return 1;
  }
  rs[n + 2] = rs[n + 1]; // (loaded by the last op, never used)
  rs[n + 2].flags &= ~PLLHIP_FUSED_RELOAD_NEXT;
  } // segments

  const size_t rec_bytes = recs.size() * sizeof(FusedRec);
  const size_t seg_bytes = (size_t)PLLHIP_FUSED_MAX_SEGS * sizeof(FusedSeg);
  const size_t src_bytes = (srcs.size() + 1) * sizeof(FusedSrc);
  const size_t job_bytes = (jobs.size() + 1) * sizeof(FusedPairJob);
  const size_t tab_bytes = rowtab.size() * sizeof(unsigned long long);
  const size_t bytes = rec_bytes + seg_bytes + src_bytes + job_bytes + tab_bytes;
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
  char * stage = static_cast<char *>(c->h_plan[b]);
  memcpy(stage, recs.data(), rec_bytes);
  memcpy(stage + rec_bytes, segs.data(), segs.size() * sizeof(FusedSeg));
  if (!srcs.empty()) memcpy(stage + rec_bytes + seg_bytes, srcs.data(), srcs.size() * sizeof(FusedSrc));
  if (!jobs.empty()) memcpy(stage + rec_bytes + seg_bytes + src_bytes, jobs.data(), jobs.size() * sizeof(FusedPairJob));
  memcpy(stage + rec_bytes + seg_bytes + src_bytes + job_bytes, rowtab.data(), tab_bytes);
  HIP_TRY(hipMemcpyAsync(c->d_plan, c->h_plan[b], bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipEventRecord(c->plan_done[b], c->stream));
  c->plan_pending[b] = true;
  if (!c->d_sink) HIP_TRY(hipMalloc(&c->d_sink, (size_t)c->num_cus * 16 * 96 * sizeof(double2))); // 1536 B per wave
  if (!c->d_tile_counter)
  {
    HIP_TRY(hipMalloc((void **)&c->d_tile_counter, 2 * PLLHIP_TILE_COUNTER_BYTES));
    HIP_TRY(hipMemsetAsync(c->d_tile_counter, 0, 2 * PLLHIP_TILE_COUNTER_BYTES, c->stream));
  }
  // what a repeated call with the same op list needs (pllhip_relaunch_fused)
  c->fused_last_segs_offset = rec_bytes;
  c->fused_last_srcs_offset = rec_bytes + seg_bytes;
  c->fused_last_jobs_offset = rec_bytes + seg_bytes + src_bytes;
  c->fused_last_rowtab_offset = rec_bytes + seg_bytes + src_bytes + job_bytes;
  c->fused_last_jobs = (unsigned int)jobs.size();
  c->fused_last_count = plans[0].size();
  c->fused_last_longest = longest;
  c->fused_last_nsegs = nsegs;
  c->fused_last_nslots = nslots;
  c->fused_last_mode = mode;
  c->fused_last_epoch = c->layout_epoch;
  return pllhip_relaunch_fused(c);
}

// The device copy of the plan is still that of the previous call (same op list: the plan
// holds buffer addresses, not values -- P-matrices, tip characters and CLVs are read when the
// kernels run): tip tables and the list kernel again, no planning, no upload.
int pllhip_relaunch_fused(pllhip_ctx * c)
{
  const unsigned int count = c->fused_last_count, nslots = c->fused_last_nslots, njobs = c->fused_last_jobs; // (count: segment 0's)
  const int mode = c->fused_last_mode;
  static_assert(PLLHIP_TILE_COUNTER_BYTES == 256 * 128, "one counter per thread of k_dna_pair_tables' first workgroup");
  const FusedRec * d_plan = (const FusedRec *)c->d_plan;
  const FusedPairJob * d_jobs = (const FusedPairJob *)(static_cast<const char *>(c->d_plan) + c->fused_last_jobs_offset);
  const FusedBases bases = {c->pmatrix, c->d_pairtab,
                            (const unsigned long long *)(static_cast<const char *>(c->d_plan) + c->fused_last_rowtab_offset),
                            (const FusedSrc *)(static_cast<const char *>(c->d_plan) + c->fused_last_srcs_offset),
                            (const FusedSeg *)(static_cast<const char *>(c->d_plan) + c->fused_last_segs_offset),
                            c->fused_last_nsegs};
  if (njobs)
  {
    switch (c->sh.rate_cats)
    {
      case 1: k_dna_pair_tables<1><<<4 * njobs, 256, 0, c->stream>>>(d_jobs, njobs, nullptr); break;
      case 2: k_dna_pair_tables<2><<<4 * njobs, 256, 0, c->stream>>>(d_jobs, njobs, nullptr); break;
      case 8: k_dna_pair_tables<8><<<4 * njobs, 256, 0, c->stream>>>(d_jobs, njobs, nullptr); break;
      default: k_dna_pair_tables<4><<<4 * njobs, 256, 0, c->stream>>>(d_jobs, njobs, nullptr); break;
    }
    HIP_TRY(hipGetLastError());
  }
  switch (c->sh.rate_cats)
  {
    case 1: return launch_fused_rc<1>(c, d_plan, bases, count, nslots, mode);
    case 2: return launch_fused_rc<2>(c, d_plan, bases, count, nslots, mode);
    case 8: return launch_fused_rc<8>(c, d_plan, bases, count, nslots, mode);
    default: return launch_fused_rc<4>(c, d_plan, bases, count, nslots, mode);
  }
}
