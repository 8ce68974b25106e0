// partials_aa_mfma.hip -- 20-state inner-inner CLV update on the f64 matrix cores.
//
// Replaces pll_core_update_partial_ii for 20 states (AVX2-flag kernel
// core_partials_avx2.c:568).  Per (site, rate) the update is two 20x20 . 20
// mat-vecs and an element-wise product; over many sites that is the dense
// contraction  X[20 x N] = P_k[20 x 20] . CLV_k[20 x N]  with N = sites.
//
// Instruction choice (measured on MI355X, tools/mfma_f64_bench.hip):
//   v_mfma_f64_16x16x4_f64     ~136 cycles/instr -> 33-43 TFLOP/s chip-wide, and a
//                              20-row P needs 2 row tiles (37.5 % of the work is padding)
//   v_mfma_f64_4x4x4_4b_f64    ~17 cycles/instr  -> 65 TFLOP/s, and 20 = 5 x 4: no padding
// so the kernel uses the 4-block 4x4x4 form.  Its lane layout (probed with
// tools/mfma_layout_probe.hip; blk = (lane>>2)&3):
//   A: lane holds A_blk[i = lane&3][k = lane>>4]
//   B: lane holds B_blk[k = lane>>4][j = lane&3]
//   D: lane holds D_blk[i = lane>>4][j = lane&3]
// With block blk = sites 4blk..4blk+3 of a 16-site tile, B and D are indexed by
// (site s = lane&15, q = lane>>4): B = state 4c+q of site s, D = state 4g+q of
// site s.  A = P[4g + (lane&3)][4c + q], the same for all four blocks.
// 5 row groups x 5 k-chunks = 25 MFMAs per (child, rate); 200 per 16-site tile.
//
// Tiling.  A wave owns 16 consecutive sites x all RC categories: ONE contiguous
// block of 16*RC*160 bytes per child in HBM.  It is copied by LDS-DMA
// (global_load_lds_dwordx4, no VGPR staging) into a per-wave LDS image whose 16
// site rows are padded by one 16-byte granule (pad lanes load a dummy granule),
// so the B-operand reads of a wave hit 64 different banks.  Left child first;
// once its operands are in registers the image is refilled with the right child
// while the left products run, and as soon as the right operands are in registers
// with the NEXT tile's left child (the schedule is spelled out inside the kernel).
// Both children produce the same accumulator layout, so x*y and the "< 2^-256" test
// are register-local; the per-site AND over the 4 lanes of a column is one
// __ballot.  The product tile returns through the same image, one tile late, so
// the parent CLV is written 16 contiguous bytes per lane.  P-matrices sit in LDS
// ([child][rate][20][20], bank-conflict-free for the A pattern); A operands are
// fetched five at a time right before their MFMAs.  Two waves per SIMD (LDS-bound:
// two workgroups of four images per CU).  GATHER variants follow site-repeat row
// maps (host/repeats.c): a tile's 16 site rows are then addressed one by one.
//
// Roofline.  1932 B and 6320 flop per site-update: 3.3 flop/B, below
// the f64 machine balance, so HBM is the bound: 200 MFMAs x 17 cycles per 64
// elements per SIMD is ~25 % of the time HBM needs for the tile's 31 KB.
//
// Numerics.  Every CLV and scaler count of these kernels is the reference's bit for bit:
//   inner-inner  the four FMA chains strided by j mod 4 and the pairwise tree of core_partials_avx2.c:632-750, which
//                v_mfma_f64_4x4x4 reproduces step for step when the contraction is chunked as {m, m+4, m+8, m+12} and
//                the fifth step follows (round 3; aa_mfma.hpp: rate_matvec_chain);
//   tip-inner    the reference's kernel for these ops, also under the AVX2 flag, multiplies and adds separately
//                (core_partials.c:427-441 -> core_partials_avx.c:1097-1340): the inner child's mat-vec runs on the
//                vector unit in that order (round 4; aa_mfma.hpp: rate_matvec_plain), the tip's factor is a table row;
//   tip-tip      one multiplication of two table rows.
// PLLHIP_AA_EXACT=1 selects the all-vector kernels of partials_gen_tile.hip instead (k_gen_wide in the AVX2-flag
// order): same CLVs, and per-site lnL bit for bit too (the lnL / sumtable kernels of this path sum a row in one chain).
#include "ctx.hpp"
#include "numerics.hpp"

#include <type_traits>

#include "aa_mfma.hpp"

// KIND 0: inner-inner.  KIND 1: tip-inner -- the left factor is not a mat-vec but
// a row of the precomputed tip table (a.ltab, [code][rate][state]), which takes
// the place of the left P-matrix in LDS; the right child goes through the MFMAs.
//
// SPLIT (round 4): 20-state data with a number of rate categories other than 1, 2 or 4 -- 8 (Gamma with eight
// categories), 3, 5, 6, 7, 10 ... (free-rate models).  A tile of 16 sites x 8 categories does not fit the LDS budget (two
// images per wave, 20 KB each; 51 KB of matrices) and other counts have no tile geometry at all, so an op runs as
// SEVERAL launches of this kernel, each over a CHUNK of 4, 2 or 1 categories of every site (R = 8: 4 + 4; 6: 4 + 2; 7:
// 4 + 2 + 1; 3: 2 + 1): the site stride in HBM is that of all categories (a.rate_cats), the chunk starts at category
// a.rate_first, matrices and tip tables are the chunk's.  SPLIT = 1: the first chunk, 3: a middle one, 2: the last.
// Per-rate scalers need nothing else.  With per-site scalers a site scales iff ALL categories are small: every chunk
// but the last stores its products unscaled and leaves / ANDs its verdict per site in the parent's scale buffer (1 =
// all small so far), the last reads it, decides, scales its own chunk in registers and -- rarely -- the earlier ones in
// place.  Same arithmetic per entry as the reference's kernels (core_partials_avx2.c:568-803,
// core_partials_avx.c:1097-1340), same scaling rule: same bits.  Until round 4 such partitions ran on the all-vector
// kernels at a third of this rate.
template <int RC, int MODE, bool NT, int KIND, bool GATHER, int SPLIT = 0>
__global__ __launch_bounds__(256, 2) void k_aa_ii_mfma(PartialsBatch batch)
{
  static_assert(SPLIT == 0 || !GATHER, "chunks of the categories do not follow row maps");
  const PartialsArgs & a = batch.op[blockIdx.y];
  const unsigned int RT = SPLIT ? a.rate_cats : (unsigned int)RC; // categories of the CLV
  const unsigned int RF = SPLIT ? a.rate_first : 0u;              // first category of this launch
  using G = aa_geom<RC>;
  extern __shared__ double smem[];
  // LDS: [left part][right P-matrices RC x 20 x 20][4 wave images]
  //   left part = left P-matrices (KIND 0) or the tip table [maxstates][RC][20] (KIND 1)
  const unsigned int left_elems = (KIND == 0) ? RC * 400u : a.maxstates * RC * 20u;
  double * ptab = smem;
  double * ptab_r = smem + left_elems;
  for (unsigned int t = threadIdx.x; t < left_elems; t += blockDim.x)
    ptab[t] = (KIND == 0) ? a.lmat[t] : a.ltab[t];
  for (unsigned int t = threadIdx.x; t < RC * 400u; t += blockDim.x) ptab_r[t] = a.rmat[t];
  __syncthreads();

  const unsigned int lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned int s = lane & 15u, q = lane >> 4;
  // the lane's A-operand addresses: into the left matrices (inner-inner) and the right ones
  const char * la14, * la5, * ra14, * ra5;
  chain_lane_bases(ptab, lane, la14, la5);
  chain_lane_bases(ptab_r, lane, ra14, ra5);
  // (tip-inner: the inner child's products on the vector unit in the reference's non-fused order, aa_mfma.hpp)
  const char * const rrow = reinterpret_cast<const char *>(ptab_r) + q * 160u;
  char * region = reinterpret_cast<char *>(ptab_r + RC * 400) + wave * G::REGION_B;
  constexpr int ROW_B = G::ROW_G * 16;

  unsigned int toff[G::N_IT];
  tile_offsets<RC>(lane, toff, RT, RF);
  unsigned int store_mask = 0; // bit it: granule it*64+lane of the image is data (not pad, not past the tile)
#pragma unroll
  for (int it = 0; it < G::N_IT; ++it)
  {
    const int P = it * 64 + (int)lane;
    if (P < G::TILE_G && (P % G::ROW_G) < G::ROW_G - 1) store_mask |= 1u << it;
  }
  const size_t sites = a.sites;
  const size_t tiles = (sites + 15) / 16;
  const unsigned int wpb = blockDim.x >> 6; // waves per workgroup
  const size_t nwaves = (size_t)gridDim.x * wpb;
  const unsigned int * ls = a.lscaler ? a.lscaler : a.zero;
  const unsigned int * rs = a.rscaler ? a.rscaler : a.zero;
  const bool has_l = a.lscaler != nullptr, has_r = a.rscaler != nullptr;

  // One tile = load left, load right, multiply, scale, store.  Its stores are
  // issued one iteration late -- after the NEXT tile's left operands have been
  // read out of the image and before that tile's right child is requested -- so
  // that (a) the next left child is already on its way while this tile is
  // multiplied, and (b) a wave never sits waiting for its own write
  // acknowledgements with nothing else in flight (vmcnt counts loads and stores
  // in one queue).  Measured on the 200 k-site op: 72 -> see DESIGN.md 2.2.
  const size_t first = (size_t)blockIdx.x * wpb + wave;
  if (first >= tiles) return; // (no barrier follows)
  double x[RC][5];            // products of the tile whose stores are pending
  unsigned int psc[RC];       // its parent scaler count(s), lanes q == 0
  size_t prev_site0 = 0;
  bool have_prev = false;

  // write the pending tile: transpose through the (currently free) image,
  // 16 bytes per lane to HBM
  auto flush = [&]() {
#pragma unroll
    for (int k = 0; k < RC; ++k)
#pragma unroll
      for (int g = 0; g < 5; ++g)
        *reinterpret_cast<double *>(region + s * ROW_B + k * 160 + (4 * g + q) * 8) = x[k][g];
    // all LDS reads first, into registers of their own: a store whose data
    // registers are reloaded for the next store makes the compiler wait for the
    // store to COMPLETE (vmcnt) before the reload -- 11 serialised round trips
    double2 v[G::N_IT];
#pragma unroll
    for (int it = 0; it < G::N_IT; ++it)
    {
      int P = it * 64 + (int)lane;
      if (P > G::TILE_G - 1) P = G::TILE_G - 1;
      v[it] = *reinterpret_cast<const double2 *>(region + P * 16);
    }
    // destination = (uniform) start of the tile in the parent CLV + the same per-lane
    // offsets the DMA uses; pad lanes and sites past the end do not store
    const unsigned long long ob = (unsigned long long)(a.parent + prev_site0 * (size_t)(RT * 20u));
    const unsigned int olo = __builtin_amdgcn_readfirstlane((unsigned int)ob);
    const unsigned int ohi = __builtin_amdgcn_readfirstlane((unsigned int)(ob >> 32));
    char * obase = reinterpret_cast<char *>(((unsigned long long)ohi << 32) | olo);
    const unsigned int left_sites = (unsigned int)(sites - prev_site0 < 16 ? sites - prev_site0 : 16);
#pragma unroll
    for (int it = 0; it < G::N_IT; ++it)
    {
      const unsigned int site = SPLIT ? (unsigned int)((it * 64 + (int)lane) / G::ROW_G) : toff[it] / (unsigned int)(RC * 160);
      if (((store_mask >> it) & 1u) && site < left_sites)
        st16<NT>(reinterpret_cast<double2 *>(obase + toff[it]), v[it].x, v[it].y);
    }
    const size_t n = prev_site0 + s;
    if (MODE == SCALE_SITE && q == 0 && n < sites)
    {
      if (SPLIT == 1 || SPLIT == 3) const_cast<unsigned int *>(a.lidx)[n] = psc[0]; // (the verdict buffer, see the scaling step)
      else a.pscaler[n] = psc[0];
    }
    if (MODE == SCALE_RATE && q == 0 && n < sites)
#pragma unroll
      for (int k = 0; k < RC; ++k) a.pscaler[n * RT + RF + k] = psc[k];
    // the image is about to be refilled by a DMA: its reads must have left LDS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

  // Operand order: the FIRST operand tile of a tile is the left child (inner-inner) or the
  // inner child (tip-inner), the SECOND the right child (inner-inner only).
  //
  // Per-site words are requested ahead of the DMA behind which they would otherwise queue
  // (loads return in order): the first operand's scaler count and the tip code with the
  // request of that operand -- one tile ahead --, the second operand's scaler count at the
  // top of its tile, before the second operand itself.
  //
  // Site repeats (GATHER): a parent row takes each child at the row a.lidx / a.ridx name
  // (nullptr = same index; for a tip the entry IS the tip code).  The row of the first
  // operand is needed when that operand is requested, so it is fetched two tiles ahead;
  // the row of the second operand one tile ahead.
  const double * const first_clv = (KIND == 0) ? a.left : a.right;
  const unsigned int * const first_idx = (KIND == 0) ? a.lidx : a.ridx;
  const unsigned int * const first_sc = (KIND == 0) ? ls : rs;
  const bool has_first = (KIND == 0) ? has_l : has_r;
  unsigned int fsc_next[RC], code_next = 0; // first operand's scaler count(s), tip code: next tile
  unsigned int frow_next = 0, frow_next2 = 0, srow_next = 0; // rows: first operand (t+1, t+2), second (t+1)
  auto row_of = [&](const unsigned int * idx, size_t tile) -> unsigned int {
    const size_t n = tile * 16 + s;
    return (GATHER && idx) ? idx[n] : (unsigned int)n; // (lists carry zeroed slack)
  };
  auto request_first = [&](size_t tile, unsigned int frow) {
    // scaler count(s) of the first operand's row, tip code of the parent row, then the tile
    const size_t n = tile * 16 + s;
    if (KIND == 1) code_next = (GATHER && a.lidx) ? a.lidx[n] : a.ltip[n < sites ? n : 0];
#pragma unroll
    for (int k = 0; k < RC; ++k)
    {
      const size_t e = (MODE == SCALE_RATE) ? (size_t)frow * RT + RF + k : (size_t)frow;
      const bool used = (MODE == SCALE_RATE) || (MODE == SCALE_SITE && k == 0);
      fsc_next[k] = used ? first_sc[has_first ? e : 0] : 0u;
    }
    if (GATHER) dma_tile_rows<RC, NT>(first_clv, frow, toff, region);
    else dma_tile<RC, NT>(first_clv, tile * 16, toff, region, RT);
  };
  {
    // prologue: the rows of the first tile have to be here before anything can be requested
    unsigned int frow0 = row_of(first_idx, first);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(frow0));
    srow_next = (KIND == 0) ? row_of(a.ridx, first) : 0u;
    frow_next2 = row_of(first_idx, first + nwaves < tiles ? first + nwaves : first);
    request_first(first, frow0);
  }
  for (size_t tile = first; tile < tiles; tile += nwaves)
  {
    const size_t site0 = tile * 16;
    const size_t next = tile + nwaves;
    double b[RC][5], xl[RC][5];

    // ---- first operand tile of this iteration has been requested earlier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    read_b_chain<RC>(region, s, q, b);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // what came in with it is taken out of the registers the next requests overwrite
    unsigned int csc[RC], code = code_next, srow = srow_next;
    frow_next = frow_next2;
    asm volatile("" : "+v"(code), "+v"(srow), "+v"(frow_next)); // consumed HERE, nothing in flight
#pragma unroll
    for (int k = 0; k < RC; ++k)
    {
      csc[k] = fsc_next[k];
      asm volatile("" : "+v"(csc[k]));
    }
    if (have_prev) flush();
    if (KIND == 0)
    {
      // second operand: its scaler count(s) first, then the tile
      unsigned int ssc[RC];
#pragma unroll
      for (int k = 0; k < RC; ++k)
      {
        const size_t e = (MODE == SCALE_RATE) ? (size_t)srow * RT + RF + k : (size_t)srow;
        const bool used = (MODE == SCALE_RATE) || (MODE == SCALE_SITE && k == 0);
        ssc[k] = used ? rs[has_r ? e : 0] : 0u;
      }
      if (GATHER) dma_tile_rows<RC, NT>(a.right, srow, toff, region);
      else dma_tile<RC, NT>(a.right, site0, toff, region, RT);
      tile_matvec_chain<RC, 0>(la14, la5, b, lane, xl); // overlaps the right child's DMA and the stores
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < RC; ++k)
      {
        csc[k] += ssc[k];
        asm volatile("" : "+v"(csc[k]));
      }
      read_b_chain<RC>(region, s, q, b);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // ---- next tile's first operand: in flight while this tile is finished; behind it the
    // rows that will be needed one and two tiles from now
    if (next < tiles)
    {
      request_first(next, frow_next);
      if (GATHER)
      {
        if (KIND == 0) srow_next = row_of(a.ridx, next);
        frow_next2 = row_of(first_idx, next + nwaves < tiles ? next + nwaves : next);
      }
      else
      {
        srow_next = (unsigned int)(next * 16 + s);
        // (round 3: this line was missing -- from a wave's THIRD tile on the first operand's scaler
        // counts were read at the second tile's sites.  Invisible until sites, taxa and scaling
        // events came together: 100,000 sites x 200 taxa; found by the whole-list kernel disagreeing)
        frow_next2 = (unsigned int)((next + nwaves < tiles ? next + nwaves : next) * 16 + s);
      }
    }
    if (KIND == 1)
    {
      // left factor: tip row sums of this column's site, states 4g+q
      if (code >= a.maxstates) code = 0;
#pragma unroll
      for (int k = 0; k < RC; ++k)
#pragma unroll
        for (int g = 0; g < 5; ++g) xl[k][g] = ptab[(code * RC + k) * 20 + 4 * g + q];
    }
    // ---- right products, product + scaling (core_partials_avx2.c:752-800), a rate at a time
    bool small_site = true;
    bool small_rate[RC];
    auto right_rate = [&](auto kc) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      if (k >= RC) return;
      constexpr int kk = k < RC ? k : 0;
      double yk[5];
      if (KIND == 1) rate_matvec_plain<kk * 3200>(rrow, b[kk], lane, yk);
      else rate_matvec_chain<kk * 3200>(ra14, ra5, b[kk], lane, yk);
      small_rate[kk] = true;
#pragma unroll
      for (int g = 0; g < 5; ++g)
      {
        x[kk][g] = xl[kk][g] * yk[g];
        small_rate[kk] = small_rate[kk] && (x[kk][g] < PLLHIP_SCALE_THRESHOLD);
      }
      small_site = small_site && small_rate[kk];
    };
    right_rate(std::integral_constant<int, 0>{});
    right_rate(std::integral_constant<int, 1>{});
    right_rate(std::integral_constant<int, 2>{});
    right_rate(std::integral_constant<int, 3>{});
    // (round 6) the scaling certificate (ctx.hpp): this kernel works in the reference's order, but a child may carry
    // the matrix cores' rounding of a tip-inner op of the whole-list kernel; then a decision is the reference's for
    // certain only while no block's largest entry lies within rounding distance of the threshold.  Two integer
    // instructions on the high word per category (a window of 2^-20); the exact window only behind that.
#ifndef PLLHIP_NO_II_CERT /* (tool build: what the test costs a launch that does not need it) */
    if (MODE != SCALE_NONE && SPLIT == 0 && a.cert)
    {
      bool wide = false, inside = false;
#pragma unroll
      for (int k = 0; k < RC; ++k)
      {
        const double mx = fmax(fmax(fmax(x[k][0], x[k][1]), fmax(x[k][2], x[k][3])), x[k][4]);
        const bool w = ((unsigned int)__double2hiint(mx) - 0x2FEFFFFFu) < 2u;
        wide = wide || w;
        inside = inside || (w && fabs(mx - PLLHIP_SCALE_THRESHOLD) < __hiloint2double((int)((767u - a.pad_) << 20), 0));
      }
      if (__ballot(wide))
        if (__ballot(inside && site0 + s < sites) && lane == 0u) *a.cert = 1u;
    }
#endif
    if (MODE == SCALE_SITE && (SPLIT == 1 || SPLIT == 3))
    {
      // not the last chunk of the categories: no decision yet -- the verdict so far goes to the verdict buffer (the
      // parent's scale buffer, a.lidx; a scratch array when that buffer is also a child's: the op works in place)
      const size_t n = site0 + s;
      const unsigned int before = (SPLIT == 3 && n < sites) ? a.lidx[n] : 1u;
      psc[0] = (column_all(small_site, s) && before != 0u) ? 1u : 0u;
    }
    else if (MODE == SCALE_SITE)
    {
      bool scale = column_all(small_site, s);
      if (SPLIT == 2)
      {
        // last chunk: the site scales iff the earlier chunks said "all small" too; their entries (written unscaled by
        // those launches) are then scaled in place -- rare: a wave-uniform mask of such sites, 10 granules per category
        const size_t n = site0 + s;
        const unsigned int earlier = (n < sites) ? a.lidx[n] : 0u;
        scale = scale && earlier != 0u;
        const unsigned long long fix = __ballot(scale && q == 0 && n < sites);
        if (fix)
        {
          double2 * prow = reinterpret_cast<double2 *>(a.parent + site0 * (size_t)(RT * 20u));
          for (unsigned int sl = 0; sl < 16u; ++sl)
            if ((fix >> sl) & 1ull)
              for (unsigned int g = lane; g < RF * 10u; g += 64u)
              {
                double2 v = prow[(size_t)sl * (RT * 10u) + g];
                v.x *= PLLHIP_SCALE_FACTOR;
                v.y *= PLLHIP_SCALE_FACTOR;
                prow[(size_t)sl * (RT * 10u) + g] = v;
              }
        }
      }
      if (scale)
#pragma unroll
        for (int k = 0; k < RC; ++k)
#pragma unroll
          for (int g = 0; g < 5; ++g) x[k][g] *= PLLHIP_SCALE_FACTOR;
      psc[0] = csc[0] + (scale ? 1u : 0u);
    }
    if (MODE == SCALE_RATE)
    {
#pragma unroll
      for (int k = 0; k < RC; ++k)
      {
        const bool scale = column_all(small_rate[k], s);
        if (scale)
#pragma unroll
          for (int g = 0; g < 5; ++g) x[k][g] *= PLLHIP_SCALE_FACTOR;
        psc[k] = csc[k] + (scale ? 1u : 0u);
      }
    }
    prev_site0 = site0;
    have_prev = true;
  }
  // ---- the last tile: nothing is in flight into the image any more
  flush();
}

// ---- tip tables, built once per op: tab[code][k][i] = sum_{j in tipmap[code]} P[k][i][j]
// (core_partials_avx.c:1140-1177); left and right tables back to back
__global__ __launch_bounds__(256) void k_aa_tip_tables(PartialsBatch batch, double * __restrict__ tab,
                                                       unsigned int maxstates, unsigned int rate_cats,
                                                       int both)
{
  // blockIdx.y = op of the batch; its two tables sit back to back in `tab`
  const PartialsArgs & a = batch.op[blockIdx.y];
  const unsigned int per = maxstates * rate_cats * 20;
  const unsigned int total = both ? 2 * per : per;
  double * out = tab + (size_t)blockIdx.y * 2 * per;
  for (unsigned int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x)
  {
    const unsigned int side = t / per, u = t % per;
    const unsigned int code = u / (rate_cats * 20), ki = u % (rate_cats * 20);
    out[t] = masksum_seq((side ? a.rmat : a.lmat) + (size_t)ki * 20, a.tipmap[code], 20);
  }
}

// tip-tip: parent = ltab[code_l] (.) rtab[code_r].  One lane per 16 bytes, waves
// in rounds of 64 sites (codes fetched once per round, one site per lane), the
// GS = 10*RC store instructions of a round each one contiguous KiB.
// (CHUNK: the launch writes RC of the a.rate_cats categories of every site, from a.rate_first on -- see k_aa_ii_mfma's
// SPLIT; the scale buffer is cleared by the chunk that starts at category 0)
template <int RC, int MODE, bool NT, bool GATHER, bool CHUNK = false>
__global__ __launch_bounds__(256) void k_aa_tt_rounds(PartialsBatch batch)
{
  static_assert(!CHUNK || !GATHER, "chunks of the categories do not follow row maps");
  const PartialsArgs & a = batch.op[blockIdx.y];
  constexpr unsigned int GS = RC * 10; // 16-byte granules per site (of this launch)
  const unsigned int GT = CHUNK ? a.rate_cats * 10u : GS, G0 = CHUNK ? a.rate_first * 10u : 0u;
  extern __shared__ double smem[];
  const unsigned int per = a.maxstates * RC * 20;
  double * tl = smem, * tr = smem + per;
  for (unsigned int t = threadIdx.x; t < per; t += blockDim.x) { tl[t] = a.ltab[t]; tr[t] = a.rtab[t]; }
  __syncthreads();
  const unsigned int lane = threadIdx.x & 63u;
  const size_t sites = a.sites;
  const size_t rounds = (sites + 63) / 64;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  double2 * __restrict__ out = reinterpret_cast<double2 *>(a.parent);
  for (size_t r = wave; r < rounds; r += nwaves)
  {
    const size_t site0 = r * 64;
    // (site repeats: a parent row names the two tip codes it is the product of)
    unsigned int cl = (GATHER && a.lidx) ? a.lidx[site0 + lane]
                                         : ((site0 + lane < sites) ? a.ltip[site0 + lane] : 0u);
    unsigned int cr = (GATHER && a.ridx) ? a.ridx[site0 + lane]
                                         : ((site0 + lane < sites) ? a.rtip[site0 + lane] : 0u);
    if (cl >= a.maxstates) cl = 0;
    if (cr >= a.maxstates) cr = 0;
    const size_t gbase = site0 * GS, gend = sites * GS;
#pragma unroll 10
    for (unsigned int j = 0; j < GS; ++j)
    {
      const unsigned int gg = j * 64 + lane;          // granule within the round
      const unsigned int sl = gg / GS, rr = gg - sl * GS; // site in round, granule in site
      const unsigned int c1 = (unsigned int)__shfl((int)cl, (int)sl, 64);
      const unsigned int c2 = (unsigned int)__shfl((int)cr, (int)sl, 64);
      const double2 x = *reinterpret_cast<const double2 *>(tl + c1 * RC * 20 + rr * 2);
      const double2 y = *reinterpret_cast<const double2 *>(tr + c2 * RC * 20 + rr * 2);
      // plain stores: a write-only stream is slower with the non-temporal hint (see partials.hip)
      if (gbase + gg < gend)
        st16<false>(CHUNK ? out + (site0 + sl) * GT + G0 + rr : out + gbase + gg, x.x * y.x, x.y * y.y);
    }
    // no scaling test on tip-tip; the scaler is cleared (core_partials_avx.c:552-553)
    if (CHUNK && a.rate_first != 0u) continue;
    if (MODE == SCALE_SITE && site0 + lane < sites) a.pscaler[site0 + lane] = 0u;
    if (MODE == SCALE_RATE)
    {
      const unsigned int RA = CHUNK ? a.rate_cats : (unsigned int)RC;
      for (unsigned int t = lane; t < 64 * RA; t += 64)
        if (site0 * RA + t < sites * RA) a.pscaler[site0 * RA + t] = 0u;
    }
  }
}

template <int RC, int KIND, int SPLIT = 0>
static int launch_rc(pllhip_ctx * c, const PartialsBatch & b, unsigned int count, int mode, bool nt)
{
  using G = aa_geom<RC>;
  const PartialsArgs & a = b.op[0];
  // (site repeats: the ops of a launch differ in their number of rows; the grid covers
  // the largest, a wave past its op's tiles returns at once)
  size_t rows_max = 0;
  bool gather = false;
  for (unsigned int i = 0; i < count; ++i)
  {
    if (b.op[i].sites > rows_max) rows_max = b.op[i].sites;
    gather = gather || (!SPLIT && (b.op[i].lidx || b.op[i].ridx)); // (SPLIT: lidx is the verdict buffer)
  }
  const size_t tiles = (rows_max + 15) / 16;
  // Two workgroups per CU share its 160 KB of LDS; each holds the P tables (or the tip
  // table) once and one 11 KB image per wave.  Four waves per workgroup: a fifth fits
  // for inner-inner ops (25.6 + 5 x 11 = 80 KB) and was measured -- 69.0 us per op
  // against 66.7 us: at 8 waves per CU the kernel is no longer short of bytes in flight.
  const size_t left_elems = (KIND == 0) ? (size_t)RC * 400 : (size_t)a.maxstates * RC * 20;
  const size_t fixed = (left_elems + (size_t)RC * 400) * sizeof(double);
  const unsigned int wpb = 4u;
  size_t blocks = (tiles + wpb - 1) / wpb;
  // the P-matrix staging per workgroup is amortised over several tiles per wave
  // (PLLHIP_AA_GRID_CAP: tests make every wave walk many tiles of a small partition)
  const size_t cap = pllhip_env("PLLHIP_AA_GRID_CAP") ? (size_t)atoi(pllhip_env("PLLHIP_AA_GRID_CAP")) : (size_t)c->num_cus * 2;
  if (blocks > cap) blocks = cap;
  const dim3 grid((unsigned int)blocks, count), block(64 * wpb);
  const size_t lds = fixed + wpb * (size_t)G::REGION_B;
  if (lds > 80 * 1024) return 1; // two workgroups per CU must fit
  // more than 64 KB of dynamic LDS has to be requested per kernel
#define AA_LAUNCH_ONE(KERNEL)                                                                 \
  do {                                                                                        \
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&KERNEL),                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));       \
    hipLaunchKernelGGL(KERNEL, grid, block, lds, c->stream, b);                               \
  } while (0)
#define AA_LAUNCH(MODEV)                                                                      \
  do {                                                                                        \
    if (SPLIT) {                                                                              \
      if (gather) return 1; /* (chunks of the categories do not follow row maps) */           \
      if (nt) AA_LAUNCH_ONE((k_aa_ii_mfma<RC, MODEV, true, KIND, false, SPLIT>));             \
      else AA_LAUNCH_ONE((k_aa_ii_mfma<RC, MODEV, false, KIND, false, SPLIT>));               \
    }                                                                                         \
    else if (gather) AA_LAUNCH_ONE((k_aa_ii_mfma<RC, MODEV, false, KIND, true, 0>));          \
    else if (nt) AA_LAUNCH_ONE((k_aa_ii_mfma<RC, MODEV, true, KIND, false, 0>));              \
    else AA_LAUNCH_ONE((k_aa_ii_mfma<RC, MODEV, false, KIND, false, 0>));                     \
  } while (0)
  if (mode == SCALE_NONE) AA_LAUNCH(0);
  else if (mode == SCALE_SITE) AA_LAUNCH(1);
  else AA_LAUNCH(2);
#undef AA_LAUNCH
#undef AA_LAUNCH_ONE
  HIP_TRY(hipGetLastError());
  return 0;
}

template <int RC, bool CHUNK = false>
static int launch_tt(pllhip_ctx * c, const PartialsBatch & b, unsigned int count, int mode, bool nt)
{
  const PartialsArgs & a = b.op[0];
  size_t rows_max = 0;
  bool gather = false;
  for (unsigned int i = 0; i < count; ++i)
  {
    if (b.op[i].sites > rows_max) rows_max = b.op[i].sites;
    gather = gather || (!CHUNK && (b.op[i].lidx || b.op[i].ridx));
  }
  const size_t rounds = (rows_max + 63) / 64;
  size_t blocks = (rounds + 3) / 4;
  const size_t cap = (size_t)c->num_cus * 8; // 29 KB of table staging per workgroup
  if (blocks > cap) blocks = cap;
  const size_t lds = 2 * (size_t)a.maxstates * RC * 20 * sizeof(double);
  const dim3 grid((unsigned int)blocks, count), block(256);
#define TT_LAUNCH(MODEV)                                                                         \
  do {                                                                                           \
    if (CHUNK) hipLaunchKernelGGL((k_aa_tt_rounds<RC, MODEV, false, false, true>), grid, block, lds, c->stream, b); \
    else if (gather) hipLaunchKernelGGL((k_aa_tt_rounds<RC, MODEV, false, true>), grid, block, lds, c->stream, b); \
    else if (nt) hipLaunchKernelGGL((k_aa_tt_rounds<RC, MODEV, true, false>), grid, block, lds, c->stream, b); \
    else hipLaunchKernelGGL((k_aa_tt_rounds<RC, MODEV, false, false>), grid, block, lds, c->stream, b); \
  } while (0)
  if (mode == SCALE_NONE) TT_LAUNCH(0);
  else if (mode == SCALE_SITE) TT_LAUNCH(1);
  else TT_LAUNCH(2);
#undef TT_LAUNCH
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---- inner-inner ops whose two children are tip-tip results of the same list
//
// Such a child is, per site, one of maxstates^2 vectors (one per pair of tip characters),
// so "P x child" -- the factor the inner-inner kernel forms with 25 MFMAs per rate -- is
// one of maxstates^2 vectors too.  They are tabulated with the SAME kernels that would have
// produced them site by site (a tip-tip launch over all character pairs, then an
// inner-inner launch whose other factor is exactly 1: identity matrix times a vector of
// ones), so the entries are bit-identical to what k_aa_ii_mfma computes, and the op itself
// becomes parent = TL[pair 1] (.) TR[pair 2] with the usual scaling test: a write stream
// of 8 x states x rate_cats bytes per site instead of three times that (k_aa_cherry_rounds;
// the tables, 338 KB each for 23 characters, are read through L2).
struct CherryArgs
{
  const double * tl, * tr;                 // [pair][rate][state]
  const unsigned char * t1, * t2, * t3, * t4; // characters of the four tips
  double * parent;
  unsigned int * pscaler;
  unsigned int sites, maxstates;
};
struct CherryBatch
{
  CherryArgs op[PLLHIP_BATCH_MAX];
};

template <int RC, int MODE, bool NT> // MODE: SCALE_NONE or SCALE_SITE
__global__ __launch_bounds__(256) void k_aa_cherry_rounds(CherryBatch batch)
{
  const CherryArgs & a = batch.op[blockIdx.y];
  constexpr unsigned int GS = RC * 10; // 16-byte granules per site
  const unsigned int lane = threadIdx.x & 63u;
  const size_t sites = a.sites;
  const size_t rounds = (sites + 63) / 64;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  double2 * __restrict__ out = reinterpret_cast<double2 *>(a.parent);
  const double2 * __restrict__ tl = reinterpret_cast<const double2 *>(a.tl);
  const double2 * __restrict__ tr = reinterpret_cast<const double2 *>(a.tr);
  const unsigned int ms = a.maxstates;
  for (size_t r = wave; r < rounds; r += nwaves)
  {
    const size_t site0 = r * 64;
    const bool mine = site0 + lane < sites;
    unsigned int c1 = mine ? a.t1[site0 + lane] : 0u, c2 = mine ? a.t2[site0 + lane] : 0u;
    unsigned int c3 = mine ? a.t3[site0 + lane] : 0u, c4 = mine ? a.t4[site0 + lane] : 0u;
    if (c1 >= ms) c1 = 0;
    if (c2 >= ms) c2 = 0;
    if (c3 >= ms) c3 = 0;
    if (c4 >= ms) c4 = 0;
    const unsigned int pl = c1 * ms + c2, pr = c3 * ms + c4;
    const size_t gbase = site0 * GS, gend = sites * GS;
    unsigned long long all_small = ~0ull; // bit s: every entry of site s of the round below the threshold
#pragma unroll 10
    for (unsigned int j = 0; j < GS; ++j)
    {
      const unsigned int gg = j * 64 + lane;              // granule within the round
      const unsigned int sl = gg / GS, rr = gg - sl * GS; // site in round, granule in site
      const unsigned int p1 = (unsigned int)__shfl((int)pl, (int)sl, 64);
      const unsigned int p2 = (unsigned int)__shfl((int)pr, (int)sl, 64);
      const double2 x = tl[(size_t)p1 * GS + rr];
      const double2 y = tr[(size_t)p2 * GS + rr];
      const double v0 = x.x * y.x, v1 = x.y * y.y;
      const bool in = gbase + gg < gend;
      if (in) st16<false>(out + gbase + gg, v0, v1);
      if (MODE == SCALE_SITE)
      {
        // the 64 granules of this step belong to at most three consecutive sites, s0..s1
        const bool small = (v0 < PLLHIP_SCALE_THRESHOLD) & (v1 < PLLHIP_SCALE_THRESHOLD);
        const unsigned int s0 = (j * 64u) / GS, s1 = (j * 64u + 63u) / GS;
        if (!__ballot(small && in))
          // the common case, one ballot: nothing here is small, none of these sites scales
          all_small &= ~(((2ull << (s1 - s0)) - 1ull) << s0);
        else
        {
#pragma unroll
          for (unsigned int t = 0; t < 3; ++t)
            if (__ballot(in && !small && sl == s0 + t)) all_small &= ~(1ull << (s0 + t));
        }
      }
    }
    if (MODE == SCALE_SITE)
    {
      // rare: the sites whose entries are all small are multiplied by 2^256 in place
      if (all_small)
        for (unsigned int j = 0; j < GS; ++j)
        {
          const unsigned int gg = j * 64 + lane;
          const unsigned int sl = gg / GS;
          if (gbase + gg < gend && ((all_small >> sl) & 1ull))
          {
            double2 v = out[gbase + gg];
            v.x *= PLLHIP_SCALE_FACTOR;
            v.y *= PLLHIP_SCALE_FACTOR;
            out[gbase + gg] = v;
          }
        }
      // both children are tip-tip results: nothing to inherit
      if (mine) a.pscaler[site0 + lane] = (unsigned int)((all_small >> lane) & 1ull);
    }
  }
}

bool pllhip_aa_chunks_enabled()
{
  static const bool on = !(pllhip_env("PLLHIP_AA_CHUNKS") && atoi(pllhip_env("PLLHIP_AA_CHUNKS")) == 0) &&
                         !(pllhip_env("PLLHIP_AA_RC8") && atoi(pllhip_env("PLLHIP_AA_RC8")) == 0);
  return on;
}

bool pllhip_aa_fast_covers(const pllhip_ctx * c, int kind)
{
  const unsigned int R = c->sh.rate_cats;
  // (category counts other than 1, 2, 4 -- round 4: several launches per op, each over a chunk of 4, 2 or 1 of the
  // categories, k_aa_ii_mfma's SPLIT; PLLHIP_AA_CHUNKS=0: the all-vector kernels as before)
  if (c->aa_exact || c->sh.states != 20 || R == 0) return false;
  const bool whole = R == 1 || R == 2 || R == 4;
  if (!whole && !pllhip_aa_chunks_enabled()) return false;
  if (kind == 0) return true;
  // tip kinds: both tables of an op (of a chunk) must fit the workgroup's LDS next to its other data
  return c->maxstates > 0 && c->maxstates <= 32 &&
         2 * (size_t)c->maxstates * (R < 4 ? R : 4) * 20 * sizeof(double) <= 60 * 1024;
}

// Category counts other than 1, 2, 4: every op of the batch chunk by chunk (k_aa_ii_mfma's SPLIT, k_aa_tt_rounds'
// CHUNK).  Each launch sees the matrices -- and, for a tip child, a tip table -- of its categories; the per-site
// verdict of the chunks so far travels in the parent's scale buffer unless that buffer is also a child's (an op that
// scales in place).
template <int RC, int KIND>
static int launch_chunk(pllhip_ctx * c, const PartialsBatch & h, unsigned int count, int mode, bool nt, int split)
{
  switch (split)
  {
    case 1: return launch_rc<RC, KIND, 1>(c, h, count, mode, nt);
    case 3: return launch_rc<RC, KIND, 3>(c, h, count, mode, nt);
    default: return launch_rc<RC, KIND, 2>(c, h, count, mode, nt);
  }
}

static int launch_chunks(pllhip_ctx * c, const PartialsBatch & b, unsigned int count, int kind, int mode, bool nt)
{
  const unsigned int R = b.op[0].rate_cats, ms = b.op[0].maxstates;
  if (kind >= 1)
  {
    // tip tables of all chunks side by side: [chunk][op][left, right][code][category of the chunk][state]
    const size_t need = 2 * (size_t)ms * R * 20 * PLLHIP_BATCH_MAX;
    if (c->tiptab_elems < need)
    {
      HIP_TRY(hipStreamSynchronize(c->stream));
      if (c->d_tiptab) HIP_TRY(hipFree(c->d_tiptab));
      c->d_tiptab = nullptr;
      HIP_TRY(hipMalloc((void **)&c->d_tiptab, need * sizeof(double)));
      ++c->layout_epoch;
      c->tiptab_elems = need;
    }
  }
  for (unsigned int rf = 0; rf < R;)
  {
    const unsigned int left = R - rf, rc = left >= 4 ? 4u : (left >= 2 ? 2u : 1u);
    const int split = rf == 0 ? 1 : (rf + rc == R ? 2 : 3);
    PartialsBatch h = b;
    for (unsigned int i = 0; i < count; ++i)
    {
      PartialsArgs & a = h.op[i];
      if (a.lmat) a.lmat += (size_t)rf * 400;
      if (a.rmat) a.rmat += (size_t)rf * 400;
      a.rate_first = rf;
      a.lidx = a.pscaler;
      a.ridx = nullptr;
      if (kind != 2 && mode == SCALE_SITE && a.pscaler && (a.pscaler == a.lscaler || a.pscaler == a.rscaler))
      {
        // (one array per op of the launch that needs one, grown on demand: such ops are rare)
        const size_t per_op = (size_t)c->sh.sites + PLLHIP_TAIL_SITES;
        if (c->split_verdicts_ops < count)
        {
          HIP_TRY(hipStreamSynchronize(c->stream));
          if (c->split_verdicts) HIP_TRY(hipFree(c->split_verdicts));
          c->split_verdicts = nullptr;
          c->split_verdicts_ops = 0;
          HIP_TRY(hipMalloc((void **)&c->split_verdicts, (size_t)count * per_op * sizeof(unsigned int)));
          c->split_verdicts_ops = count;
          // (earlier chunks of THIS op list have not run yet with the old array: rf == 0 is where an op first gets here)
        }
        a.lidx = c->split_verdicts + (size_t)i * per_op;
      }
    }
    if (kind >= 1)
    {
      // the tips' row sums over this chunk's categories (k_aa_tip_tables reads the pre-offset matrices)
      const size_t per = (size_t)ms * rc * 20;
      double * tab = c->d_tiptab + 2 * (size_t)ms * rf * 20 * PLLHIP_BATCH_MAX;
      k_aa_tip_tables<<<dim3(8, count), 256, 0, c->stream>>>(h, tab, ms, rc, kind == 2 ? 1 : 0);
      HIP_TRY(hipGetLastError());
      for (unsigned int i = 0; i < count; ++i)
      {
        h.op[i].ltab = tab + (size_t)i * 2 * per;
        h.op[i].rtab = h.op[i].ltab + per;
      }
    }
    int rcode;
    if (kind == 2)
      rcode = rc == 4 ? launch_tt<4, true>(c, h, count, mode, nt)
                      : (rc == 2 ? launch_tt<2, true>(c, h, count, mode, nt) : launch_tt<1, true>(c, h, count, mode, nt));
    else if (kind == 1)
      rcode = rc == 4 ? launch_chunk<4, 1>(c, h, count, mode, nt, split)
                      : (rc == 2 ? launch_chunk<2, 1>(c, h, count, mode, nt, split) : launch_chunk<1, 1>(c, h, count, mode, nt, split));
    else
      rcode = rc == 4 ? launch_chunk<4, 0>(c, h, count, mode, nt, split)
                      : (rc == 2 ? launch_chunk<2, 0>(c, h, count, mode, nt, split) : launch_chunk<1, 0>(c, h, count, mode, nt, split));
    if (rcode) return rcode;
    rf += rc;
  }
  return 0;
}

int pllhip_launch_aa_batch(pllhip_ctx * c, PartialsBatch & b, unsigned int count, int kind, int mode)
{
  const unsigned int R = b.op[0].rate_cats;
  const bool nt = pllhip_use_nt(c);
  if (!(R == 1 || R == 2 || R == 4)) return launch_chunks(c, b, count, kind, mode, nt);
  if (kind >= 1)
  {
    // tip row-sum tables of every op of the batch, one launch
    const size_t per = (size_t)b.op[0].maxstates * R * 20;
    const size_t need = 2 * per * PLLHIP_BATCH_MAX;
    if (c->tiptab_elems < need)
    {
      HIP_TRY(hipStreamSynchronize(c->stream));
      if (c->d_tiptab) HIP_TRY(hipFree(c->d_tiptab));
      c->d_tiptab = nullptr;
      HIP_TRY(hipMalloc((void **)&c->d_tiptab, need * sizeof(double)));
      ++c->layout_epoch;
      c->tiptab_elems = need;
    }
    k_aa_tip_tables<<<dim3(8, count), 256, 0, c->stream>>>(b, c->d_tiptab, b.op[0].maxstates, R,
                                                           kind == 2 ? 1 : 0);
    HIP_TRY(hipGetLastError());
    for (unsigned int i = 0; i < count; ++i)
    {
      b.op[i].ltab = c->d_tiptab + (size_t)i * 2 * per;
      b.op[i].rtab = b.op[i].ltab + per;
    }
  }
  if (kind == 2)
  {
    switch (R)
    {
      case 1: return launch_tt<1>(c, b, count, mode, nt);
      case 2: return launch_tt<2>(c, b, count, mode, nt);
      default: return launch_tt<4>(c, b, count, mode, nt);
    }
  }
  if (kind == 1)
  {
    switch (R)
    {
      case 1: return launch_rc<1, 1>(c, b, count, mode, nt);
      case 2: return launch_rc<2, 1>(c, b, count, mode, nt);
      default: return launch_rc<4, 1>(c, b, count, mode, nt);
    }
  }
  switch (R)
  {
    case 1: return launch_rc<1, 0>(c, b, count, mode, nt);
    case 2: return launch_rc<2, 0>(c, b, count, mode, nt);
    default: return launch_rc<4, 0>(c, b, count, mode, nt);
  }
}

// tip-inner ops whose tip tables (PartialsArgs::ltab) the caller has made itself; no scaling
static int launch_ti_with_tables(pllhip_ctx * c, const PartialsBatch & b, unsigned int count)
{
  const bool nt = pllhip_use_nt(c);
  switch (b.op[0].rate_cats)
  {
    case 1: return launch_rc<1, 1>(c, b, count, SCALE_NONE, nt);
    case 2: return launch_rc<2, 1>(c, b, count, SCALE_NONE, nt);
    default: return launch_rc<4, 1>(c, b, count, SCALE_NONE, nt);
  }
}

// ---- host side of the lookup ops (see k_aa_cherry_rounds)
//
// ops[i]: the inner-inner op; kid1[i] / kid2[i]: the tip-tip ops that produced its children
// (characters and matrices are taken from them).  mode: SCALE_NONE or SCALE_SITE.
bool pllhip_aa_cherry_covers(const pllhip_ctx * c, int mode)
{
  const char * e = pllhip_env("PLLHIP_AA_CHERRY"); // 0: never, 2: whatever the partition's size (tests)
  const bool off = e && atoi(e) == 0;
  return !off && !c->cherry_pool_failed && c->sh.states == 20 && (c->sh.rate_cats == 1 || c->sh.rate_cats == 2 || c->sh.rate_cats == 4) && pllhip_aa_fast_covers(c, 0) && pllhip_aa_fast_covers(c, 2) &&
         c->rows.empty() && mode != SCALE_RATE && c->maxstates >= 1 && c->maxstates <= 32 && !c->sh.asc_states;
}

// Do `lookups` lookup ops on `levels` tree levels pay for their tables?  (partials.hip: the model)
bool pllhip_aa_cherry_pays(const pllhip_ctx * c, unsigned int lookups, unsigned int levels)
{
  const char * e = pllhip_env("PLLHIP_AA_CHERRY");
  if (e && atoi(e) == 2) return true;
  const double saved_us = (double)lookups * c->sh.sites * (1932.0 - 646.0) / 5.5e6; // bytes / (bytes per us)
  const double cost_us = 8.0 * 6.0 * levels;
  return saved_us >= 2.0 * cost_us;
}

static __global__ void k_aa_cherry_consts(unsigned char * hi, unsigned char * lo, double * ones, double * ident,
                                          unsigned int ms, unsigned int rows, unsigned int RC)
{
  const unsigned int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < rows)
  {
    hi[t] = (unsigned char)(t < ms * ms ? t / ms : 0);
    lo[t] = (unsigned char)(t < ms * ms ? t % ms : 0);
    lo[rows + t] = 0; // (a row of zero characters as long as the pair list)
  }
  if (t < rows * RC * 20) ones[t] = 1.0;
  if (t < RC * 400) ident[t] = ((t % 400) / 20 == t % 20) ? 1.0 : 0.0;
}

// tip-inner lookup ops: the tip's factor as "pair" table rows (code, 0): the row sums of
// k_aa_tip_tables, same expression
struct TipMats
{
  const double * m[PLLHIP_BATCH_MAX]; // nullptr: not a tip-inner lookup op
};

static __global__ void k_aa_tiprow_tables(CherryBatch batch, TipMats lmats,
                                          const unsigned int * __restrict__ tipmap, unsigned int ms,
                                          unsigned int RC)
{
  const CherryArgs & a = batch.op[blockIdx.y];
  const double * lmat = lmats.m[blockIdx.y];
  if (!lmat) return;
  double * tl = const_cast<double *>(a.tl);
  const unsigned int per = ms * RC * 20;
  for (unsigned int t = blockIdx.x * blockDim.x + threadIdx.x; t < per; t += gridDim.x * blockDim.x)
  {
    const unsigned int code = t / (RC * 20), ki = t % (RC * 20);
    tl[((size_t)code * ms) * RC * 20 + ki] = masksum_seq(lmat + (size_t)ki * 20, tipmap[code], 20);
  }
}

// scratch shared by every lookup-table build of a context: the characters of all pairs, a CLV
// of ones, identity matrices, a row of zero characters
static int cherry_scratch(pllhip_ctx * c, size_t rows, size_t row_elems, unsigned int chunk)
{
  const unsigned int R = c->sh.rate_cats, ms = c->maxstates;
  if (c->cherry_pool && c->cherry_ms != ms)
  {
    // (the character map was replaced by one with another number of codes)
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipFree(c->cherry_pool));
    HIP_TRY(hipFree(c->cherry_codes));
    if (c->cherry_pool_all) HIP_TRY(hipFree(c->cherry_pool_all));
    c->cherry_pool = nullptr;
    c->cherry_codes = nullptr;
    c->cherry_pool_all = nullptr;
    c->cherry_pool_all_ops = 0;
    ++c->layout_epoch;
  }
  if (!c->cherry_pool)
  {
    c->cherry_ms = ms;
    // per lookup op of a chunk: pair CLVs of the two children, TL, TR; then the constants
    const size_t per_op = 4 * rows * row_elems;
    HIP_TRY(hipMalloc((void **)&c->cherry_pool, (chunk * per_op + rows * row_elems + (size_t)R * 400) * sizeof(double)));
    HIP_TRY(hipMalloc((void **)&c->cherry_codes, 3 * rows));
    ++c->layout_epoch;
    if (!c->cherry_zero)
    {
      HIP_TRY(hipMalloc((void **)&c->cherry_zero, (size_t)c->sh.sites + PLLHIP_TAIL_SITES));
      HIP_TRY(hipMemsetAsync(c->cherry_zero, 0, (size_t)c->sh.sites + PLLHIP_TAIL_SITES, c->stream));
    }
    double * ones = c->cherry_pool + chunk * per_op;
    k_aa_cherry_consts<<<(unsigned int)((rows * row_elems + 255) / 256), 256, 0, c->stream>>>(
        c->cherry_codes, c->cherry_codes + rows, ones, ones + rows * row_elems, ms, (unsigned int)rows, R);
    HIP_TRY(hipGetLastError());
  }
  return 0;
}

// The tables of n <= PLLHIP_BATCH_MAX / 2 lookup ops (ops / kid1 / kid2 [0..n)), built in `pool`
// (n x 4 x rows x row_elems doubles); ch.op[i] describes op i for k_aa_cherry_rounds.
static int cherry_tables(pllhip_ctx * c, double * pool, const PartialsArgs * ops, const PartialsArgs * kid1,
                         const PartialsArgs * kid2, unsigned int n, CherryBatch & ch)
{
  const unsigned int R = c->sh.rate_cats, ms = c->maxstates;
  const size_t pairs = (size_t)ms * ms;
  const size_t rows = pairs + PLLHIP_TAIL_SITES; // the kernels load whole tiles
  const size_t row_elems = (size_t)R * 20;
  const unsigned int chunk = PLLHIP_BATCH_MAX / 2;
  const size_t per_op = 4 * rows * row_elems;
  double * ones = c->cherry_pool + chunk * per_op;
  double * ident = ones + rows * row_elems;
  PartialsBatch tt, ii, ti;
  unsigned int ntab = 0, nii = 0, nti = 0;
  TipMats h_lmats;
  memset(&h_lmats, 0, sizeof(h_lmats));
  bool any_tip_left = false;
  for (unsigned int i = 0; i < n; ++i)
  {
    const PartialsArgs & op = ops[i];
    double * base = pool + i * per_op;
    double * pair_clv[2] = {base, base + rows * row_elems};
    double * table[2] = {base + 2 * rows * row_elems, base + 3 * rows * row_elems};
    const PartialsArgs * kid[2] = {&kid1[i], &kid2[i]};
    // tip-inner lookup op (kid 1 is a dummy): the left factor is the tip's own table
    const bool tip_left = kid[0]->lmat == nullptr;
    h_lmats.m[i] = tip_left ? op.lmat : nullptr;
    for (int s = 0; s < 2; ++s)
    {
      if (s == 0 && tip_left) continue;
      // the child over all character pairs, by the tip-tip kernel and the child op's matrices
      PartialsArgs & t = tt.op[ntab];
      memset(&t, 0, sizeof(t));
      t.parent = pair_clv[s];
      t.ltip = c->cherry_codes;
      t.rtip = c->cherry_codes + rows;
      t.lmat = kid[s]->lmat;
      t.rmat = kid[s]->rmat;
      t.tipmap = c->tipmap;
      t.zero = c->d_zero;
      t.sites = (unsigned int)pairs;
      t.rate_cats = R;
      t.states = 20;
      t.maxstates = ms;
      ++ntab;
      // P x child by the kernel the op itself would have run, the other factor being exactly 1:
      //   inner-inner op: the inner-inner kernel (fused chains, core_partials_avx2.c:632-750), identity x ones;
      //   tip-inner op:   the tip-inner kernel (products and sums rounded separately, core_partials_avx.c:1229-1284),
      //                   a tip table of ones and a row of zero characters
      PartialsArgs & u = tip_left ? ti.op[nti++] : ii.op[nii++];
      memset(&u, 0, sizeof(u));
      u.parent = table[s];
      if (tip_left)
      {
        u.right = pair_clv[s];
        u.rmat = op.rmat;
        u.ltab = ones; // [code][rate][state], every entry 1.0
        u.ltip = c->cherry_codes + 2 * rows;
      }
      else
      {
        u.left = pair_clv[s];
        u.right = ones;
        u.lmat = s == 0 ? op.lmat : op.rmat;
        u.rmat = ident;
      }
      u.tipmap = c->tipmap;
      u.zero = c->d_zero;
      u.sites = (unsigned int)pairs;
      u.rate_cats = R;
      u.states = 20;
      u.maxstates = ms;
    }
    CherryArgs & k = ch.op[i];
    k.tl = table[0];
    k.tr = table[1];
    k.t1 = tip_left ? op.ltip : kid[0]->ltip;
    k.t2 = tip_left ? c->cherry_zero : kid[0]->rtip;
    k.t3 = kid[1]->ltip;
    k.t4 = kid[1]->rtip;
    k.parent = op.parent;
    k.pscaler = op.pscaler;
    k.sites = op.sites;
    k.maxstates = ms;
  }
  int rc = pllhip_launch_aa_batch(c, tt, ntab, 2, SCALE_NONE);
  if (rc) return rc;
  if (nii && (rc = pllhip_launch_aa_batch(c, ii, nii, 0, SCALE_NONE))) return rc;
  if (nti && (rc = launch_ti_with_tables(c, ti, nti))) return rc;
  for (unsigned int i = 0; i < n; ++i) any_tip_left = any_tip_left || h_lmats.m[i];
  if (any_tip_left)
  {
    k_aa_tiprow_tables<<<dim3(4, n), 256, 0, c->stream>>>(ch, h_lmats, c->tipmap, ms, R);
    HIP_TRY(hipGetLastError());
  }
  return 0;
}

int pllhip_launch_aa_cherries(pllhip_ctx * c, const PartialsArgs * ops, const PartialsArgs * kid1,
                              const PartialsArgs * kid2, unsigned int count, int mode)
{
  const unsigned int R = c->sh.rate_cats, ms = c->maxstates;
  const size_t rows = (size_t)ms * ms + PLLHIP_TAIL_SITES;
  const size_t row_elems = (size_t)R * 20;
  const unsigned int chunk = PLLHIP_BATCH_MAX / 2; // two table ops per lookup op and launch
  int rc = cherry_scratch(c, rows, row_elems, chunk);
  if (rc) return rc;
  for (unsigned int first = 0; first < count; first += chunk)
  {
    const unsigned int n = (count - first < chunk) ? count - first : chunk;
    CherryBatch ch;
    rc = cherry_tables(c, c->cherry_pool, ops + first, kid1 + first, kid2 + first, n, ch);
    if (rc) return rc;
    const size_t rounds = ((size_t)c->sh.sites + 63) / 64;
    size_t blocks = (rounds + 3) / 4;
    const size_t cap = (size_t)c->num_cus * 8;
    if (blocks > cap) blocks = cap;
    const dim3 grid((unsigned int)blocks, n);
#define CHERRY_LAUNCH(RCV)                                                                                  \
  do {                                                                                                      \
    if (mode == SCALE_SITE) k_aa_cherry_rounds<RCV, SCALE_SITE, false><<<grid, 256, 0, c->stream>>>(ch);   \
    else k_aa_cherry_rounds<RCV, SCALE_NONE, false><<<grid, 256, 0, c->stream>>>(ch);                      \
  } while (0)
    switch (R)
    {
      case 1: CHERRY_LAUNCH(1); break;
      case 2: CHERRY_LAUNCH(2); break;
      default: CHERRY_LAUNCH(4); break;
    }
#undef CHERRY_LAUNCH
    HIP_TRY(hipGetLastError());
  }
  return 0;
}

// How many lookup ops a list may have: their tables (4 x (maxstates^2 + 64) x rate_cats x 20 doubles each, 1.5 MB for
// the protein alphabet) live in one pool next to the CLVs; a tree with thousands of cherries would take gigabytes
// (ADVICE r3).  1 GiB by default (PLLHIP_AA_LOOKUP_MB), never more than 1/16 of the CLV arena's size + 64 MB; ops
// beyond that stay ordinary inner-inner / tip-inner ops.
unsigned int pllhip_aa_lookup_budget(const pllhip_ctx * c)
{
  const size_t rows = (size_t)c->maxstates * c->maxstates + PLLHIP_TAIL_SITES;
  const size_t per_op = 4 * rows * (size_t)c->sh.rate_cats * 20 * sizeof(double);
  size_t budget = (size_t)1 << 30;
  if (const char * e = pllhip_env("PLLHIP_AA_LOOKUP_MB"))
    if (atoi(e) >= 0) budget = (size_t)atoi(e) << 20;
  const size_t share = c->clv_arena_bytes / 16 + ((size_t)64 << 20);
  if (budget > share) budget = share;
  return (unsigned int)(budget / (per_op ? per_op : 1));
}

// The tables of ALL `count` lookup ops of a list at once (the whole-list kernel,
// partials_aa_fused.hip, walks a tile of sites through every op): same builders, a pool of
// their own that grows with the list.
int pllhip_aa_lookup_tables(pllhip_ctx * c, const PartialsArgs * ops, const PartialsArgs * kid1,
                            const PartialsArgs * kid2, unsigned int count, AaLookupTables * out, AaLookupJob * jobs)
{
  const unsigned int R = c->sh.rate_cats, ms = c->maxstates;
  const size_t rows = (size_t)ms * ms + PLLHIP_TAIL_SITES;
  const size_t row_elems = (size_t)R * 20;
  const size_t per_op = 4 * rows * row_elems;
  const unsigned int chunk = PLLHIP_BATCH_MAX / 2;
  int rc = cherry_scratch(c, rows, row_elems, chunk);
  if (rc) return rc;
  if (c->cherry_pool_all_ops < count)
  {
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->cherry_pool_all) HIP_TRY(hipFree(c->cherry_pool_all));
    c->cherry_pool_all = nullptr;
    c->cherry_pool_all_ops = 0;
    const size_t n = ((size_t)count + 15) & ~(size_t)15;
    if (hipMalloc((void **)&c->cherry_pool_all, n * per_op * sizeof(double)) != hipSuccess)
    {
      // Not an error of the call: the list runs without lookup ops (the caller plans it again; return value 1 =
      // "not taken").  Remembered, so that later calls do not repeat the synchronisation, the free and the failing
      // allocation.
      (void)hipGetLastError();
      c->cherry_pool_all = nullptr;
      c->cherry_pool_failed = true;
      return 1;
    }
    c->cherry_pool_all_ops = (unsigned int)n;
    ++c->layout_epoch;
  }
  if (jobs)
  {
    // places only (the same ones cherry_tables uses); the caller's prepare kernel fills them
    for (unsigned int i = 0; i < count; ++i)
    {
      double * base = c->cherry_pool_all + (size_t)i * per_op;
      double * t0 = base + 2 * rows * row_elems, * t1 = base + 3 * rows * row_elems;
      const PartialsArgs & op = ops[i];
      const bool tip_left = kid1[i].lmat == nullptr; // a tip-inner lookup op: the left factor is the tip's own table
      out[i] = AaLookupTables{t0, t1, tip_left ? op.ltip : kid1[i].ltip, tip_left ? c->cherry_zero : kid1[i].rtip,
                              kid2[i].ltip, kid2[i].rtip};
      jobs[2 * i] = tip_left ? AaLookupJob{nullptr, op.lmat, nullptr, t0, 2u, 0u}
                             : AaLookupJob{op.lmat, kid1[i].lmat, kid1[i].rmat, t0, 0u, 0u};
      jobs[2 * i + 1] = AaLookupJob{op.rmat, kid2[i].lmat, kid2[i].rmat, t1, tip_left ? 1u : 0u, 0u};
    }
    return 0;
  }
  for (unsigned int first = 0; first < count; first += chunk)
  {
    const unsigned int n = (count - first < chunk) ? count - first : chunk;
    CherryBatch ch;
    rc = cherry_tables(c, c->cherry_pool_all + (size_t)first * per_op, ops + first, kid1 + first, kid2 + first, n, ch);
    if (rc) return rc;
    for (unsigned int i = 0; i < n; ++i)
      out[first + i] = AaLookupTables{ch.op[i].tl, ch.op[i].tr, ch.op[i].t1, ch.op[i].t2, ch.op[i].t3, ch.op[i].t4};
  }
  return 0;
}
