// partials_gen_tile.hip -- CLV updates for state counts without a dedicated kernel
// (2 = binary, 3..19 = morphological / DNA+gap, 21..64 = e.g. 61 codons).
//
// Reference: pll_core_update_partial_ii / _ti / _tt, the plain C kernels
// (core_partials.c:510-662, :354-507, :82-199) -- for these state counts the sums are
// formed left to right, and so they are here, one multiply and one add per term.
//
// Mapping.  A workgroup takes TILES of TS consecutive sites.  Both child tiles are
// copied to LDS with fully coalesced loads (a tile is one contiguous TS x rate_cats x
// states block of the CLV), the P-matrices sit in LDS TRANSPOSED ([j][i]), and one lane
// forms one output entry (site s, rate k, state i):
//     x = sum_j Pl_k[i][j] * L[s][k][j],   y likewise,   out = x * y
// Lanes of a wave differ in i fastest: the P reads of one instruction are consecutive
// LDS words (conflict-free for any state count) and the CLV reads are broadcasts of a
// few addresses.  The products go to an LDS image of the parent tile; once all rate
// categories of the tile are there the scaling decision (per site or per rate, from
// flags raised by entries >= 2^-256) is known, and the tile leaves for HBM as one
// contiguous, coalesced, possibly rescaled block together with its scaler counts.
// When all 2 x rate_cats matrices fit in 32 KB they stay resident for the whole kernel;
// otherwise (61 states: 59.5 KB per pair) the pair of the current category is reloaded
// per tile from L2.
//
// Roofline: 3 x 8 x rate_cats x states B/site of HBM traffic against
// rate_cats x states x (4 states + 1) flop: HBM-bound up to ~12 states, then bound by
// the two LDS reads per multiply-add (see DESIGN.md 2.1 for the measured rates).
#include "ctx.hpp"
#include "numerics.hpp"

struct GenTileGeom
{
  unsigned int ts;       // sites per tile
  unsigned int resident; // all matrices of the op stay in LDS
  unsigned int inv_s;    // ceil(2^32 / states): e / states == umulhi(e, inv_s) for e < 2^26
  unsigned int inv_span; // the same for states * rate_cats
};

template <int KIND> // 0 = inner-inner, 1 = tip-inner, 2 = tip-tip
__global__ __launch_bounds__(256) void k_gen_tile(PartialsBatch batch, int mode, GenTileGeom g)
{
  const PartialsArgs & a = batch.op[blockIdx.y];
  extern __shared__ double smem[];
  const unsigned int S = a.states, R = a.rate_cats, span = S * R, SS = S * S;
  const unsigned int TS = g.ts, nmat = g.resident ? R : 1u;
  const unsigned int tid = threadIdx.x;

  double * s_pl = smem;                                // [nmat][j][i]
  double * s_pr = s_pl + (size_t)nmat * SS;
  double * s_l = s_pr + (size_t)nmat * SS;             // [TS][R][S], inner child 1 (KIND 0)
  double * s_r = s_l + (KIND == 0 ? (size_t)TS * span : 0);
  double * s_out = s_r + (KIND != 2 ? (size_t)TS * span : 0);
  unsigned int * s_big = (unsigned int *)(s_out + (size_t)TS * span); // [TS][R]: some entry >= threshold
  unsigned int * s_lmask = s_big + TS * R;             // [TS] state masks of tip child 1 (KIND 1, 2)
  unsigned int * s_rmask = s_lmask + TS;               // [TS] of tip child 2 (KIND 2)

  auto load_matrices = [&](unsigned int slot, unsigned int k) {
    for (unsigned int t = tid; t < SS; t += 256u)
    {
      const unsigned int i = __umulhi(t, g.inv_s), j = t - i * S;
      s_pl[slot * SS + j * S + i] = a.lmat[(size_t)k * SS + t];
      s_pr[slot * SS + j * S + i] = a.rmat[(size_t)k * SS + t];
    }
  };
  if (g.resident)
    for (unsigned int k = 0; k < R; ++k) load_matrices(k, k);

  const size_t ntiles = ((size_t)a.sites + TS - 1) / TS;
  for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x)
  {
    const size_t n0 = tile * TS;
    const unsigned int ns = (a.sites - n0 < TS) ? (unsigned int)(a.sites - n0) : TS;
    const unsigned int cnt = ns * span;
    __syncthreads(); // the previous tile has left s_out / s_big
    if (KIND == 0)
      for (unsigned int e = tid; e < cnt; e += 256u) s_l[e] = a.left[n0 * span + e];
    if (KIND != 2)
      for (unsigned int e = tid; e < cnt; e += 256u) s_r[e] = a.right[n0 * span + e];
    if (KIND >= 1)
      for (unsigned int s = tid; s < ns; s += 256u) s_lmask[s] = a.tipmap[a.ltip[n0 + s]];
    if (KIND == 2)
      for (unsigned int s = tid; s < ns; s += 256u) s_rmask[s] = a.tipmap[a.rtip[n0 + s]];
    for (unsigned int f = tid; f < ns * R; f += 256u) s_big[f] = 0u;

    for (unsigned int k = 0; k < R; ++k)
    {
      if (!g.resident)
      {
        __syncthreads(); // category k-1 is done with the matrices
        load_matrices(0, k);
      }
      if (k == 0 || !g.resident) __syncthreads();
      const double * pl = s_pl + (g.resident ? k * SS : 0u);
      const double * pr = s_pr + (g.resident ? k * SS : 0u);
      for (unsigned int e = tid; e < ns * S; e += 256u)
      {
        const unsigned int s = __umulhi(e, g.inv_s), i = e - s * S;
        const unsigned int row = s * span + k * S;
        double x = 0.0, y = 0.0;
        const unsigned int lm = (KIND >= 1) ? s_lmask[s] : 0u, rm = (KIND == 2) ? s_rmask[s] : 0u;
        // both sums advance together, four terms per trip, so that eight (sixteen) LDS reads
        // are in flight instead of one
#pragma unroll 4
        for (unsigned int j = 0; j < S; ++j)
        {
          if (KIND == 0) x += pl[j * S + i] * s_l[row + j];
          else if ((lm >> j) & 1u) x += pl[j * S + i];
          if (KIND != 2) y += pr[j * S + i] * s_r[row + j];
          else if ((rm >> j) & 1u) y += pr[j * S + i];
        }
        const double p = x * y;
        s_out[row + i] = p;
        if (KIND != 2 && !(p < PLLHIP_SCALE_THRESHOLD)) s_big[s * R + k] = 1u;
      }
    }
    __syncthreads();

    // the tile leaves as one contiguous block (tip-tip updates never scale, as in the reference)
    for (unsigned int e = tid; e < cnt; e += 256u)
    {
      double p = s_out[e];
      if (KIND != 2 && mode != SCALE_NONE)
      {
        const unsigned int s = __umulhi(e, g.inv_span);
        bool big;
        if (mode == SCALE_RATE)
          big = s_big[s * R + __umulhi(e - s * span, g.inv_s)] != 0u;
        else
        {
          big = false;
          for (unsigned int k = 0; k < R; ++k) big = big || (s_big[s * R + k] != 0u);
        }
        if (!big) p *= PLLHIP_SCALE_FACTOR;
      }
      a.parent[n0 * span + e] = p;
    }
    if (mode == SCALE_SITE)
      for (unsigned int s = tid; s < ns; s += 256u)
      {
        unsigned int v = 0u;
        if (KIND != 2)
        {
          if (KIND == 0 && a.lscaler) v += a.lscaler[n0 + s];
          if (a.rscaler) v += a.rscaler[n0 + s];
          bool big = false;
          for (unsigned int k = 0; k < R; ++k) big = big || (s_big[s * R + k] != 0u);
          v += big ? 0u : 1u;
        }
        a.pscaler[n0 + s] = v;
      }
    if (mode == SCALE_RATE)
      for (unsigned int f = tid; f < ns * R; f += 256u)
      {
        unsigned int v = 0u;
        if (KIND != 2)
        {
          if (KIND == 0 && a.lscaler) v += a.lscaler[n0 * R + f];
          if (a.rscaler) v += a.rscaler[n0 * R + f];
          v += s_big[f] ? 0u : 1u;
        }
        a.pscaler[n0 * R + f] = v;
      }
  }
}

// ------------------------------------------------------------ few states: rows in registers
//
// Up to 16 states a (site, rate) row of a CLV is at most 128 bytes: one lane takes one row,
// keeps both child rows and the parent row in registers (state count known at compile
// time, loops unrolled) and walks the rows of P_l / P_r in LDS -- the lanes of a wave read
// rate_cats distinct LDS addresses per instruction, matrices padded by two words so that
// they fall in different banks.  Consecutive lanes own consecutive rows, so a wave's
// loads and stores cover one contiguous block of up to 64 x states x 8 B.  A wave holds
// 64 / rate_cats whole sites (any rate_cats up to 64; a few lanes idle when it is not a
// power of two), so the per-site scaling decision is one __ballot.  This is also where
// 4-state data with an unusual rate_cats (3, 5, 6, 10 ...) runs, in the pairwise order of
// the reference's 4x4 kernels.
template <int KIND, int SC>
__global__ __launch_bounds__(256) void k_gen_rows(PartialsBatch batch, int mode)
{
  const PartialsArgs & a = batch.op[blockIdx.y];
  extern __shared__ double smem[];
  const unsigned int R = a.rate_cats, tid = threadIdx.x;
  constexpr unsigned int MP = SC * SC + 2; // matrix stride in LDS
  double * s_pl = smem;
  double * s_pr = smem + R * MP;
  for (unsigned int t = tid; t < R * SC * SC; t += 256u)
  {
    const unsigned int k = t / (SC * SC), ij = t % (SC * SC);
    s_pl[k * MP + ij] = a.lmat[t];
    s_pr[k * MP + ij] = a.rmat[t];
  }
  __syncthreads();

  // a wave takes 64 / R whole sites per trip (all 64 lanes when R is a power of two):
  // lane = g * R + k is row k of the wave's g-th site, consecutive lanes own consecutive rows
  const unsigned int lane = tid & 63u;
  const unsigned int spw = 64u / R;
  const unsigned int g = lane / R, k = lane - g * R;
  const size_t trips = ((size_t)a.sites + spw - 1) / spw;
  for (size_t trip = (size_t)blockIdx.x * 4u + (tid >> 6); trip < trips; trip += (size_t)gridDim.x * 4u)
  {
    // (the lane's matrices are the same in every trip: without this the compiler hoists
    // all 2 x S x S LDS reads out of the loop and spills them)
    unsigned int koff = k * MP;
    const bool act = g < spw && trip * spw + g < a.sites;
    const size_t n = act ? trip * spw + g : 0;
    const size_t it = act ? n * R + k : 0;
    double l[SC], r[SC], out[SC];
    unsigned int lmask = 0u, rmask = 0u;
    if (KIND == 0)
      for (int j = 0; j < SC; ++j) l[j] = a.left[it * SC + j];
    else
      lmask = (SC == 4) ? a.ltip[n] : a.tipmap[a.ltip[n]]; // 4 states: the tip code is the mask
    if (KIND != 2)
      for (int j = 0; j < SC; ++j) r[j] = a.right[it * SC + j];
    else
      rmask = (SC == 4) ? a.rtip[n] : a.tipmap[a.rtip[n]];
    unsigned int inherited = 0u;
    if (KIND != 2 && mode != SCALE_NONE)
    {
      const size_t w = (mode == SCALE_RATE) ? it : n;
      if (KIND == 0 && a.lscaler) inherited += a.lscaler[w];
      if (a.rscaler) inherited += a.rscaler[w];
    }

    bool small = true;
#pragma unroll
    for (int i = 0; i < SC; ++i)
    {
      asm volatile("" : "+v"(koff)); // per matrix row: keeps one row's reads in flight, not S x S
      const double * pl = s_pl + koff;
      const double * pr = s_pr + koff;
      double x = 0.0, y = 0.0;
      if (SC == 4)
      {
        // 4 states keep the pairwise order of the reference's 4x4 kernels (numerics.hpp)
        x = (KIND == 0) ? dot4(pl + i * 4, l[0], l[1], l[2], l[3]) : masksum4(pl + i * 4, lmask);
        y = (KIND != 2) ? dot4(pr + i * 4, r[0], r[1], r[2], r[3]) : masksum4(pr + i * 4, rmask);
      }
      else
      {
#pragma unroll
        for (int j = 0; j < SC; ++j)
        {
          if (KIND == 0) x += pl[i * SC + j] * l[j];
          else if ((lmask >> j) & 1u) x += pl[i * SC + j];
          if (KIND != 2) y += pr[i * SC + j] * r[j];
          else if ((rmask >> j) & 1u) y += pr[i * SC + j];
        }
      }
      out[i] = x * y;
      small = small && (out[i] < PLLHIP_SCALE_THRESHOLD);
    }

    if (KIND != 2 && mode != SCALE_NONE)
    {
      bool scale = small;
      if (mode == SCALE_SITE)
      {
        // all rate_cats lanes of the site must agree
        const unsigned long long b = __ballot(small || !act);
        const unsigned long long grp = b >> (g * R);
        const unsigned long long full = (R >= 64u) ? ~0ull : ((1ull << R) - 1ull);
        scale = (grp & full) == full;
      }
      if (scale)
#pragma unroll
        for (int i = 0; i < SC; ++i) out[i] *= PLLHIP_SCALE_FACTOR;
      if (act && (mode == SCALE_RATE || k == 0u))
        a.pscaler[mode == SCALE_RATE ? it : n] = inherited + (scale ? 1u : 0u);
    }
    else if (mode != SCALE_NONE && act && (mode == SCALE_RATE || k == 0u))
      a.pscaler[mode == SCALE_RATE ? it : n] = 0u;
    if (act)
#pragma unroll
      for (int i = 0; i < SC; ++i) a.parent[it * SC + i] = out[i];
  }
}

// 4 states come here only with a rate_cats the dedicated kernels of partials.hip do not cover
static bool gen_rows_covers(unsigned int S, unsigned int R)
{
  return R >= 1 && R <= 64 && S >= 2 && S <= 16 && 2 * (size_t)R * (S * S + 2) * sizeof(double) <= 65536;
}

template <int KIND, int SC>
static int launch_gen_rows_sc(pllhip_ctx * c, const PartialsBatch & b, unsigned int count, int mode)
{
  const unsigned int R = c->sh.rate_cats;
  size_t sites = 0;
  for (unsigned int i = 0; i < count; ++i)
    if (b.op[i].sites > sites) sites = b.op[i].sites;
  if (!sites) return 0;
  // a workgroup's four waves take 64 / R sites each per trip
  const size_t trips = (sites + 64 / R - 1) / (64 / R);
  const dim3 grid(pllhip_stream_grid(c, trips * 64, 256), count);
  const size_t lds = 2 * (size_t)R * (SC * SC + 2) * sizeof(double);
  k_gen_rows<KIND, SC><<<grid, 256, lds, c->stream>>>(b, mode);
  HIP_TRY(hipGetLastError());
  return 0;
}

template <int KIND>
static int launch_gen_rows(pllhip_ctx * c, const PartialsBatch & b, unsigned int count, int mode)
{
  switch (c->sh.states)
  {
    case 2: return launch_gen_rows_sc<KIND, 2>(c, b, count, mode);
    case 3: return launch_gen_rows_sc<KIND, 3>(c, b, count, mode);
    case 4: return launch_gen_rows_sc<KIND, 4>(c, b, count, mode);
    case 5: return launch_gen_rows_sc<KIND, 5>(c, b, count, mode);
    case 6: return launch_gen_rows_sc<KIND, 6>(c, b, count, mode);
    case 7: return launch_gen_rows_sc<KIND, 7>(c, b, count, mode);
    case 8: return launch_gen_rows_sc<KIND, 8>(c, b, count, mode);
    case 9: return launch_gen_rows_sc<KIND, 9>(c, b, count, mode);
    case 10: return launch_gen_rows_sc<KIND, 10>(c, b, count, mode);
    case 11: return launch_gen_rows_sc<KIND, 11>(c, b, count, mode);
    case 12: return launch_gen_rows_sc<KIND, 12>(c, b, count, mode);
    case 13: return launch_gen_rows_sc<KIND, 13>(c, b, count, mode);
    case 14: return launch_gen_rows_sc<KIND, 14>(c, b, count, mode);
    case 15: return launch_gen_rows_sc<KIND, 15>(c, b, count, mode);
    default: return launch_gen_rows_sc<KIND, 16>(c, b, count, mode);
  }
}

// ------------------------------------------------ 9 to 64 states: P rows in registers
//
// From ~10 states on the update is bound by arithmetic and operand delivery, not by HBM
// (61 states: 10 flop per byte), and the LDS-tiled kernel above spends two LDS reads per
// multiply-add with little occupancy.  Here one LANE owns one output state i of one site
// and keeps row i of the current P-matrix in registers (states padded to SP, a multiple
// of 8, with zeros).  LPS = 16, 32 or 64 lanes make one site, so a wave works on
// Q = 64 / LPS sites at once; the child rows of a GROUP of WT x Q consecutive sites are
// copied into a wave-private LDS block (coalesced 8 x states byte rows) and read back as
// broadcasts within each site's lanes, two columns per 16-byte read:
//     x_i = sum_j P[i][j] (VGPR) * l[j] (one LDS address per site)
// so a multiply-add costs half an LDS instruction and no address arithmetic.  Per rate
// category the wave loads its rows of P_l, forms x for the group (x goes back into the
// LDS row it was computed from), loads its rows of P_r, forms y and stores x * y.
// Waves never synchronise with each other.  A tip child contributes the sum of the P
// entries its state mask selects, ascending (core_partials.c:113-127): a row of a per-op
// table [character][rate][state] built by k_gen_tip_tables, so a tip-tip update is two
// table rows multiplied and a tip-inner update needs one P-row pass instead of two.
// Whether a site (or a (site, rate) row) must be rescaled is known only after all its
// entries exist; that is rare (once every 10-20 tree levels), so the products are
// stored unscaled and the few rows concerned are multiplied by 2^256 in place
// afterwards by the lanes that wrote them.
// The LDS columns beyond the last state stay zero and so do the P columns: the padded
// terms add +0.0, which leaves every sum bit-exact.
// steps per group: as many as keep a workgroup's four LDS blocks within 64 KB (two workgroups per CU)
__host__ __device__ constexpr int gen_wide_wt(int SP, int LPS)
{
  return LPS == 64 ? 16 : LPS == 32 ? (SP == 32 ? 14 : 16) : 12;
}

// ORD 1 = 20 states in the order of the reference's AVX2-flag kernels (numerics.hpp): inner
// children four accumulators strided by j mod 4 and a pairwise tree, fused for inner-inner
// updates (core_partials_avx2.c:671-750), multiply-then-add for the inner side of tip-inner
// updates (core_partials_avx.c:1237-1262); tip sums ascending as everywhere.  This is what
// runs 20-state data when the matrix-core kernels do not apply (PLLHIP_AA_EXACT=1, rate_cats
// other than 1, 2, 4).
template <int ORD, bool FUSED>
__device__ __forceinline__ void wide_term(double (&acc)[4], int j, double p, double v)
{
  if (ORD == 0) acc[0] += p * v;
  else if (FUSED) acc[j & 3] = fma(p, v, acc[j & 3]);
  else acc[j & 3] = acc[j & 3] + p * v;
}

template <int ORD>
__device__ __forceinline__ double wide_sum(const double (&acc)[4])
{
  return ORD == 0 ? acc[0] : pairsum4(acc[0], acc[1], acc[2], acc[3]);
}

template <int KIND, int SP, int LPS, int ORD = 0> // SP: states rounded up to a multiple of 8; LPS: lanes per site
__global__ __launch_bounds__(256) void k_gen_wide(PartialsBatch batch, int mode)
{
  const PartialsArgs & a = batch.op[blockIdx.y];
  extern __shared__ double smem[];
  constexpr int WT = gen_wide_wt(SP, LPS), Q = 64 / LPS, ROWS = WT * Q;
  constexpr int SPL = (Q == 1) ? SP : SP + 2; // LDS row stride: the Q rows read together fall in different banks
  const unsigned int S = a.states, R = a.rate_cats;
  const unsigned int lane = threadIdx.x & 63u;
  const unsigned int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned int q = lane / LPS, li = lane % LPS;
  const bool lane_on = li < S;
  const unsigned int i = lane_on ? li : S - 1u; // idle lanes shadow the last state
  double * s_l = smem + (size_t)wave * (2 * ROWS * SPL); // [ROWS][SPL] rows of child 1, then x
  double * s_r = s_l + ROWS * SPL;                       // rows of child 2
  for (unsigned int t = lane; t < 2u * ROWS * SPL; t += 64u) s_l[t] = 0.0;
  const size_t groups = ((size_t)a.sites + ROWS - 1) / ROWS;
  constexpr unsigned long long SLOT_FULL = (LPS == 64) ? ~0ull : ((1ull << (LPS % 64)) - 1ull);

  for (size_t grp = (size_t)blockIdx.x * 4u + wave; grp < groups; grp += (size_t)gridDim.x * 4u)
  {
    const size_t n0 = grp * ROWS;
    unsigned long long site_small = ~0ull; // bit r: every entry of site n0 + r below the threshold so far
    // characters of the tip children for the sites this lane's slot meets (row t * Q + q);
    // a tip's factor is a row of its per-op table ltab / rtab[code][rate][state] (the sum
    // of the P entries the character's state mask selects, k_gen_tip_tables)
    unsigned int lcode[KIND >= 1 ? WT : 1], rcode[KIND == 2 ? WT : 1];
    if (KIND >= 1)
#pragma unroll
      for (int t = 0; t < WT; ++t)
      {
        const unsigned int ch = a.ltip[n0 + t * Q + q];
        lcode[t] = ch < a.maxstates ? ch : 0u;
      }
    if (KIND == 2)
#pragma unroll
      for (int t = 0; t < WT; ++t)
      {
        const unsigned int ch = a.rtip[n0 + t * Q + q];
        rcode[t] = ch < a.maxstates ? ch : 0u;
      }

    for (unsigned int k = 0; k < R; ++k)
    {
      double P[SP];
      {
        // the rows of the inner children for this category: up to 2 x WT loads in flight, then LDS
        // (sites past the end read the zeroed slack behind the CLV)
        double lv[KIND == 0 ? WT : 1], rv[KIND != 2 ? WT : 1];
        if (KIND == 0)
#pragma unroll
          for (int t = 0; t < WT; ++t) lv[t] = a.left[((n0 + t * Q + q) * R + k) * S + i];
        if (KIND != 2)
#pragma unroll
          for (int t = 0; t < WT; ++t) rv[t] = a.right[((n0 + t * Q + q) * R + k) * S + i];
        if (lane_on)
        {
          if (KIND == 0)
#pragma unroll
            for (int t = 0; t < WT; ++t) s_l[(t * Q + q) * SPL + li] = lv[t];
          if (KIND != 2)
#pragma unroll
            for (int t = 0; t < WT; ++t) s_r[(t * Q + q) * SPL + li] = rv[t];
        }
      }
      // row i of a P-matrix, zero beyond the last state (only the last chunk of 8 can be partial)
      auto load_p_row = [&](const double * prow) {
#pragma unroll
        for (int j = 0; j < SP - 8; ++j) P[j] = prow[j];
#pragma unroll
        for (int j = SP - 8; j < SP; ++j)
        {
          const double v = prow[(unsigned int)j < S ? (unsigned int)j : S - 1u]; // unconditional load
          P[j] = ((unsigned int)j < S) ? v : 0.0;
        }
      };
      // (the site loops stay rolled: one step's broadcast reads are enough to keep in
      // flight, and x goes back into the row it was computed from -- every lane of the
      // site is done with that row, LDS executes a wave's accesses in order)
      if (KIND == 0)
      {
        load_p_row(a.lmat + ((size_t)k * S + i) * S);
#pragma unroll 1
        for (int t = 0; t < WT; ++t)
        {
          const double * row = s_l + (t * Q + q) * SPL;
          double a4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int j = 0; j < SP; ++j) wide_term<ORD, true>(a4, j, P[j], row[j]);
          const double acc = wide_sum<ORD>(a4);
          if (lane_on) s_l[(t * Q + q) * SPL + li] = acc;
        }
      }
      if (KIND != 2) load_p_row(a.rmat + ((size_t)k * S + i) * S);
      unsigned long long rate_small = 0ull;
#pragma unroll 1
      for (int t = 0; t < WT; ++t)
      {
        const double * row = s_r + (t * Q + q) * SPL;
        double acc;
        if (KIND != 2)
        {
          double a4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int j = 0; j < SP; ++j) wide_term<ORD, KIND == 0>(a4, j, P[j], row[j]);
          acc = wide_sum<ORD>(a4);
        }
        else
          acc = a.rtab[((size_t)rcode[t] * R + k) * S + i];
        const size_t n = n0 + t * Q + q;
        const double x = (KIND == 0) ? s_l[(t * Q + q) * SPL + i] : a.ltab[((size_t)lcode[t] * R + k) * S + i];
        const double p = x * acc;
        if (lane_on && n < a.sites) a.parent[(n * R + k) * S + i] = p;
        if (KIND != 2)
        {
          const unsigned long long b = __ballot(p < PLLHIP_SCALE_THRESHOLD || !lane_on);
#pragma unroll
          for (int qq = 0; qq < Q; ++qq)
            if (((b >> (qq * LPS)) & SLOT_FULL) == SLOT_FULL) rate_small |= 1ull << (t * Q + qq);
        }
      }
      site_small &= rate_small;
      if (mode == SCALE_RATE)
      {
        if (rate_small)
#pragma unroll 1
          for (int t = 0; t < WT; ++t)
          {
            const size_t n = n0 + t * Q + q;
            if (((rate_small >> (t * Q + q)) & 1ull) && lane_on && n < a.sites)
              a.parent[(n * R + k) * S + i] *= PLLHIP_SCALE_FACTOR;
          }
        if (lane < (unsigned int)ROWS && n0 + lane < a.sites)
        {
          const size_t w = (n0 + lane) * R + k;
          a.pscaler[w] = (KIND == 2) ? 0u
                                     : ((KIND == 0 && a.lscaler) ? a.lscaler[w] : 0u) +
                                           (a.rscaler ? a.rscaler[w] : 0u) +
                                           (unsigned int)((rate_small >> lane) & 1ull);
        }
      }
    }
    if (mode == SCALE_SITE)
    {
      if (KIND != 2 && site_small)
#pragma unroll 1
        for (int t = 0; t < WT; ++t)
        {
          const size_t n = n0 + t * Q + q;
          if (((site_small >> (t * Q + q)) & 1ull) && lane_on && n < a.sites)
            for (unsigned int k = 0; k < R; ++k) a.parent[(n * R + k) * S + i] *= PLLHIP_SCALE_FACTOR;
        }
      if (lane < (unsigned int)ROWS && n0 + lane < a.sites)
        a.pscaler[n0 + lane] = (KIND == 2) ? 0u
                                           : ((KIND == 0 && a.lscaler) ? a.lscaler[n0 + lane] : 0u) +
                                                 (a.rscaler ? a.rscaler[n0 + lane] : 0u) +
                                                 (unsigned int)((site_small >> lane) & 1ull);
    }
  }
}

// tip tables of the ops of a batch: tab[op][side][code][k][i] = sum of P_k[i][j] over the
// states j of tipmap[code], ascending (core_partials.c:725-770 builds the same sums as its
// tip-tip lookup)
__global__ __launch_bounds__(256) void k_gen_tip_tables(PartialsBatch batch, double * __restrict__ tab,
                                                        unsigned int maxstates, int both)
{
  const PartialsArgs & a = batch.op[blockIdx.y];
  const unsigned int S = a.states, RS = a.rate_cats * S;
  const unsigned int per = maxstates * RS;
  const unsigned int total = both ? 2 * per : per;
  double * out = tab + (size_t)blockIdx.y * 2 * per;
  for (unsigned int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x)
  {
    const unsigned int side = t / per, u = t % per;
    const unsigned int code = u / RS, ki = u % RS;
    out[t] = masksum_seq((side ? a.rmat : a.lmat) + (size_t)ki * S, a.tipmap[code], S);
  }
}

static bool gen_wide_covers(unsigned int S, int kind)
{
  // tip children are 32-bit state masks: tip kinds exist up to 32 states only
  return S >= 9 && S <= 64 && (kind == 0 || S <= 32);
}

template <int KIND, int SP, int LPS, int ORD = 0>
static int launch_gen_wide_shape(pllhip_ctx * c, const PartialsBatch & b, const dim3 & grid, int mode)
{
  const size_t lds = 4 * 2 * (size_t)(gen_wide_wt(SP, LPS) * (64 / LPS)) * (LPS == 64 ? SP : SP + 2) * sizeof(double);
  if (lds > 65536)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gen_wide<KIND, SP, LPS, ORD>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  k_gen_wide<KIND, SP, LPS, ORD><<<grid, 256, lds, c->stream>>>(b, mode);
  HIP_TRY(hipGetLastError());
  return 0;
}

template <int KIND>
static int launch_gen_wide(pllhip_ctx * c, const PartialsBatch & b_in, unsigned int count, int mode)
{
  size_t sites = 0;
  for (unsigned int i = 0; i < count; ++i)
    if (b_in.op[i].sites > sites) sites = b_in.op[i].sites;
  if (!sites) return 0;
  const unsigned int S = c->sh.states;
  PartialsBatch b = b_in;
  if (KIND >= 1)
  {
    // tip row-sum tables of every op of the batch, one launch (the buffer is reused by
    // the next batch: same stream, so that launch waits for this batch's kernel)
    if (!c->maxstates)
    {
      pllhip_set_error("tip-state update: tipmap not uploaded");
      return -1;
    }
    const size_t per = (size_t)c->maxstates * c->sh.rate_cats * S;
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
    k_gen_tip_tables<<<dim3(8, count), 256, 0, c->stream>>>(b, c->d_tiptab, c->maxstates, KIND == 2 ? 1 : 0);
    HIP_TRY(hipGetLastError());
    for (unsigned int i = 0; i < count; ++i)
    {
      b.op[i].ltab = c->d_tiptab + (size_t)i * 2 * per;
      b.op[i].rtab = b.op[i].ltab + per;
      b.op[i].maxstates = c->maxstates;
    }
  }
  const size_t rows = S <= 16 ? gen_wide_wt(16, 16) * 4 : S <= 24 ? gen_wide_wt(24, 32) * 2
                      : S <= 32 ? gen_wide_wt(32, 32) * 2 : gen_wide_wt(64, 64);
  const size_t groups = (sites + rows - 1) / rows;
  const size_t cap = (size_t)c->num_cus * 2; // 2 workgroups of 4 waves per CU
  const size_t need = (groups + 3) / 4;
  const dim3 grid((unsigned int)(need < cap ? need : cap), count);
  if (S == 20) return launch_gen_wide_shape<KIND, 24, 32, 1>(c, b, grid, mode);
  if (S <= 16) return launch_gen_wide_shape<KIND, 16, 16>(c, b, grid, mode);
  if (S <= 24) return launch_gen_wide_shape<KIND, 24, 32>(c, b, grid, mode);
  if (S <= 32) return launch_gen_wide_shape<KIND, 32, 32>(c, b, grid, mode);
  if (KIND == 0)
  {
    if (S <= 40) return launch_gen_wide_shape<0, 40, 64>(c, b, grid, mode);
    if (S <= 48) return launch_gen_wide_shape<0, 48, 64>(c, b, grid, mode);
    if (S <= 56) return launch_gen_wide_shape<0, 56, 64>(c, b, grid, mode);
    return launch_gen_wide_shape<0, 64, 64>(c, b, grid, mode);
  }
  pllhip_set_error("tip-state kernels cover at most 32 states");
  return -1;
}

static size_t gen_tile_lds(unsigned int S, unsigned int R, int kind, unsigned int ts, bool resident)
{
  const size_t span = (size_t)S * R;
  const size_t images = (kind == 0 ? 3 : kind == 1 ? 2 : 1);
  return 2 * (size_t)(resident ? R : 1) * S * S * sizeof(double) + images * ts * span * sizeof(double) +
         (size_t)ts * R * sizeof(unsigned int) + 2 * (size_t)ts * sizeof(unsigned int);
}

// tile geometry of a partition; ts == 0: the shape is not covered (more than 64 states)
static GenTileGeom gen_tile_geom(const pllhip_ctx * c, int kind, size_t * lds)
{
  const unsigned int S = c->sh.states, R = c->sh.rate_cats;
  GenTileGeom g = {0u, 0u, 0u, 0u};
  if (S == 4 || S == 20 || S < 2 || S > 64 || R > 64) return g;
  g.resident = (2 * (size_t)R * S * S * sizeof(double) <= 32768) ? 1u : 0u;
  // two workgroups per CU while the matrices are small, one (150 KB) for the large ones
  const size_t budget = g.resident ? 65536 : 150 * 1024;
  const size_t fixed = gen_tile_lds(S, R, kind, 0, g.resident);
  const size_t per_site = gen_tile_lds(S, R, kind, 1, g.resident) - fixed;
  if (fixed + per_site > budget) return g;
  size_t ts = (budget - fixed) / per_site;
  // a tile should give every lane a few outputs per category, not more: smaller tiles
  // balance better over the workgroups
  const size_t want = (8 * 256 + S - 1) / S;
  if (ts > want) ts = want;
  if (ts > 256) ts = 256;
  g.ts = (unsigned int)ts;
  g.inv_s = (unsigned int)((0x100000000ull + S - 1) / S);
  g.inv_span = (unsigned int)((0x100000000ull + (size_t)S * R - 1) / ((size_t)S * R));
  *lds = gen_tile_lds(S, R, kind, g.ts, g.resident);
  return g;
}

// does one of the kernels of this file cover the partition's shape?  (4 and 20 states are
// asked only after their dedicated kernels declined)
bool pllhip_gen_tile_covers(const pllhip_ctx * c)
{
  size_t lds;
  if (c->sh.states == 20) return true; // the P-row kernels in the AVX2-flag order
  if (gen_rows_covers(c->sh.states, c->sh.rate_cats)) return true;
  return gen_tile_geom(c, 0, &lds).ts != 0u;
}

template <int KIND>
static int launch_gen_tile(pllhip_ctx * c, const PartialsBatch & b, unsigned int count, int mode)
{
  size_t lds = 0;
  const GenTileGeom g = gen_tile_geom(c, KIND, &lds);
  size_t tiles = 0;
  for (unsigned int i = 0; i < count; ++i)
  {
    const size_t t = ((size_t)b.op[i].sites + g.ts - 1) / g.ts;
    if (t > tiles) tiles = t;
  }
  if (!tiles) return 0;
  const size_t per_cu = g.resident ? 2 : 1;
  const size_t cap = (size_t)c->num_cus * per_cu * 2; // two tiles in flight per resident workgroup slot
  if (lds > 65536)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gen_tile<KIND>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const dim3 grid((unsigned int)(tiles < cap ? tiles : cap), count);
  k_gen_tile<KIND><<<grid, 256, lds, c->stream>>>(b, mode, g);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ops of one kind and scaling mode, mutually independent, in one launch
int pllhip_launch_gen_batch(pllhip_ctx * c, const PartialsBatch & b, unsigned int count, int kind, int mode)
{
  if (c->sh.states != 20 && gen_rows_covers(c->sh.states, c->sh.rate_cats))
  {
    if (kind == 0) return launch_gen_rows<0>(c, b, count, mode);
    if (kind == 1) return launch_gen_rows<1>(c, b, count, mode);
    return launch_gen_rows<2>(c, b, count, mode);
  }
  if (gen_wide_covers(c->sh.states, kind))
  {
    if (kind == 0) return launch_gen_wide<0>(c, b, count, mode);
    if (kind == 1) return launch_gen_wide<1>(c, b, count, mode);
    return launch_gen_wide<2>(c, b, count, mode);
  }
  if (kind == 0) return launch_gen_tile<0>(c, b, count, mode);
  if (kind == 1) return launch_gen_tile<1>(c, b, count, mode);
  return launch_gen_tile<2>(c, b, count, mode);
}
