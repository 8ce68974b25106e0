// likelihood_aa_mfma.hip -- 20-state edge / root log-likelihood on the f64 matrix cores.
//
// Replaces pll_core_edge_loglikelihood_ii (core_likelihood_avx2.c:333),
// _ti_20x20 (core_likelihood_avx2.c:111) and the 20-state root kernel
// (core_likelihood_avx2.c:25).  Same tiling as partials_aa_mfma.hip: a wave owns
// 16 sites x RC categories; the child tile goes through the LDS image and the
// 25 MFMAs per category (P . child), then the PARENT tile is read from the image
// in the accumulator's own layout (lane (s, q): states 4g+q of site s), so
//   term[k] = sum_g ( x[k][g] * pi_k[4g+q] ) * parent[k][g]
// is register-local; two __shfl_xor (16, 32) add the four q-lanes of a site.
// Lanes 0..15 then own one site each: category weights, +I term, log, scalers,
// pattern weight, optional per-site store.  HBM: 1292 B per site.
//
// Numerics: one fused chain per row (as the CLV kernel), then lane sums in a
// fixed order; per-site lnL agrees with the reference to ~1e-15 relative.
// PLLHIP_AA_EXACT=1 selects the bit-exact vector kernel (likelihood.hip).
//
// H (round 4): EIGHT rate categories as two halves of four per tile (H = 2, RC = 4): a tile of 16 sites x 8
// categories is two images, so the wave takes the child and parent tiles of categories 0..3, then those of 4..7 --
// four DMAs per tile instead of two, each of half the size, all eight matrices (or the 8-category tip table) in LDS --
// and keeps the eight category terms in registers for the per-site tail.  Until then such partitions took the vector
// kernel: 1.46 ms per call on 200,000 sites, a fifth of a whole evaluation.
#include <stdlib.h>

#include "aa_mfma.hpp"
#include "lnl_common.hpp"

template <int RC, int KIND, bool NT, bool GATHER, int H = 1>
__global__ __launch_bounds__(256, 2) void k_lnl_aa_mfma(LnlArgs a)
{
  static_assert(H == 1 || (H == 2 && RC == 4 && !GATHER), "halves of 8 categories; no site repeats");
  constexpr int RT = RC * H; // categories of the CLVs
  using G = aa_geom<RC>;
  extern __shared__ double smem[];
  // LDS: [P-matrices RT x 20 x 20 (II) | pi-weighted tip table maxstates x RT x 20 (TI)][4 images]
  const unsigned int head = (KIND == EDGE_II) ? RT * 400u : (KIND == EDGE_TI ? a.maxstates * RT * 20u : 0u);
  double * tab = smem;
  if (KIND == EDGE_II)
    for (unsigned int t = threadIdx.x; t < head; t += blockDim.x) tab[t] = a.pmat[t];
  if (KIND == EDGE_TI)
    for (unsigned int t = threadIdx.x; t < head; t += blockDim.x)
    {
      const unsigned int code = t / (RT * 20), kk = (t / 20) % RT, j = t % 20;
      // rowsum * pi (core_likelihood_avx2.c:191-233)
      tab[t] = masksum_seq(a.pmat + ((size_t)kk * 20 + j) * 20, a.tipmap[code], 20) *
               a.freqs[(size_t)a.freqs_indices[kk] * 20 + j];
    }
  // per-category model words the per-site tail needs: in LDS, because a global load
  // issued while the next tile's DMA is in flight returns only after that DMA
  __shared__ double s_model[RT][2];  // prop_invar, rate weight
  __shared__ double s_freqs[RT][20]; // frequencies of the category's rate matrix
  for (unsigned int t = threadIdx.x; t < RT * 20u; t += blockDim.x)
    s_freqs[t / 20u][t % 20u] = a.freqs[(size_t)a.freqs_indices[t / 20u] * 20 + t % 20u];
  if (threadIdx.x < RT)
  {
    s_model[threadIdx.x][0] = a.prop_invar[a.freqs_indices[threadIdx.x]];
    s_model[threadIdx.x][1] = a.rate_weights[threadIdx.x];
  }
  __syncthreads();

  const unsigned int lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned int s = lane & 15u, q = lane >> 4;
  char * region = reinterpret_cast<char *>(smem + head) + wave * G::REGION_B;

  // pi_k[4g+q] for this lane (H = 2: read from LDS half by half, 80 registers otherwise)
  double pi[RC][5];
  auto load_pi = [&](int half) {
#pragma unroll
    for (int k = 0; k < RC; ++k)
#pragma unroll
      for (int g = 0; g < 5; ++g) pi[k][g] = s_freqs[half * RC + k][4 * g + q];
  };
  if (H == 1) load_pi(0);

  unsigned int toff[G::N_IT];
  tile_offsets<RC, RT>(lane, toff); // (a half's categories: the CLV pointer is moved, not the offsets)
  const size_t sites = a.sites;
  const size_t tiles = (sites + 15) / 16;
  const size_t nwaves = (size_t)gridDim.x * 4;
  const bool per_rate = a.rate_scalers && KIND != ROOT;
  double acc = 0.0;

  // Schedule of a tile (one LDS image per wave): its first operand tile -- the child
  // for EDGE_II, the parent otherwise -- and the per-site words (scaler counts, pattern
  // weight, invariant-state index, tip code) were requested while the PREVIOUS tile was
  // being finished; per-site words first, because loads return in order and anything
  // requested behind a DMA costs that DMA's latency at its first use.  Nothing is
  // clamped: every per-site array has PLLHIP_TAIL_SITES of slack.
  const size_t first = (size_t)blockIdx.x * 4 + wave;
  const bool has_ps = a.pscaler != nullptr, has_cs = (KIND == EDGE_II && a.cscaler != nullptr);
  const unsigned int * psp = has_ps ? a.pscaler : a.zero;
  const unsigned int * csp = has_cs ? a.cscaler : a.zero;
  const int * invp = a.invariant ? a.invariant : reinterpret_cast<const int *>(a.zero);
  const bool has_inv = a.invariant != nullptr;
  unsigned int w_next = 0, code_next = 0, ps_next[RT], cs_next[RT];
  int inv_next = -1;
  // Site repeats (GATHER): site n of a CLV stored by class lives in row a.pidx[n] /
  // a.cidx[n] (nullptr = n), and so do its scaler counts.  The rows of a tile are needed
  // when its first operand and its per-site words are requested -- one tile ahead -- so
  // they are fetched two tiles ahead, behind that request.
  unsigned int prow_next = 0, crow_next = 0, prow_next2 = 0, crow_next2 = 0;
  auto rows_of = [&](const unsigned int * idx, size_t tile) -> unsigned int {
    const size_t n = tile * 16 + s;
    return (GATHER && idx) ? idx[n] : (unsigned int)n; // (the maps carry slack)
  };
  auto request_tile = [&](size_t tile, unsigned int prow, unsigned int crow) {
    const size_t n = tile * 16 + s;
    w_next = a.pattern_weights[n];
    inv_next = invp[has_inv ? n : 0];
    if (KIND == EDGE_TI) code_next = a.tip[n];
#pragma unroll
    for (int k = 0; k < RT; ++k)
    {
      const bool used = per_rate || k == 0;
      const size_t ep = per_rate ? (size_t)prow * RT + k : (size_t)prow;
      const size_t ec = per_rate ? (size_t)crow * RT + k : (size_t)crow;
      ps_next[k] = used ? psp[has_ps ? ep : 0] : 0u;
      cs_next[k] = used ? csp[has_cs ? ec : 0] : 0u;
    }
    if (!GATHER) dma_tile<RC, NT, RT>(KIND == EDGE_II ? a.child : a.parent, tile * 16, toff, region);
    else if (KIND == EDGE_II) dma_tile_rows<RC, NT>(a.child, crow, toff, region);
    else dma_tile_rows<RC, NT>(a.parent, prow, toff, region);
  };
  if (first < tiles)
  {
    unsigned int p0 = rows_of(a.pidx, first), c0 = rows_of(a.cidx, first);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(p0), "+v"(c0));
    prow_next = p0;
    crow_next = c0;
    const size_t second = first + nwaves < tiles ? first + nwaves : first;
    prow_next2 = rows_of(a.pidx, second);
    crow_next2 = rows_of(a.cidx, second);
    request_tile(first, p0, c0);
  }
  for (size_t tile = first; tile < tiles; tile += nwaves)
  {
    const size_t site0 = tile * 16;
    const size_t next = tile + nwaves;
    double b[RC][5], x[RC][5];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // this tile's per-site words have landed with the DMA: take them out of the
    // registers the next request will overwrite
    unsigned int w_cur = w_next, code = code_next, sc_cur[RT]; // (scaler counts: parent + child)
    int inv_cur = has_inv ? inv_next : -1;
    const unsigned int prow_cur = prow_next; // rows of THIS tile (its second operand is still to come)
    prow_next = prow_next2;
    crow_next = crow_next2;
    asm volatile("" : "+v"(w_cur), "+v"(code), "+v"(inv_cur), "+v"(prow_next), "+v"(crow_next));
#pragma unroll
    for (int k = 0; k < RT; ++k)
    {
      sc_cur[k] = ps_next[k] + cs_next[k];
      asm volatile("" : "+v"(sc_cur[k]));
    }
    if (KIND == EDGE_TI && code >= a.maxstates) code = 0;

    double term[RT];
    // one half of the tile's categories (H = 1: all of them); its first operand tile is in the image
    auto half_terms = [&](auto hc) __attribute__((always_inline)) {
      constexpr int h = decltype(hc)::value;
      if (h >= H) return;
      if (h > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (H > 1) load_pi(h);
      if (KIND == EDGE_II)
      {
        read_b_operands<RC>(region, s, q, b);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (GATHER) dma_tile_rows<RC, NT>(a.parent, prow_cur, toff, region);
        else dma_tile<RC, NT, RT>(a.parent + h * RC * 20, site0, toff, region);
        tile_matvec<RC>(tab + h * RC * 400, b, lane, x);
#pragma unroll
        for (int k = 0; k < RC; ++k)
#pragma unroll
          for (int g = 0; g < 5; ++g) x[k][g] = x[k][g] * pi[k][g];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      else
      {
#pragma unroll
        for (int k = 0; k < RC; ++k)
#pragma unroll
          for (int g = 0; g < 5; ++g)
            x[k][g] = (KIND == EDGE_TI) ? tab[(code * RT + h * RC + k) * 20 + 4 * g + q] : pi[k][g];
      }
      read_b_operands<RC>(region, s, q, b); // parent CLV, states 4g+q
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (h + 1 < H)
        // the other half of this tile
        dma_tile<RC, NT, RT>((KIND == EDGE_II ? a.child : a.parent) + (h + 1) * RC * 20, site0, toff, region);
      else if (next < tiles)
      {
        request_tile(next, prow_next, crow_next);
        const size_t after = next + nwaves < tiles ? next + nwaves : next;
        prow_next2 = rows_of(a.pidx, after);
        crow_next2 = rows_of(a.cidx, after);
      }
#pragma unroll
      for (int k = 0; k < RC; ++k)
      {
        double t = 0.0;
#pragma unroll
        for (int g = 0; g < 5; ++g) t += x[k][g] * b[k][g];
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        term[(h < H ? h : 0) * RC + k] = t;
      }
    };
    half_terms(std::integral_constant<int, 0>{});
    half_terms(std::integral_constant<int, 1>{});

    // lanes 0..15: one site each
    const size_t n = site0 + s;
    if (q == 0 && n < sites)
    {
      unsigned int site_scalings = 0, rel[RT];
      if (per_rate)
      {
        unsigned int mn = 0xffffffffu;
#pragma unroll
        for (int k = 0; k < RT; ++k)
        {
          const unsigned int v = sc_cur[k];
          rel[k] = v;
          mn = v < mn ? v : mn;
        }
        site_scalings = mn;
#pragma unroll
        for (int k = 0; k < RT; ++k)
        {
          const unsigned int d = rel[k] - mn;
          rel[k] = d > PLLHIP_SCALE_RATE_MAXDIFF ? PLLHIP_SCALE_RATE_MAXDIFF : d;
        }
      }
      else
      {
#pragma unroll
        for (int k = 0; k < RT; ++k) rel[k] = 0;
        site_scalings = sc_cur[0];
      }
      double terma = 0.0;
#pragma unroll
      for (int k = 0; k < RT; ++k)
      {
        // category term -> weighted contribution (core_likelihood_avx2.c:480-500)
        double tr = term[k];
        if (rel[k] > 0) tr *= scale_minlh(rel[k]);
        const double pinv = s_model[k][0];
        const double w = s_model[k][1];
        if (pinv > 0.0)
        {
          const double inv_lk = (inv_cur == -1) ? 0.0 : s_freqs[k][inv_cur];
          terma += w * (tr * (1.0 - pinv) + inv_lk * pinv);
        }
        else
          terma += tr * w;
      }
      double lk = log(terma);
      if (site_scalings) lk += (double)site_scalings * log(PLLHIP_SCALE_THRESHOLD);
      lk *= (double)w_cur;
      if (a.persite) a.persite[n] = lk;
      acc += lk;
    }
  }
  block_sum_to_partials(acc, a.reduce);
}

template <int RC, int H = 1>
static int launch_lnl_rc(pllhip_ctx * c, LnlArgs & a, int kind, unsigned int * grid_out)
{
  using G = aa_geom<RC>;
  constexpr int RT = RC * H;
  const size_t tiles = ((size_t)a.sites + 15) / 16;
  size_t blocks = (tiles + 3) / 4;
  const size_t cap = getenv("PLLHIP_AA_GRID_CAP") ? (size_t)atoi(getenv("PLLHIP_AA_GRID_CAP")) : (size_t)c->num_cus * 2; // (tests: many tiles per wave)
  if (blocks > cap) blocks = cap;
  const size_t head = (kind == EDGE_II) ? (size_t)RT * 400
                                        : (kind == EDGE_TI ? (size_t)a.maxstates * RT * 20 : 0);
  const size_t lds = head * sizeof(double) + 4 * (size_t)G::REGION_B;
  if (lds > 80 * 1024) return 1;
  const bool nt = pllhip_use_nt(c);
  const bool gather = a.pidx || a.cidx;
  if (H > 1 && gather) return 1;
  const dim3 grid((unsigned int)blocks), block(256);
  a.reduce = pllhip_reduce_out(c, (unsigned int)blocks);
#define LNL_ONE(KERNEL)                                                                        \
  do {                                                                                         \
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&KERNEL),                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));        \
    hipLaunchKernelGGL(KERNEL, grid, block, lds, c->stream, a);                                \
  } while (0)
#define LNL_KIND(KINDV)                                                     \
  do {                                                                      \
    if (H > 1) {                                                            \
      if (nt) LNL_ONE((k_lnl_aa_mfma<RC, KINDV, true, false, H>));          \
      else LNL_ONE((k_lnl_aa_mfma<RC, KINDV, false, false, H>));            \
    }                                                                       \
    else if (gather) LNL_ONE((k_lnl_aa_mfma<RC, KINDV, false, true>));      \
    else if (nt) LNL_ONE((k_lnl_aa_mfma<RC, KINDV, true, false>));          \
    else LNL_ONE((k_lnl_aa_mfma<RC, KINDV, false, false>));                 \
  } while (0)
  if (kind == EDGE_II) LNL_KIND(EDGE_II);
  else if (kind == EDGE_TI) LNL_KIND(EDGE_TI);
  else LNL_KIND(ROOT);
#undef LNL_KIND
#undef LNL_ONE
  HIP_TRY(hipGetLastError());
  *grid_out = (unsigned int)blocks;
  return 0;
}

// returns 1 if not covered (caller falls back to the vector kernels)
int pllhip_launch_lnl_aa_mfma(pllhip_ctx * c, LnlArgs & a, int kind, unsigned int * grid_out)
{
  if (kind == EDGE_TI && (a.maxstates == 0 || a.maxstates > 32)) return 1;
  switch (a.rate_cats)
  {
    case 1: return launch_lnl_rc<1>(c, a, kind, grid_out);
    case 2: return launch_lnl_rc<2>(c, a, kind, grid_out);
    case 4: return launch_lnl_rc<4>(c, a, kind, grid_out);
    case 8:
    {
      static const bool rc8 = !(getenv("PLLHIP_AA_RC8") && atoi(getenv("PLLHIP_AA_RC8")) == 0);
      return rc8 ? launch_lnl_rc<4, 2>(c, a, kind, grid_out) : 1;
    }
    default: return 1;
  }
}
