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
#include <stdlib.h>

#include "aa_mfma.hpp"
#include "lnl_common.hpp"

template <int RC, int KIND, bool NT, bool GATHER>
__global__ __launch_bounds__(256, 2) void k_lnl_aa_mfma(LnlArgs a)
{
  using G = aa_geom<RC>;
  extern __shared__ double smem[];
  // LDS: [P-matrices RC x 20 x 20 (II) | pi-weighted tip table maxstates x RC x 20 (TI)][4 images]
  const unsigned int head = (KIND == EDGE_II) ? RC * 400u : (KIND == EDGE_TI ? a.maxstates * RC * 20u : 0u);
  double * tab = smem;
  if (KIND == EDGE_II)
    for (unsigned int t = threadIdx.x; t < head; t += blockDim.x) tab[t] = a.pmat[t];
  if (KIND == EDGE_TI)
    for (unsigned int t = threadIdx.x; t < head; t += blockDim.x)
    {
      const unsigned int code = t / (RC * 20), kk = (t / 20) % RC, j = t % 20;
      // rowsum * pi (core_likelihood_avx2.c:191-233)
      tab[t] = masksum_seq(a.pmat + ((size_t)kk * 20 + j) * 20, a.tipmap[code], 20) *
               a.freqs[(size_t)a.freqs_indices[kk] * 20 + j];
    }
  // per-category model words the per-site tail needs: in LDS, because a global load
  // issued while the next tile's DMA is in flight returns only after that DMA
  __shared__ double s_model[RC][2];  // prop_invar, rate weight
  __shared__ double s_freqs[RC][20]; // frequencies of the category's rate matrix
  for (unsigned int t = threadIdx.x; t < RC * 20u; t += blockDim.x)
    s_freqs[t / 20u][t % 20u] = a.freqs[(size_t)a.freqs_indices[t / 20u] * 20 + t % 20u];
  if (threadIdx.x < RC)
  {
    s_model[threadIdx.x][0] = a.prop_invar[a.freqs_indices[threadIdx.x]];
    s_model[threadIdx.x][1] = a.rate_weights[threadIdx.x];
  }
  __syncthreads();

  const unsigned int lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned int s = lane & 15u, q = lane >> 4;
  char * region = reinterpret_cast<char *>(smem + head) + wave * G::REGION_B;

  // pi_k[4g+q] for this lane
  double pi[RC][5];
#pragma unroll
  for (int k = 0; k < RC; ++k)
#pragma unroll
    for (int g = 0; g < 5; ++g) pi[k][g] = s_freqs[k][4 * g + q];

  unsigned int toff[G::N_IT];
  tile_offsets<RC>(lane, toff);
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
  unsigned int w_next = 0, code_next = 0, ps_next[RC], cs_next[RC];
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
    for (int k = 0; k < RC; ++k)
    {
      const bool used = per_rate || k == 0;
      const size_t ep = per_rate ? (size_t)prow * RC + k : (a.pscaler_by_site ? n : (size_t)prow);
      const size_t ec = per_rate ? (size_t)crow * RC + k : (size_t)crow;
      ps_next[k] = used ? psp[has_ps ? ep : 0] : 0u;
      cs_next[k] = used ? csp[has_cs ? ec : 0] : 0u;
    }
    if (!GATHER) dma_tile<RC, NT>(KIND == EDGE_II ? a.child : a.parent, tile * 16, toff, region);
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
    unsigned int w_cur = w_next, code = code_next, ps_cur[RC], cs_cur[RC];
    int inv_cur = has_inv ? inv_next : -1;
    const unsigned int prow_cur = prow_next; // rows of THIS tile (its second operand is still to come)
    prow_next = prow_next2;
    crow_next = crow_next2;
    asm volatile("" : "+v"(w_cur), "+v"(code), "+v"(inv_cur), "+v"(prow_next), "+v"(crow_next));
#pragma unroll
    for (int k = 0; k < RC; ++k)
    {
      ps_cur[k] = ps_next[k];
      cs_cur[k] = cs_next[k];
      asm volatile("" : "+v"(ps_cur[k]), "+v"(cs_cur[k]));
    }
    if (KIND == EDGE_II)
    {
      read_b_operands<RC>(region, s, q, b);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (GATHER) dma_tile_rows<RC, NT>(a.parent, prow_cur, toff, region);
      else dma_tile<RC, NT>(a.parent, site0, toff, region);
      tile_matvec<RC>(tab, b, lane, x);
#pragma unroll
      for (int k = 0; k < RC; ++k)
#pragma unroll
        for (int g = 0; g < 5; ++g) x[k][g] = x[k][g] * pi[k][g];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    else
    {
      if (KIND == EDGE_TI && code >= a.maxstates) code = 0;
#pragma unroll
      for (int k = 0; k < RC; ++k)
#pragma unroll
        for (int g = 0; g < 5; ++g)
          x[k][g] = (KIND == EDGE_TI) ? tab[(code * RC + k) * 20 + 4 * g + q] : pi[k][g];
    }
    read_b_operands<RC>(region, s, q, b); // parent CLV, states 4g+q
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (next < tiles)
    {
      request_tile(next, prow_next, crow_next);
      const size_t after = next + nwaves < tiles ? next + nwaves : next;
      prow_next2 = rows_of(a.pidx, after);
      crow_next2 = rows_of(a.cidx, after);
    }

    double term[RC];
#pragma unroll
    for (int k = 0; k < RC; ++k)
    {
      double t = 0.0;
#pragma unroll
      for (int g = 0; g < 5; ++g) t += x[k][g] * b[k][g];
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      term[k] = t;
    }

    // lanes 0..15: one site each
    const size_t n = site0 + s;
    if (q == 0 && n < sites)
    {
      unsigned int site_scalings = 0, rel[RC];
      if (per_rate)
      {
        unsigned int mn = 0xffffffffu;
#pragma unroll
        for (int k = 0; k < RC; ++k)
        {
          const unsigned int v = ps_cur[k] + cs_cur[k];
          rel[k] = v;
          mn = v < mn ? v : mn;
        }
        site_scalings = mn;
#pragma unroll
        for (int k = 0; k < RC; ++k)
        {
          const unsigned int d = rel[k] - mn;
          rel[k] = d > PLLHIP_SCALE_RATE_MAXDIFF ? PLLHIP_SCALE_RATE_MAXDIFF : d;
        }
      }
      else
      {
#pragma unroll
        for (int k = 0; k < RC; ++k) rel[k] = 0;
        site_scalings = ps_cur[0] + cs_cur[0];
      }
      double terma = 0.0;
#pragma unroll
      for (int k = 0; k < RC; ++k)
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

template <int RC>
static int launch_lnl_rc(pllhip_ctx * c, LnlArgs & a, int kind, unsigned int * grid_out)
{
  using G = aa_geom<RC>;
  const size_t tiles = ((size_t)a.sites + 15) / 16;
  size_t blocks = (tiles + 3) / 4;
  const size_t cap = pllhip_env("PLLHIP_AA_GRID_CAP") ? (size_t)atoi(pllhip_env("PLLHIP_AA_GRID_CAP")) : (size_t)c->num_cus * 2; // (tests: many tiles per wave)
  if (blocks > cap) blocks = cap;
  const size_t head = (kind == EDGE_II) ? (size_t)RC * 400
                                        : (kind == EDGE_TI ? (size_t)a.maxstates * RC * 20 : 0);
  const size_t lds = head * sizeof(double) + 4 * (size_t)G::REGION_B;
  if (lds > 80 * 1024) return 1;
  const bool nt = pllhip_use_nt(c);
  const bool gather = a.pidx || a.cidx;
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
    if (gather) LNL_ONE((k_lnl_aa_mfma<RC, KINDV, false, true>));           \
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

// ---- category counts other than 1, 2, 4 (round 4): the tile's categories in CHUNKS of RC (the largest of 4, 2, 1 that
// divides the count): per chunk the child tile, the 25 MFMAs per category, the parent tile -- 2 x rate_cats / RC DMAs of
// RC x 160 bytes per site instead of two of the whole row --, all matrices (or the whole tip table) in LDS, the site's
// sum over categories accumulated chunk by chunk in the lane that finishes the site (same order of additions as the
// kernel above: k = 0, 1, 2 ...).  Until then such partitions took the vector kernel (likelihood.hip): 8 categories,
// 200,000 sites 1,458 us per call, this kernel 87.  No site repeats (such partitions never store by class).
template <int RC, int KIND, bool NT>
__global__ __launch_bounds__(256, 2) void k_lnl_aa_chunks(LnlArgs a)
{
  using G = aa_geom<RC>;
  extern __shared__ double smem[];
  const unsigned int RT = a.rate_cats, H = RT / RC;
  // LDS: [P-matrices RT x 20 x 20 (II) | pi-weighted tip table maxstates x RT x 20 (TI)][4 images][freqs RT x 20][model RT x 2]
  const unsigned int head = (KIND == EDGE_II) ? RT * 400u : (KIND == EDGE_TI ? a.maxstates * RT * 20u : 0u);
  double * tab = smem;
  double * s_freqs = smem + head + 4 * (G::REGION_B / 8); // [k][20]
  double * s_model = s_freqs + RT * 20u;                  // [k][prop_invar, rate weight]
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
  for (unsigned int t = threadIdx.x; t < RT * 20u; t += blockDim.x)
    s_freqs[t] = a.freqs[(size_t)a.freqs_indices[t / 20u] * 20 + t % 20u];
  for (unsigned int t = threadIdx.x; t < RT; t += blockDim.x)
  {
    s_model[2 * t] = a.prop_invar[a.freqs_indices[t]];
    s_model[2 * t + 1] = a.rate_weights[t];
  }
  __syncthreads();

  const unsigned int lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned int s = lane & 15u, q = lane >> 4;
  char * region = reinterpret_cast<char *>(smem + head) + wave * G::REGION_B;
  unsigned int toff[G::N_IT];
  tile_offsets<RC>(lane, toff, RT); // (a chunk's categories: the CLV pointer is moved, not the offsets)
  const size_t sites = a.sites;
  const size_t tiles = (sites + 15) / 16;
  const size_t nwaves = (size_t)gridDim.x * 4;
  const bool per_rate = a.rate_scalers && KIND != ROOT;
  double acc = 0.0;

  const size_t first = (size_t)blockIdx.x * 4 + wave;
  const bool has_ps = a.pscaler != nullptr, has_cs = (KIND == EDGE_II && a.cscaler != nullptr);
  const unsigned int * psp = has_ps ? a.pscaler : a.zero;
  const unsigned int * csp = has_cs ? a.cscaler : a.zero;
  const int * invp = a.invariant ? a.invariant : reinterpret_cast<const int *>(a.zero);
  const bool has_inv = a.invariant != nullptr;
  const double * const first_clv = (KIND == EDGE_II) ? a.child : a.parent;
  unsigned int w_next = 0, code_next = 0, sc_next = 0;
  int inv_next = -1;
  // per-site words first, then the first chunk's first operand (loads return in order)
  auto request_tile = [&](size_t tile) {
    const size_t n = tile * 16 + s;
    w_next = a.pattern_weights[n];
    inv_next = invp[has_inv ? n : 0];
    if (KIND == EDGE_TI) code_next = a.tip[n];
    sc_next = per_rate ? 0u : psp[has_ps ? n : 0] + csp[has_cs ? n : 0];
    dma_tile<RC, NT>(first_clv, tile * 16, toff, region, RT);
  };
  if (first < tiles) request_tile(first);
  for (size_t tile = first; tile < tiles; tile += nwaves)
  {
    const size_t site0 = tile * 16, n = site0 + s;
    const size_t next = tile + nwaves;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned int w_cur = w_next, code = code_next, site_scalings = sc_next;
    int inv_cur = has_inv ? inv_next : -1;
    asm volatile("" : "+v"(w_cur), "+v"(code), "+v"(inv_cur), "+v"(site_scalings));
    if (KIND == EDGE_TI && code >= a.maxstates) code = 0;
    if (per_rate)
    {
      // the site's smallest count over ALL categories comes first (core_likelihood.c:320-331); rare mode: plain loads
      unsigned int mn = 0xffffffffu;
      for (unsigned int k = 0; k < RT; ++k)
      {
        const unsigned int v = psp[has_ps ? n * RT + k : 0] + csp[has_cs ? n * RT + k : 0];
        mn = v < mn ? v : mn;
      }
      site_scalings = mn;
    }
    double terma = 0.0;
    for (unsigned int h = 0; h < H; ++h)
    {
      double b[RC][5], x[RC][5], pi[RC][5];
      unsigned int rel[RC];
#pragma unroll
      for (int k = 0; k < RC; ++k)
      {
        rel[k] = 0u;
        if (per_rate)
        {
          const size_t e = n * RT + h * RC + k;
          const unsigned int d = psp[has_ps ? e : 0] + csp[has_cs ? e : 0] - site_scalings;
          rel[k] = d > PLLHIP_SCALE_RATE_MAXDIFF ? PLLHIP_SCALE_RATE_MAXDIFF : d;
        }
#pragma unroll
        for (int g = 0; g < 5; ++g) pi[k][g] = s_freqs[(h * RC + k) * 20 + 4 * g + q];
      }
      if (h > 0 || per_rate) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (KIND == EDGE_II)
      {
        read_b_operands<RC>(region, s, q, b);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        dma_tile<RC, NT>(a.parent + h * RC * 20, site0, toff, region, RT);
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
      if (h + 1 < H) dma_tile<RC, NT>(first_clv + (h + 1) * RC * 20, site0, toff, region, RT); // the tile's next chunk
      else if (next < tiles) request_tile(next);
#pragma unroll
      for (int k = 0; k < RC; ++k)
      {
        double t = 0.0;
#pragma unroll
        for (int g = 0; g < 5; ++g) t += x[k][g] * b[k][g];
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        // category term -> weighted contribution (core_likelihood_avx2.c:480-500); lanes 0..15 own a site each
        if (rel[k] > 0) t *= scale_minlh(rel[k]);
        const double pinv = s_model[2 * (h * RC + k)];
        const double w = s_model[2 * (h * RC + k) + 1];
        if (pinv > 0.0)
        {
          const double inv_lk = (inv_cur == -1) ? 0.0 : s_freqs[(h * RC + k) * 20 + inv_cur];
          terma += w * (t * (1.0 - pinv) + inv_lk * pinv);
        }
        else
          terma += t * w;
      }
    }
    if (q == 0 && n < sites)
    {
      double lk = log(terma);
      if (site_scalings) lk += (double)site_scalings * log(PLLHIP_SCALE_THRESHOLD);
      lk *= (double)w_cur;
      if (a.persite) a.persite[n] = lk;
      acc += lk;
    }
  }
  block_sum_to_partials(acc, a.reduce);
}

template <int RC>
static int launch_lnl_chunks(pllhip_ctx * c, LnlArgs & a, int kind, unsigned int * grid_out)
{
  using G = aa_geom<RC>;
  if (a.pidx || a.cidx) return 1;
  const unsigned int RT = a.rate_cats;
  const size_t tiles = ((size_t)a.sites + 15) / 16;
  size_t blocks = (tiles + 3) / 4;
  const size_t head = (kind == EDGE_II) ? (size_t)RT * 400 : (kind == EDGE_TI ? (size_t)a.maxstates * RT * 20 : 0);
  const size_t lds = (head + (size_t)RT * 22) * sizeof(double) + 4 * (size_t)G::REGION_B;
  if (lds > 150 * 1024) return 1;
  // (two workgroups per CU while their LDS allows)
  const size_t per_cu = lds <= 80 * 1024 ? 2 : 1;
  const size_t cap = pllhip_env("PLLHIP_AA_GRID_CAP") ? (size_t)atoi(pllhip_env("PLLHIP_AA_GRID_CAP")) : (size_t)c->num_cus * per_cu;
  if (blocks > cap) blocks = cap;
  const bool nt = pllhip_use_nt(c);
  const dim3 grid((unsigned int)blocks), block(256);
  a.reduce = pllhip_reduce_out(c, (unsigned int)blocks);
#define LNL_ONE(KERNEL)                                                                        \
  do {                                                                                         \
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&KERNEL),                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));        \
    hipLaunchKernelGGL(KERNEL, grid, block, lds, c->stream, a);                                \
  } while (0)
#define LNL_KIND(KINDV)                                          \
  do {                                                           \
    if (nt) LNL_ONE((k_lnl_aa_chunks<RC, KINDV, true>));         \
    else LNL_ONE((k_lnl_aa_chunks<RC, KINDV, false>));           \
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
    default: break;
  }
  if (!pllhip_aa_chunks_enabled()) return 1;
  if (a.rate_cats % 4 == 0) return launch_lnl_chunks<4>(c, a, kind, grid_out);
  if (a.rate_cats % 2 == 0) return launch_lnl_chunks<2>(c, a, kind, grid_out);
  return launch_lnl_chunks<1>(c, a, kind, grid_out);
}
