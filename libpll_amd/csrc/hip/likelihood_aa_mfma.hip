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
#include "aa_mfma.hpp"
#include "lnl_common.hpp"

template <int RC, int KIND, bool NT>
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
  __syncthreads();

  const unsigned int lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned int s = lane & 15u, q = lane >> 4;
  char * region = reinterpret_cast<char *>(smem + head) + wave * G::REGION_B;

  // pi_k[4g+q] for this lane
  double pi[RC][5];
#pragma unroll
  for (int k = 0; k < RC; ++k)
#pragma unroll
    for (int g = 0; g < 5; ++g) pi[k][g] = a.freqs[(size_t)a.freqs_indices[k] * 20 + 4 * g + q];

  const size_t sites = a.sites;
  const size_t tiles = (sites + 15) / 16;
  const size_t nwaves = (size_t)gridDim.x * 4;
  const bool per_rate = a.rate_scalers && KIND != ROOT;
  double acc = 0.0;

  for (size_t tile = (size_t)blockIdx.x * 4 + wave; tile < tiles; tile += nwaves)
  {
    const size_t site0 = tile * 16;
    double b[RC][5], x[RC][5];
    if (KIND == EDGE_II)
    {
      dma_tile<RC, NT>(a.child, site0, sites, region, lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      read_b_operands<RC>(region, s, q, b);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      dma_tile<RC, NT>(a.parent, site0, sites, region, lane);
      tile_matvec<RC>(tab, b, lane, x);
#pragma unroll
      for (int k = 0; k < RC; ++k)
#pragma unroll
        for (int g = 0; g < 5; ++g) x[k][g] = x[k][g] * pi[k][g];
    }
    else
    {
      dma_tile<RC, NT>(a.parent, site0, sites, region, lane);
      unsigned int code = 0;
      if (KIND == EDGE_TI)
      {
        code = (site0 + s < sites) ? a.tip[site0 + s] : 0u;
        if (code >= a.maxstates) code = 0;
      }
#pragma unroll
      for (int k = 0; k < RC; ++k)
#pragma unroll
        for (int g = 0; g < 5; ++g)
          x[k][g] = (KIND == EDGE_TI) ? tab[(code * RC + k) * 20 + 4 * g + q] : pi[k][g];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    read_b_operands<RC>(region, s, q, b); // parent CLV, states 4g+q
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

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
          unsigned int v = a.pscaler ? a.pscaler[n * RC + k] : 0u;
          if (KIND == EDGE_II && a.cscaler) v += a.cscaler[n * RC + k];
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
        if (a.pscaler) site_scalings += a.pscaler[n];
        if (KIND == EDGE_II && a.cscaler) site_scalings += a.cscaler[n];
      }
      double terma = 0.0;
#pragma unroll
      for (int k = 0; k < RC; ++k) terma += category_term<false>(a, term[k], (unsigned int)k, n, rel[k]);
      acc += site_loglk(a, terma, n, site_scalings);
    }
    // the image is refilled by the next tile's DMA
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  block_sum_to_partials(acc, a.reduce);
}

template <int RC>
static int launch_lnl_rc(pllhip_ctx * c, LnlArgs & a, int kind, unsigned int * grid_out)
{
  using G = aa_geom<RC>;
  const size_t tiles = ((size_t)a.sites + 15) / 16;
  size_t blocks = (tiles + 3) / 4;
  const size_t cap = (size_t)c->num_cus * 2;
  if (blocks > cap) blocks = cap;
  const size_t head = (kind == EDGE_II) ? (size_t)RC * 400
                                        : (kind == EDGE_TI ? (size_t)a.maxstates * RC * 20 : 0);
  const size_t lds = head * sizeof(double) + 4 * (size_t)G::REGION_B;
  if (lds > 80 * 1024) return 1;
  const bool nt = pllhip_use_nt(c);
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
    if (nt) LNL_ONE((k_lnl_aa_mfma<RC, KINDV, true>));                      \
    else LNL_ONE((k_lnl_aa_mfma<RC, KINDV, false>));                        \
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
    default: return 1;
  }
}
