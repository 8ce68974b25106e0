// derivatives.hip -- branch-length derivative machinery.
//
// (1) sumtable   sum[n,k,j] = (sum_m pi_m Vinv[m,j] Pclv[n,k,m]) * (sum_m V[j,m] Cclv[n,k,m])
//     replaces pll_core_update_sumtable_ii / _ti (core_derivatives.c:125,277;
//     AVX2-flag kernels core_derivatives_avx.c:25,462, core_derivatives_avx2.c:24,274).
//     Observation: with A_k[j][m] = pi_m Vinv_k[m][j] and B_k = V_k this is exactly
//     the CLV update (A.P) (.) (B.C) without scaling, so the sumtable is produced by
//     the SAME streaming kernels as pll_update_partials (partials.hip), fed with
//     two small per-category matrices built by k_build_sumtable_mats.  The table
//     stays in HBM; 384 B per site for 4x4.
// (2) derivatives  per site (L, L', L'') = sum_k w_k sum_j sum[n,k,j] * diagp[k,j,0..2]
//     d = sum_n w_n (-L'/L),  dd = sum_n w_n ((L'/L)^2 - L''/L)
//     replaces the site loop of pll_core_likelihood_derivatives
//     (core_derivatives.c:501, AVX2 kernel core_derivatives_avx2.c:523).
//     Streams the table once per Newton iteration: 132 B per site for 4x4.
//     The (rate_cats x states x 4) diagptable is built by the host with libm exp,
//     exactly like core_derivatives.c:560-575, and passed in.
//     Kernels: k_derivatives_dna (4 states, 16 bytes per lane, the table itself a kernel
//     argument), k_derivatives_aa_tile (20 states, LDS-DMA tiles), k_derivatives_rows (lane per
//     (site, rate) row, every other shape), k_derivatives_gen (tables beyond 64 KB of LDS).
//     The ascertainment-bias terms come from asc_bias.hip and are added by the final sum.
#include "ctx.hpp"
#include "numerics.hpp"
#include "aa_mfma.hpp"
#include "lnl_common.hpp"

struct SumMatArgs
{
  double * left;   // [R][S][S]  A_k[j][m] = freqs_k[m] * inv_eigenvecs_k[m][j]
  double * right;  // [R][S][S]  B_k[j][m] = eigenvecs_k[j][m]
  const double * eigenvecs;
  const double * inv_eigenvecs;
  const double * freqs;
  unsigned int states, rate_cats;
  unsigned int params_indices[PLLHIP_MAX_RATE_CATS];
};

__global__ void k_build_sumtable_mats(SumMatArgs a)
{
  const unsigned int S = a.states;
  const unsigned int total = a.rate_cats * S * S;
  for (unsigned int t = blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += gridDim.x * blockDim.x)
  {
    const unsigned int k = t / (S * S), j = (t / S) % S, m = t % S;
    const size_t pi = a.params_indices[k];
    a.left[t] = a.freqs[pi * S + m] * a.inv_eigenvecs[pi * S * S + m * S + j];
    a.right[t] = a.eigenvecs[pi * S * S + j * S + m];
  }
}

// per-rate scaling mode: bring every category of a site to the site's minimum
// scaler, capped at 4 steps (core_derivatives.c:100-118,187-191)
__global__ __launch_bounds__(256) void k_sumtable_rescale(double * __restrict__ sum,
                                                          const unsigned int * __restrict__ ps,
                                                          const unsigned int * __restrict__ cs,
                                                          const unsigned int * __restrict__ pmap,
                                                          const unsigned int * __restrict__ cmap,
                                                          unsigned int sites, unsigned int R,
                                                          unsigned int S)
{
  for (size_t n = blockIdx.x * (size_t)blockDim.x + threadIdx.x; n < sites;
       n += (size_t)gridDim.x * blockDim.x)
  {
    // (site repeats: scale buffers stored by class are read through the site -> row maps)
    const size_t np = pmap ? pmap[n] : n, nc = cmap ? cmap[n] : n;
    unsigned int mn = 0xffffffffu;
    for (unsigned int k = 0; k < R; ++k)
    {
      unsigned int v = (ps ? ps[np * R + k] : 0) + (cs ? cs[nc * R + k] : 0);
      mn = v < mn ? v : mn;
    }
    for (unsigned int k = 0; k < R; ++k)
    {
      unsigned int d = (ps ? ps[np * R + k] : 0) + (cs ? cs[nc * R + k] : 0) - mn;
      if (!d) continue;
      if (d > PLLHIP_SCALE_RATE_MAXDIFF) d = PLLHIP_SCALE_RATE_MAXDIFF;
      const double f = d == 1 ? 0x1p-256 : d == 2 ? 0x1p-512 : d == 3 ? 0x1p-768 : 0x1p-1024;
      for (unsigned int j = 0; j < S; ++j) sum[(n * R + k) * S + j] *= f;
    }
  }
}

extern "C" int pllhip_update_sumtable(pllhip_ctx_t * c, unsigned int parent_clv,
                                      int parent_scaler, unsigned int child_clv,
                                      int child_scaler, const unsigned int * h_params_indices,
                                      unsigned int slot)
{
  PLLHIP_ALL_SHARDS_PAR(c, pllhip_update_sumtable(s, parent_clv, parent_scaler, child_clv, child_scaler, h_params_indices, slot));
  HIP_TRY(hipSetDevice(c->sh.device));
  PLLHIP_CERT_FIRST(c);
  const unsigned int nodes = (unsigned int)c->clv.size();
  if (slot >= PLLHIP_SUMTABLE_MAX_SLOTS || parent_clv >= nodes || child_clv >= nodes ||
      parent_scaler >= (int)c->sh.scale_buffers || child_scaler >= (int)c->sh.scale_buffers)
  {
    pllhip_set_error("pllhip_update_sumtable: index out of range");
    return -1;
  }
  const bool tp = pllhip_is_tip(c, parent_clv), tc = pllhip_is_tip(c, child_clv);
  if (tp && tc)
  {
    // the reference asserts here (derivatives.c:191-195)
    pllhip_set_error("pllhip_update_sumtable: tip-tip edge has no sumtable");
    return -1;
  }
  if (!c->sumtable[slot])
    HIP_TRY(hipMalloc((void **)&c->sumtable[slot], (c->clv_elems + PLLHIP_TAIL_SITES * c->span) * sizeof(double)));

  const unsigned int S = c->sh.states, R = c->sh.rate_cats;
  // the two matrix sets live at the start of the staging buffer's device half
  if (2 * c->pmat_elems * sizeof(double) > c->stage_bytes)
  {
    pllhip_set_error("pllhip_update_sumtable: staging buffer too small");
    return -1;
  }
  SumMatArgs m;
  m.left = (double *)c->d_stage;
  m.right = m.left + c->pmat_elems;
  m.eigenvecs = c->eigenvecs;
  m.inv_eigenvecs = c->inv_eigenvecs;
  m.freqs = c->freqs;
  m.states = S;
  m.rate_cats = R;
  for (unsigned int k = 0; k < R; ++k)
  {
    if (h_params_indices[k] >= c->sh.rate_matrices)
    {
      pllhip_set_error("pllhip_update_sumtable: params index out of range");
      return -1;
    }
    m.params_indices[k] = h_params_indices[k];
  }
  k_build_sumtable_mats<<<(R * S * S + 255) / 256, 256, 0, c->stream>>>(m);
  HIP_TRY(hipGetLastError());

  PartialsArgs a;
  memset(&a, 0, sizeof(a));
  a.parent = c->sumtable[slot];
  a.tipmap = c->tipmap;
  a.zero = c->d_zero;
  a.sites = c->sh.sites;
  a.rate_cats = R;
  a.states = S;
  a.maxstates = c->maxstates;
  const unsigned int * ps = nullptr, * cs = nullptr;
  int kind;
  if (tp || tc)
  {
    // the tip supplies the pi-weighted "left" factor (core_derivatives.c:413-429);
    // its matrix rows are A_k when the tip is the parent side.  When the tip is
    // the CHILD, the reference still takes the left factor from the tip and the
    // eigenvector factor from the inner CLV (derivatives.c:67-78).
    kind = 1;
    a.ltip = pllhip_tip_ptr(c, tp ? parent_clv : child_clv);
    a.right = c->clv[tp ? child_clv : parent_clv];
    a.lmat = m.left;
    a.rmat = m.right;
    ps = pllhip_scaler_ptr(c, tp ? child_scaler : parent_scaler);
  }
  else
  {
    kind = 0;
    a.left = c->clv[parent_clv];
    a.right = c->clv[child_clv];
    a.lmat = m.left;
    a.rmat = m.right;
    ps = pllhip_scaler_ptr(c, parent_scaler);
    cs = pllhip_scaler_ptr(c, child_scaler);
  }
  if (!a.right || (kind == 0 && !a.left))
  {
    pllhip_set_error("pllhip_update_sumtable: CLV missing");
    return -1;
  }
  // site repeats: the table is per site, a CLV stored by class is read through its
  // site -> row map (and so is the scale buffer that goes with it)
  const unsigned int * pmap = nullptr, * cmap = nullptr;
  if (!c->rows.empty())
  {
    if (kind == 0)
    {
      if (c->rows[parent_clv].classes) pmap = c->rows[parent_clv].site_id;
      if (c->rows[child_clv].classes) cmap = c->rows[child_clv].site_id;
      a.lidx = pmap;
      a.ridx = cmap;
    }
    else
    {
      const unsigned int inner = tp ? child_clv : parent_clv;
      if (c->rows[inner].classes) pmap = c->rows[inner].site_id;
      a.ridx = pmap; // the inner node is the right operand; its scaler is `ps`
    }
  }
  int rc = pllhip_launch_partials(c, a, kind, SCALE_NONE, PLLHIP_PROF_SUMTABLE);
  if (rc) return rc;
  if (c->sh.rate_scalers && (ps || cs))
  {
    k_sumtable_rescale<<<pllhip_stream_grid(c, a.sites, 256), 256, 0, c->stream>>>(
        c->sumtable[slot], ps, cs, pmap, cmap, a.sites, R, S);
    HIP_TRY(hipGetLastError());
  }
  return 0;
}

struct DerivArgs
{
  const double * __restrict__ sumtable;
  const double * __restrict__ diagp;      // device copy [R][S][4]
  const double * __restrict__ rate_weights;
  const double * __restrict__ freqs;
  const double * __restrict__ prop_invar;
  const unsigned int * __restrict__ pattern_weights;
  const int * __restrict__ invariant;
  ReduceOut reduce;
  unsigned int sites, rate_cats, states;
  unsigned int params_indices[PLLHIP_MAX_RATE_CATS];
  // 4-state data (rate_cats <= 8: 1 KB) and 20-state data on the tile kernel (rate_cats
  // <= 4: 2.5 KB): the table itself travels as a kernel argument, which saves the
  // staging copy and the stream drain in front of it
  double diag_inline[4 * 20 * 4];
};

__device__ __forceinline__ void block_sum2(double v0, double v1, const ReduceOut & ro)
{
  const double two[2] = {v0, v1};
  grid_sum<2>(two, ro);
}

// Any state count (SC = compile-time count up to 16, 0 = run-time count) and any rate_cats
// up to 64: one lane per (site, rate) row of the table, a wave takes 64 sites at a time
// in rounds of 64 / R whole sites; after round r the lanes [r * 64/R, (r+1) * 64/R) keep
// that round's site sums, so that the two divisions of the site tail run once per site,
// on all lanes (the mapping of k_lnl_rows, likelihood.hip).
template <int SC>
__global__ __launch_bounds__(256) void k_derivatives_rows(DerivArgs a)
{
  extern __shared__ double s_diag[]; // [R][S * 4 + 2]: the two extra words skew the banks between rates
  const unsigned int S = SC ? (unsigned int)SC : a.states, R = a.rate_cats, tid = threadIdx.x;
  const unsigned int DP = S * 4u + 2u;
  for (unsigned int t = tid; t < R * S * 4u; t += 256u) s_diag[(t / (S * 4u)) * DP + t % (S * 4u)] = a.diagp[t];
  __syncthreads();
  const unsigned int lane = tid & 63u;
  const unsigned int spr = 64u / R, nrounds = (64u + spr - 1u) / spr;
  const unsigned int g = lane / R, k = lane - g * R, grp0 = g * R;
  const unsigned int own_round = lane / spr;
  const int own_src = (int)((lane - own_round * spr) * R);
  const unsigned int pi = a.params_indices[k];
  const double pinv = a.prop_invar[pi];
  const double w = a.rate_weights[k];
  const size_t sites_up = ((size_t)a.sites + 63) & ~(size_t)63;
  double acc_d = 0.0, acc_dd = 0.0;
  for (size_t sbase = ((size_t)blockIdx.x * 4u + (tid >> 6)) * 64u; sbase < sites_up;
       sbase += (size_t)gridDim.x * 256u)
  {
    double o0 = 1.0, o1 = 0.0, o2 = 0.0;
    for (unsigned int round = 0; round < nrounds; ++round)
    {
      unsigned int koff = k * DP; // pinned: the table reads are loop-invariant and would be hoisted
      asm volatile("" : "+v"(koff));
      const double * dg = s_diag + koff;
      const unsigned int pos = round * spr + g;
      const bool act = g < spr && pos < 64u && sbase + pos < a.sites;
      const size_t n = act ? sbase + pos : 0;
      const double * sm = a.sumtable + (n * R + k) * S;
      double c0 = 0.0, c1 = 0.0, c2 = 0.0;
      if (SC)
      {
        double v[SC ? SC : 1];
#pragma unroll
        for (int j = 0; j < SC; ++j) v[j] = sm[j];
#pragma unroll
        for (int j = 0; j < SC; ++j)
        {
          c0 = fma(v[j], dg[j * 4 + 0], c0);
          c1 = fma(v[j], dg[j * 4 + 1], c1);
          c2 = fma(v[j], dg[j * 4 + 2], c2);
        }
      }
      else
      {
#pragma unroll 4
        for (unsigned int j = 0; j < S; ++j)
        {
          const double v = sm[j];
          c0 = fma(v, dg[j * 4 + 0], c0);
          c1 = fma(v, dg[j * 4 + 1], c1);
          c2 = fma(v, dg[j * 4 + 2], c2);
        }
      }
      if (pinv > 0.0)
      {
        // core_derivatives.c:481-491
        const int inv = a.invariant ? a.invariant[n] : -1;
        const double inv_lk = (inv == -1) ? 0.0 : a.freqs[(size_t)pi * S + inv] * pinv;
        c0 = c0 * (1.0 - pinv) + inv_lk;
        c1 = c1 * (1.0 - pinv);
        c2 = c2 * (1.0 - pinv);
      }
      c0 *= w; c1 *= w; c2 *= w;
      double l0 = 0.0, l1 = 0.0, l2 = 0.0;
      for (unsigned int i = 0; i < R; ++i)
      {
        l0 += __shfl(c0, (int)(grp0 + i), 64);
        l1 += __shfl(c1, (int)(grp0 + i), 64);
        l2 += __shfl(c2, (int)(grp0 + i), 64);
      }
      const double t0 = __shfl(l0, own_src, 64), t1 = __shfl(l1, own_src, 64), t2 = __shfl(l2, own_src, 64);
      if (own_round == round)
      {
        o0 = t0;
        o1 = t1;
        o2 = t2;
      }
    }
    if (sbase + lane < a.sites)
    {
      const double d1 = -o1 / o0;
      const double d2 = d1 * d1 - o2 / o0;
      const double pw = (double)a.pattern_weights[sbase + lane];
      acc_d += pw * d1;
      acc_dd += pw * d2;
    }
  }
  block_sum2(acc_d, acc_dd, a.reduce);
}

// 20 states: the table is walked like a CLV in partials_aa_mfma.hip -- a wave owns 16
// sites x RC rates, one contiguous block that LDS-DMA copies into a padded per-wave
// image; lane (s, q) then takes states 4c+q of its site and forms its share of the
// three dot products per rate, two __shfl_xor add the four q-lanes of a site, and lanes
// 0..15 finish one site each.  The next tile and its per-site words are requested as
// soon as the image has been read.  (One lane per (site, rate) reading 160 contiguous
// bytes of its own reached 2.3 TB/s on this layout; this one is
// bound by the DMA stream.)
template <int RC, bool NT>
__global__ __launch_bounds__(256) void k_derivatives_aa_tile(DerivArgs a)
{
  using G = aa_geom<RC>;
  extern __shared__ double smem[]; // [diag RC x 20 x 4][4 images]
  __shared__ double s_model[RC][2]; // prop_invar, rate weight of the category
  __shared__ double s_freqs[RC][20];
  double * s_diag = smem;
  for (unsigned int t = threadIdx.x; t < RC * 80u; t += blockDim.x) s_diag[t] = a.diag_inline[t];
  for (unsigned int t = threadIdx.x; t < RC * 20u; t += blockDim.x)
    s_freqs[t / 20u][t % 20u] = a.freqs[(size_t)a.params_indices[t / 20u] * 20 + t % 20u];
  if (threadIdx.x < RC)
  {
    s_model[threadIdx.x][0] = a.prop_invar[a.params_indices[threadIdx.x]];
    s_model[threadIdx.x][1] = a.rate_weights[threadIdx.x];
  }
  __syncthreads();

  const unsigned int lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned int s = lane & 15u, q = lane >> 4;
  char * region = reinterpret_cast<char *>(smem + RC * 80) + wave * G::REGION_B;
  unsigned int toff[G::N_IT];
  tile_offsets<RC>(lane, toff);

  const size_t sites = a.sites;
  const size_t tiles = (sites + 15) / 16;
  const size_t nwaves = (size_t)gridDim.x * 4;
  const size_t first = (size_t)blockIdx.x * 4 + wave;
  // (absent array: any valid word will do, the value is replaced by -1 below)
  const int * invp = a.invariant ? a.invariant : reinterpret_cast<const int *>(a.pattern_weights);
  const bool has_inv = a.invariant != nullptr;
  double acc_d = 0.0, acc_dd = 0.0;
  unsigned int w_next = 0;
  int inv_next = -1;
  if (first < tiles)
  {
    w_next = a.pattern_weights[first * 16 + s];
    inv_next = invp[has_inv ? first * 16 + s : 0];
    dma_tile<RC, NT>(a.sumtable, first * 16, toff, region);
  }
  for (size_t tile = first; tile < tiles; tile += nwaves)
  {
    const size_t next = tile + nwaves;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned int w_cur = w_next;
    int inv_cur = has_inv ? inv_next : -1;
    asm volatile("" : "+v"(w_cur), "+v"(inv_cur));
    double b[RC][5];
    read_b_operands<RC>(region, s, q, b);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (next < tiles)
    {
      w_next = a.pattern_weights[next * 16 + s];
      inv_next = invp[has_inv ? next * 16 + s : 0];
      dma_tile<RC, NT>(a.sumtable, next * 16, toff, region);
    }
    double l0 = 0.0, l1 = 0.0, l2 = 0.0;
#pragma unroll
    for (int k = 0; k < RC; ++k)
    {
      double c0 = 0.0, c1 = 0.0, c2 = 0.0;
#pragma unroll
      for (int c = 0; c < 5; ++c)
      {
        const double * dg = s_diag + (k * 20 + 4 * c + (int)q) * 4;
        c0 = fma(b[k][c], dg[0], c0);
        c1 = fma(b[k][c], dg[1], c1);
        c2 = fma(b[k][c], dg[2], c2);
      }
      c0 += __shfl_xor(c0, 16, 64);
      c1 += __shfl_xor(c1, 16, 64);
      c2 += __shfl_xor(c2, 16, 64);
      c0 += __shfl_xor(c0, 32, 64);
      c1 += __shfl_xor(c1, 32, 64);
      c2 += __shfl_xor(c2, 32, 64);
      const double pinv = s_model[k][0], w = s_model[k][1];
      if (pinv > 0.0)
      {
        // core_derivatives.c:481-491
        const double inv_lk = (inv_cur == -1) ? 0.0 : s_freqs[k][inv_cur] * pinv;
        c0 = c0 * (1.0 - pinv) + inv_lk;
        c1 = c1 * (1.0 - pinv);
        c2 = c2 * (1.0 - pinv);
      }
      l0 += c0 * w;
      l1 += c1 * w;
      l2 += c2 * w;
    }
    if (q == 0 && tile * 16 + s < sites)
    {
      const double d1 = -l1 / l0;
      const double d2 = d1 * d1 - l2 / l0;
      const double pw = (double)w_cur;
      acc_d += pw * d1;
      acc_dd += pw * d2;
    }
  }
  block_sum2(acc_d, acc_dd, a.reduce);
}

// 4 states: one lane per 16 bytes of the sumtable (two states), 2*RC lanes per
// site, waves in rounds of 64 sites like the CLV and lnL kernels: every load
// instruction of a wave is one contiguous KiB and, after a round, each lane
// finishes ONE site (two f64 divisions per site instead of per lane).
template <int RC, bool NT>
__global__ __launch_bounds__(256) void k_derivatives_dna(DerivArgs a)
{
  constexpr unsigned int W = 2 * RC, SPS = 64 / W;
  const unsigned int lane = threadIdx.x & 63u;
  const unsigned int h = lane & 1u, k = (lane >> 1) & (RC - 1);
  // diagp[k][j][0..2] for this lane's two states j = 2h, 2h+1
  double dg[2][3];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj)
#pragma unroll
    for (int t = 0; t < 3; ++t) dg[jj][t] = a.diag_inline[(k * 4 + 2 * h + jj) * 4 + t];
  const unsigned int pi = a.params_indices[k];
  const double pinv = a.prop_invar[pi];
  const double wk = a.rate_weights[k];

  double acc_d = 0.0, acc_dd = 0.0;
  const size_t sites = a.sites;
  const size_t rounds = (sites + 63) / 64;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  const double2 * __restrict__ ST = reinterpret_cast<const double2 *>(a.sumtable);
  // (absent array: any valid word will do, the value is replaced by -1 below)
  const int * inv_site = a.invariant ? a.invariant : reinterpret_cast<const int *>(a.pattern_weights);
  const bool has_inv = a.invariant != nullptr;
  for (size_t r = wave; r < rounds; r += nwaves)
  {
    double m0 = 1.0, m1 = 0.0, m2 = 0.0;
    // Everything the round needs is requested up front, unclamped (the table, the
    // weights and the invariant array carry PLLHIP_TAIL_SITES of slack): W KiB of table
    // plus the weight of the one site the lane finishes.
    // (Round 5 measured TWO rounds in flight per wave -- the next round's W KiB requested before this round's
    // arithmetic: 15.4 against 14.5 us per launch on BASELINE config 5's table, profiles/r5_result_calls_ab.txt; the
    // copies and the registers cost more than the overlap gains with ~4 rounds per wave.  Not kept.)
    const size_t n_own = r * 64 + (size_t)(lane & (W - 1)) * SPS + lane / W;
    const unsigned int pw_own = a.pattern_weights[n_own];
    const int inv_raw = inv_site[has_inv ? n_own : 0];
    const int inv_own = has_inv ? inv_raw : -1;
    const unsigned int grp0 = lane & ~(W - 1);
    double2 sv[W];
#pragma unroll
    for (unsigned int j = 0; j < W; ++j) sv[j] = ld16<NT>(ST + (r * 64 + (size_t)j * SPS) * W + lane);
#pragma unroll
    for (unsigned int j = 0; j < W; ++j)
    {
      const double2 s = sv[j];
      double c0 = fma(s.y, dg[1][0], s.x * dg[0][0]);
      double c1 = fma(s.y, dg[1][1], s.x * dg[0][1]);
      double c2 = fma(s.y, dg[1][2], s.x * dg[0][2]);
      // the other half of this (site, rate)
      c0 += dpp_pair_swap(c0);
      c1 += dpp_pair_swap(c1);
      c2 += dpp_pair_swap(c2);
      if (pinv > 0.0)
      {
        // core_derivatives.c:481-491
        const int inv = __shfl(inv_own, (int)(grp0 + j), 64); // held by the lane that owns the site
        const double inv_lk = (inv == -1) ? 0.0 : a.freqs[(size_t)pi * 4 + inv] * pinv;
        c0 = c0 * (1.0 - pinv) + inv_lk;
        c1 = c1 * (1.0 - pinv);
        c2 = c2 * (1.0 - pinv);
      }
      c0 *= wk; c1 *= wk; c2 *= wk;
      // sum the RC categories of the site (both lanes of a pair hold the category's
      // value): DPP butterflies, same association as xor-2 / xor-4 / xor-8 shuffles
      c0 = dpp_group_sum_pairs<W>(c0);
      c1 = dpp_group_sum_pairs<W>(c1);
      c2 = dpp_group_sum_pairs<W>(c2);
      if ((lane & (W - 1)) == j) { m0 = c0; m1 = c1; m2 = c2; }
    }
    if (n_own < sites)
    {
      const double d1 = -m1 / m0;
      const double d2 = d1 * d1 - m2 / m0;
      const double pw = (double)pw_own;
      acc_d += pw * d1;
      acc_dd += pw * d2;
    }
  }
  block_sum2(acc_d, acc_dd, a.reduce);
}

// any rate_cats: one lane per site
__global__ __launch_bounds__(128) void k_derivatives_gen(DerivArgs a)
{
  const unsigned int S = a.states, R = a.rate_cats;
  double acc_d = 0.0, acc_dd = 0.0;
  for (size_t n = blockIdx.x * (size_t)blockDim.x + threadIdx.x; n < a.sites;
       n += (size_t)gridDim.x * blockDim.x)
  {
    double l0 = 0.0, l1 = 0.0, l2 = 0.0;
    for (unsigned int k = 0; k < R; ++k)
    {
      const double * sm = a.sumtable + (n * R + k) * S;
      const double * dg = a.diagp + (size_t)k * S * 4;
      double c0 = 0.0, c1 = 0.0, c2 = 0.0;
      for (unsigned int j = 0; j < S; ++j)
      {
        c0 += sm[j] * dg[j * 4 + 0];
        c1 += sm[j] * dg[j * 4 + 1];
        c2 += sm[j] * dg[j * 4 + 2];
      }
      const unsigned int pi = a.params_indices[k];
      const double pinv = a.prop_invar[pi];
      if (pinv > 0.0)
      {
        const int inv = a.invariant ? a.invariant[n] : -1;
        const double inv_lk = (inv == -1) ? 0.0 : a.freqs[(size_t)pi * S + inv] * pinv;
        c0 = c0 * (1.0 - pinv) + inv_lk;
        c1 = c1 * (1.0 - pinv);
        c2 = c2 * (1.0 - pinv);
      }
      l0 += c0 * a.rate_weights[k];
      l1 += c1 * a.rate_weights[k];
      l2 += c2 * a.rate_weights[k];
    }
    const double d1 = -l1 / l0;
    const double d2 = d1 * d1 - l2 / l0;
    const double pw = (double)a.pattern_weights[n];
    acc_d += pw * d1;
    acc_dd += pw * d2;
  }
  block_sum2(acc_d, acc_dd, a.reduce);
}

// 20 states, category counts other than 1, 2, 4 (round 4): the same walk with a tile's categories in CHUNKS of RC
// (the largest of 4, 2, 1 dividing the count; k_lnl_aa_chunks in likelihood_aa_mfma.hip is the lnL's): rate_cats / RC
// DMAs of RC x 160 bytes per site, the three sums of a site accumulated chunk by chunk in category order.  The table of
// exponentials (rate_cats x 80 doubles) comes from the staged device copy, not from the kernel arguments.
template <int RC, bool NT>
__global__ __launch_bounds__(256) void k_derivatives_aa_chunks(DerivArgs a)
{
  using G = aa_geom<RC>;
  extern __shared__ double smem[]; // [diag RT x 20 x 4][4 images][freqs RT x 20][model RT x 2]
  const unsigned int RT = a.rate_cats, H = RT / RC;
  double * s_diag = smem;
  double * s_freqs = smem + RT * 80u + 4 * (G::REGION_B / 8);
  double * s_model = s_freqs + RT * 20u;
  for (unsigned int t = threadIdx.x; t < RT * 80u; t += blockDim.x) s_diag[t] = a.diagp[t];
  for (unsigned int t = threadIdx.x; t < RT * 20u; t += blockDim.x)
    s_freqs[t] = a.freqs[(size_t)a.params_indices[t / 20u] * 20 + t % 20u];
  for (unsigned int t = threadIdx.x; t < RT; t += blockDim.x)
  {
    s_model[2 * t] = a.prop_invar[a.params_indices[t]];
    s_model[2 * t + 1] = a.rate_weights[t];
  }
  __syncthreads();

  const unsigned int lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned int s = lane & 15u, q = lane >> 4;
  char * region = reinterpret_cast<char *>(smem + RT * 80u) + wave * G::REGION_B;
  unsigned int toff[G::N_IT];
  tile_offsets<RC>(lane, toff, RT);

  const size_t sites = a.sites;
  const size_t tiles = (sites + 15) / 16;
  const size_t nwaves = (size_t)gridDim.x * 4;
  const size_t first = (size_t)blockIdx.x * 4 + wave;
  const int * invp = a.invariant ? a.invariant : reinterpret_cast<const int *>(a.pattern_weights);
  const bool has_inv = a.invariant != nullptr;
  double acc_d = 0.0, acc_dd = 0.0;
  unsigned int w_next = 0;
  int inv_next = -1;
  if (first < tiles)
  {
    w_next = a.pattern_weights[first * 16 + s];
    inv_next = invp[has_inv ? first * 16 + s : 0];
    dma_tile<RC, NT>(a.sumtable, first * 16, toff, region, RT);
  }
  for (size_t tile = first; tile < tiles; tile += nwaves)
  {
    const size_t next = tile + nwaves;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned int w_cur = w_next;
    int inv_cur = has_inv ? inv_next : -1;
    asm volatile("" : "+v"(w_cur), "+v"(inv_cur));
    double l0 = 0.0, l1 = 0.0, l2 = 0.0;
    for (unsigned int h = 0; h < H; ++h)
    {
      if (h > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      double b[RC][5];
      read_b_operands<RC>(region, s, q, b);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (h + 1 < H) dma_tile<RC, NT>(a.sumtable + (h + 1) * RC * 20, tile * 16, toff, region, RT);
      else if (next < tiles)
      {
        w_next = a.pattern_weights[next * 16 + s];
        inv_next = invp[has_inv ? next * 16 + s : 0];
        dma_tile<RC, NT>(a.sumtable, next * 16, toff, region, RT);
      }
#pragma unroll
      for (int k = 0; k < RC; ++k)
      {
        const unsigned int kk = h * RC + k;
        double c0 = 0.0, c1 = 0.0, c2 = 0.0;
#pragma unroll
        for (int c = 0; c < 5; ++c)
        {
          const double * dg = s_diag + (kk * 20 + 4 * c + q) * 4;
          c0 = fma(b[k][c], dg[0], c0);
          c1 = fma(b[k][c], dg[1], c1);
          c2 = fma(b[k][c], dg[2], c2);
        }
        c0 += __shfl_xor(c0, 16, 64);
        c1 += __shfl_xor(c1, 16, 64);
        c2 += __shfl_xor(c2, 16, 64);
        c0 += __shfl_xor(c0, 32, 64);
        c1 += __shfl_xor(c1, 32, 64);
        c2 += __shfl_xor(c2, 32, 64);
        const double pinv = s_model[2 * kk], w = s_model[2 * kk + 1];
        if (pinv > 0.0)
        {
          // core_derivatives.c:481-491
          const double inv_lk = (inv_cur == -1) ? 0.0 : s_freqs[kk * 20 + inv_cur] * pinv;
          c0 = c0 * (1.0 - pinv) + inv_lk;
          c1 = c1 * (1.0 - pinv);
          c2 = c2 * (1.0 - pinv);
        }
        l0 += c0 * w;
        l1 += c1 * w;
        l2 += c2 * w;
      }
    }
    if (q == 0 && tile * 16 + s < sites)
    {
      const double d1 = -l1 / l0;
      const double d2 = d1 * d1 - l2 / l0;
      const double pw = (double)w_cur;
      acc_d += pw * d1;
      acc_dd += pw * d2;
    }
  }
  block_sum2(acc_d, acc_dd, a.reduce);
}

extern "C" int pllhip_likelihood_derivatives(pllhip_ctx_t * c, unsigned int slot,
                                             int parent_scaler, int child_scaler,
                                             const unsigned int * h_params_indices,
                                             const double * h_diagptable, double * h_d_f,
                                             double * h_dd_f)
{
  if (!c->shards.empty())
    return pllhip_group_likelihood_derivatives(c, slot, parent_scaler, child_scaler, h_params_indices, h_diagptable, h_d_f, h_dd_f);
  HIP_TRY(hipSetDevice(c->sh.device));
  if (!c->defer) PLLHIP_CERT_FIRST(c); // (reads scaler counts; the sumtable call before it has normally looked already.  A shard of a group: shard.hip looks)
  if (slot >= PLLHIP_SUMTABLE_MAX_SLOTS || !c->sumtable[slot])
  {
    pllhip_set_error("pllhip_likelihood_derivatives: sumtable slot %u empty", slot);
    return -1;
  }
  const unsigned int S = c->sh.states, R = c->sh.rate_cats;
  const size_t dbytes = (size_t)R * S * 4 * sizeof(double);
  if (dbytes > c->stage_bytes)
  {
    pllhip_set_error("pllhip_likelihood_derivatives: diagptable too large");
    return -1;
  }
  const bool dna = (S == 4 && (R == 1 || R == 2 || R == 4 || R == 8));
  const bool asc_epilogue = c->sh.asc_states && (c->asc_type & PLLHIP_AB_MASK) &&
                            (c->asc_type & PLLHIP_AB_MASK) != PLLHIP_AB_STAMATAKIS;
  const bool aa_tile = (S == 20 && !c->aa_exact && (R == 1 || R == 2 || R == 4));
  // (other category counts: the tile walk chunk by chunk, its table staged; 150 KB of LDS at most)
  const bool aa_chunks = (S == 20 && !c->aa_exact && !aa_tile && pllhip_aa_chunks_enabled() &&
                          (size_t)R * 102 * sizeof(double) + 4 * 11 * 1024 <= 150 * 1024);
  DerivArgs a;
  if (dna || aa_tile) memcpy(a.diag_inline, h_diagptable, dbytes);
  if (!(dna || aa_tile) || asc_epilogue)
  {
    HIP_TRY(hipStreamSynchronize(c->stream)); // staging buffer free?
    memcpy(c->h_stage, h_diagptable, dbytes);
    HIP_TRY(hipMemcpyAsync(c->d_stage, c->h_stage, dbytes, hipMemcpyHostToDevice, c->stream));
  }
  a.sumtable = c->sumtable[slot];
  a.diagp = (const double *)c->d_stage;
  a.rate_weights = c->rate_weights;
  a.freqs = c->freqs;
  a.prop_invar = c->prop_invar;
  a.pattern_weights = c->pattern_weights;
  a.invariant = c->invariant;
  // ordinary sites; the Stamatakis correction treats the extra per-state sites as
  // weighted sites of the same loop (core_derivatives.c:536-545), the other two
  // get an epilogue
  const size_t ordinary = c->sh.sites - c->sh.asc_states;
  a.sites = (unsigned int)ordinary;
  if ((c->asc_type & PLLHIP_AB_MASK) == PLLHIP_AB_STAMATAKIS) a.sites = c->sh.sites;
  if (parent_scaler >= (int)c->sh.scale_buffers || child_scaler >= (int)c->sh.scale_buffers)
  {
    pllhip_set_error("pllhip_likelihood_derivatives: scaler index out of range");
    return -1;
  }
  if (pllhip_asc_derivatives(c, a.sumtable, a.diagp, ordinary, parent_scaler, child_scaler,
                             &c->pending_extra))
    return -1;
  a.rate_cats = R;
  a.states = S;
  for (unsigned int k = 0; k < R; ++k)
  {
    if (h_params_indices[k] >= c->sh.rate_matrices)
    {
      pllhip_set_error("pllhip_likelihood_derivatives: params index out of range");
      return -1;
    }
    a.params_indices[k] = h_params_indices[k];
  }
  unsigned int grid;
  pllhip_prof_scope prof(c, PLLHIP_PROF_DERIVATIVES);
  if (dna)
  {
    grid = pllhip_stream_grid(c, ((size_t)a.sites + 63) / 64 * 64, 256);
    if (grid > PLLHIP_REDUCE_BLOCKS) grid = PLLHIP_REDUCE_BLOCKS;
    // (a Newton iteration is a 17 us kernel: fewer, longer workgroups -- two per CU -- leave the final
    // sum 512 values instead of 1954 and the launch less to dispatch: 33.9 -> 28.8 us per call at
    // 500 k sites; PLLHIP_DERIV_GRID for measurements)
    {
      const char * e = pllhip_env("PLLHIP_DERIV_GRID");
      const unsigned int cap = e && atoi(e) > 0 ? (unsigned int)atoi(e) : (unsigned int)c->num_cus * 2;
      if (grid > cap) grid = cap;
    }
    a.reduce = pllhip_reduce_out(c, grid, 2);
    // A table that fits the 256 MiB Infinity Cache with room to spare is re-read by every Newton
    // iteration from there: no streaming hint on its loads (the hint is for CLV-sized streams that
    // nothing will touch again)
    const size_t table_bytes = (size_t)c->sh.sites * R * S * sizeof(double);
    bool nt = pllhip_use_nt(c) && table_bytes > ((size_t)128 << 20);
    if (const char * e = pllhip_env("PLLHIP_DERIV_NT")) nt = atoi(e) != 0; // (measurements)
#define DERIV_DNA(RCV)                                                        \
    do {                                                                      \
      if (nt) k_derivatives_dna<RCV, true><<<grid, 256, 0, c->stream>>>(a);   \
      else k_derivatives_dna<RCV, false><<<grid, 256, 0, c->stream>>>(a);     \
    } while (0)
    switch (R)
    {
      case 1: DERIV_DNA(1); break;
      case 2: DERIV_DNA(2); break;
      case 4: DERIV_DNA(4); break;
      default: DERIV_DNA(8); break;
    }
#undef DERIV_DNA
  }
  else if (aa_tile)
  {
    const size_t tiles = ((size_t)a.sites + 15) / 16;
    size_t blocks = (tiles + 3) / 4;
    const size_t cap = pllhip_env("PLLHIP_AA_GRID_CAP") ? (size_t)atoi(pllhip_env("PLLHIP_AA_GRID_CAP")) : (size_t)c->num_cus * 2; // two workgroups per CU (three fit the LDS, 47 KB each, and were the default until round 5: 38.4-39.7 -> 35.0-37.6 us per call at 200 k sites, 58 -> 54.5 at 400 k, equal at 50 k; profiles/r5_newton_floor_from_c.txt) (env: tests)
    if (blocks > cap) blocks = cap;
    grid = (unsigned int)blocks;
    a.reduce = pllhip_reduce_out(c, grid, 2);
    const bool nt = pllhip_use_nt(c);
#define DERIV_AA(RCV)                                                                               \
    do {                                                                                            \
      const size_t lds = (size_t)RCV * 80 * sizeof(double) + 4 * (size_t)aa_geom<RCV>::REGION_B;    \
      if (nt) k_derivatives_aa_tile<RCV, true><<<grid, 256, lds, c->stream>>>(a);                   \
      else k_derivatives_aa_tile<RCV, false><<<grid, 256, lds, c->stream>>>(a);                     \
    } while (0)
    switch (R)
    {
      case 1: DERIV_AA(1); break;
      case 2: DERIV_AA(2); break;
      default: DERIV_AA(4); break;
    }
#undef DERIV_AA
  }
  else if (aa_chunks)
  {
    const size_t tiles = ((size_t)a.sites + 15) / 16;
    size_t blocks = (tiles + 3) / 4;
    const unsigned int rc = R % 4 == 0 ? 4u : (R % 2 == 0 ? 2u : 1u);
    const size_t region = rc == 4 ? aa_geom<4>::REGION_B : (rc == 2 ? aa_geom<2>::REGION_B : aa_geom<1>::REGION_B);
    const size_t lds = (size_t)R * 102 * sizeof(double) + 4 * region;
    const size_t per_cu = lds <= 48 * 1024 ? 3 : (lds <= 76 * 1024 ? 2 : 1);
    const size_t cap = pllhip_env("PLLHIP_AA_GRID_CAP") ? (size_t)atoi(pllhip_env("PLLHIP_AA_GRID_CAP")) : (size_t)c->num_cus * per_cu;
    if (blocks > cap) blocks = cap;
    grid = (unsigned int)blocks;
    a.reduce = pllhip_reduce_out(c, grid, 2);
    const bool nt = pllhip_use_nt(c);
#define DERIV_AA_CHUNKS(RCV)                                                                                         \
    do {                                                                                                             \
      if (nt) { HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_derivatives_aa_chunks<RCV, true>),    \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                  \
                k_derivatives_aa_chunks<RCV, true><<<grid, 256, lds, c->stream>>>(a); }                              \
      else { HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_derivatives_aa_chunks<RCV, false>),      \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                     \
             k_derivatives_aa_chunks<RCV, false><<<grid, 256, lds, c->stream>>>(a); }                                \
    } while (0)
    if (rc == 4) DERIV_AA_CHUNKS(4);
    else if (rc == 2) DERIV_AA_CHUNKS(2);
    else DERIV_AA_CHUNKS(1);
#undef DERIV_AA_CHUNKS
  }
  else if (R <= 64 && (size_t)R * (S * 4 + 2) * sizeof(double) <= 65536)
  {
    grid = pllhip_stream_grid(c, (size_t)a.sites, 256);
    if (grid > PLLHIP_REDUCE_BLOCKS) grid = PLLHIP_REDUCE_BLOCKS;
    a.reduce = pllhip_reduce_out(c, grid, 2);
    const size_t lds = (size_t)R * (S * 4 + 2) * sizeof(double);
    switch (S <= 16 ? S : 0u)
    {
#define DERIV_ROWS(SCV) case SCV: k_derivatives_rows<SCV><<<grid, 256, lds, c->stream>>>(a); break
      DERIV_ROWS(1); DERIV_ROWS(2); DERIV_ROWS(3); DERIV_ROWS(4); DERIV_ROWS(5); DERIV_ROWS(6);
      DERIV_ROWS(7); DERIV_ROWS(8); DERIV_ROWS(9); DERIV_ROWS(10); DERIV_ROWS(11); DERIV_ROWS(12);
      DERIV_ROWS(13); DERIV_ROWS(14); DERIV_ROWS(15); DERIV_ROWS(16);
#undef DERIV_ROWS
      default: k_derivatives_rows<0><<<grid, 256, lds, c->stream>>>(a); break;
    }
  }
  else
  {
    grid = pllhip_stream_grid(c, a.sites, 128);
    if (grid > PLLHIP_REDUCE_BLOCKS) grid = PLLHIP_REDUCE_BLOCKS;
    a.reduce = pllhip_reduce_out(c, grid, 2, 128);
    k_derivatives_gen<<<grid, 128, 0, c->stream>>>(a);
  }
  HIP_TRY(hipGetLastError());
  prof.stop();
  {
    int rc = pllhip_finish_reduce(c, a.reduce, grid, 2);
    if (rc) return rc;
  }
  if (c->comm)
  {
    int rc = pllhip_allreduce_result(c, 2);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->h_result, c->d_result, 2 * sizeof(double), hipMemcpyDeviceToHost,
                           c->stream));
  }
  if (c->defer) // a shard of a group: the group waits for all of them
  {
    pllhip_defer_result(c, a.reduce, c->comm != nullptr);
    return 0;
  }
  {
    int rc = pllhip_result_wait_host(c, a.reduce, c->comm != nullptr);
    if (rc) return rc;
  }
  *h_d_f = c->h_result[0];
  *h_dd_f = c->h_result[1];
  return 0;
}
