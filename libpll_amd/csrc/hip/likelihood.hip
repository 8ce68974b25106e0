// likelihood.hip -- per-site log-likelihood at an edge (or at a root CLV) and
// its sum over sites.
//
// Replaces pll_core_edge_loglikelihood_ii (core_likelihood.c:726; AVX2-flag
// kernels core_likelihood_avx.c:1079 for 4 states, core_likelihood_avx2.c:333
// for 20), _ti_4x4 / _ti (core_likelihood.c:211,412; core_likelihood_avx.c:191,
// core_likelihood_avx2.c:111) and pll_core_root_loglikelihood
// (core_likelihood.c:25).
//
// Kernels: k_lnl_dna (4 states: one lane per 16 bytes, rounds of 64 sites as in
// partials.hip; after a round every lane owns one site and finishes it -- log,
// scaler term, pattern weight, optional per-site store -- once), k_lnl_fast
// (20 states bit-exact: one lane per (site, rate), categories combined with
// __shfl in category order), k_lnl_gen (any other state count: one lane per
// site).  The default 20-state path is likelihood_aa_mfma.hip.  HBM traffic per
// site: two CLVs + two scalers + one weight (268 B for 4x4), nothing written
// unless persite_lnl is asked for.
//
// Site sum: per-lane running sums -> wave __shfl_down tree -> LDS -> one double
// per workgroup -> fixed-order final sum: by the last-arriving workgroup of the same launch
// for small grids (tickets in two levels, lnl_common.hpp), by a one-workgroup k_final_sum launch
// otherwise (measured: pllhip_reduce_out).  The host then spins on a host-mapped word instead of
// waiting for the stream (pllhip_result_wait_host).  Reproducible run to run; the reference adds
// sites sequentially, agreement is ~1e-16*sqrt(sites) relative.  The final step
// also adds the ascertainment-bias correction when one is set (asc_bias.hip).
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "ctx.hpp"
#include "numerics.hpp"

#include "lnl_common.hpp"


ReduceOut pllhip_reduce_out(pllhip_ctx * c, unsigned int grid, unsigned int ncomp, unsigned int block)
{
  ReduceOut r;
  r.partials = c->block_partials;
  r.counter = c->d_counter;
  r.result = c->d_result;
  r.host_result = c->comm ? nullptr : c->h_result_dev;
  // (the word the host spins on: h_result[3]; not with a communicator -- the collective follows)
  r.host_seq = (c->comm || c->no_spin) ? nullptr : reinterpret_cast<unsigned long long *>(c->h_result_dev + 3);
  r.seq = ++c->result_seq;
  r.extra = c->pending_extra; // set by the caller for exactly one launch
  c->pending_extra = nullptr;
  // The last-arriving workgroup finishes the sum in the same launch only for small grids
  // (PLLHIP_FUSE_MAX_GRID workgroups, default 128; PLLHIP_FUSE_REDUCE=0/1 forces either way).  Measured
  // in round 3 with two levels of tickets (64 workgroups per counter, so that no counter serialises
  // thousands of atomics): every workgroup still ends with a write-through store, a wait for it and an
  // atomic round trip, and the finisher reads all partials past the L2 -- 45.8 us per derivative call
  // at 1954 workgroups against 33.9 us with the separate 6 us launch, 31.3 against 28.8 at 512
  // (tools/call_floor_ab.sh).  Small grids are latency-bound and save the launch.
  r.fused = c->fuse_forced >= 0 ? c->fuse_forced : (grid <= c->fuse_max_grid ? 1 : 0);
  // Round 5: small grids too hand their workgroup sums to the host when they can (below) -- a kernel of eight
  // workgroups spent its last microseconds on write-through stores, two ticket atomics, an acquire fence and the
  // finisher's reads: the edge lnL call of a 2,000-site partition 16.2 -> 12.6 us from C (tools/step_floor.c,
  // profiles/r5_step_floor_from_c.txt).  The host adds in the order the finishing workgroup used -- thread t of
  // `block` takes entries t, t + block, ...; then the tree -- so the bits are those of rounds 1-4.
  const bool small = grid <= c->fuse_max_grid;
  const bool can_hostsum = !c->no_hostsum && !c->comm && !r.extra && (size_t)grid * ncomp <= PLLHIP_HOSTSUM_MAX;
  if (c->fuse_forced < 0 && small && can_hostsum) r.fused = 0;
  // Larger grids (round 4): the workgroup sums go straight to host-mapped memory and the host adds them -- in
  // k_final_sum's order, so the bits are those of rounds 1-3 -- instead of a one-workgroup launch behind the kernel
  // (4.5-6.7 us per result-returning call).  Not with a communicator (the all-reduce wants the
  // sum on the device), not with an ascertainment-bias term (a device value the final step adds).  A shard of a
  // group does the same since round 5: the group polls every shard's entries (pllhip_result_wait_pending).
  // PLLHIP_HOSTSUM=0 switches it off.
  r.host_partials = nullptr;
  c->hostsum_grid = 0;
  if (can_hostsum && !r.fused)
  {
    r.host_partials = c->h_partials_dev;
    c->hostsum_grid = grid;
    c->hostsum_ncomp = ncomp;
    c->hostsum_width = small ? block : 256u; // (k_final_sum, which larger grids used to launch, is 256 wide)
  }
  return r;
}

// the host's final sum: k_final_sum's order -- thread t of 256 adds entries t, t + 256, ..., then the tree
static void pllhip_host_final_sum(pllhip_ctx * c)
{
  const unsigned int nparts = c->hostsum_grid, width = c->hostsum_width;
  for (unsigned int comp = 0; comp < c->hostsum_ncomp; ++comp)
  {
    const double2 * part = c->h_partials + (size_t)comp * nparts;
    double s[256];
    for (unsigned int t = 0; t < width; ++t)
    {
      double v = 0.0;
      for (unsigned int i = t; i < nparts; i += width) v += part[i].x;
      s[t] = v;
    }
    for (unsigned int w = width >> 1; w > 0; w >>= 1)
      for (unsigned int t = 0; t < w; ++t) s[t] += s[t + w];
    c->h_result[comp] = s[0];
  }
}

// every workgroup's entry carries this call's sequence number?  (*from: entries below it have been seen to carry it
// already -- a poll goes on where the last one stopped instead of reading all of them again)
static bool pllhip_host_partials_landed(const pllhip_ctx * c, unsigned long long seq, size_t * from = nullptr)
{
  const volatile double2 * part = c->h_partials;
  const size_t n = (size_t)c->hostsum_grid * c->hostsum_ncomp;
  for (size_t i = from ? *from : 0; i < n; ++i)
  {
    // (tag = PLLHIP_SEQ_TAG(seq) ^ bits(value), lnl_common.hpp: holds whichever half the host happens to see first)
    const double y = part[i].y, x = part[i].x;
    unsigned long long got, val;
    memcpy(&got, &y, sizeof(got));
    memcpy(&val, &x, sizeof(val));
    if ((got ^ val) != PLLHIP_SEQ_TAG(seq))
    {
      if (from) *from = i;
      return false;
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  return true;
}

// adds `nparts` workgroup values of `ncomp` components in a fixed order
__global__ __launch_bounds__(256) void k_final_sum(ReduceOut ro, unsigned int nparts, unsigned int ncomp)
{
  __shared__ double s[256];
  for (unsigned int comp = 0; comp < ncomp; ++comp)
  {
    double v = 0.0;
    for (unsigned int i = threadIdx.x; i < nparts; i += 256) v += ro.partials[(size_t)comp * nparts + i];
    s[threadIdx.x] = v;
    __syncthreads();
    for (unsigned int w = 128; w > 0; w >>= 1)
    {
      if (threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0)
    {
      const double total = ro.extra ? s[0] + ro.extra[comp] : s[0];
      ro.result[comp] = total;
      if (ro.host_result) ro.host_result[comp] = total;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) pllhip_publish_seq(ro);
}

// Bounded spin on the host-mapped word: bounded by TIME (a count of PAUSE instructions is 15-50 ms on
// cores where PAUSE takes 140 cycles, and a whole core was burnt for that long under every multi-millisecond
// kernel).  The spin is for the short kernels behind a result-returning call (a 15 us derivative kernel, a 50 us
// lnL kernel: the stream wait costs 6 us more than the spin); after 200 us -- a long kernel, a profiler in
// between, memory that is not coherent after all -- the stream's own completion is waited for, as before round 3.
// (PLLHIP_CPU_RELAX: ctx.hpp)
int pllhip_result_wait_host(pllhip_ctx * c, const ReduceOut & ro, bool stream_work_follows)
{
  const bool hostsum = ro.host_partials != nullptr;
  if ((!ro.host_seq && !(hostsum && !c->no_spin)) || stream_work_follows)
  {
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (hostsum)
    {
      if (!pllhip_host_partials_landed(c, ro.seq))
      {
        pllhip_set_error("a result-returning kernel has finished but its workgroup sums are not in host memory");
        return -1;
      }
      pllhip_host_final_sum(c);
    }
    return 0;
  }
  const volatile unsigned long long * word = reinterpret_cast<const volatile unsigned long long *>(c->h_result + 3);
  struct timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  size_t seen = 0;
  for (;;)
  {
    for (unsigned int spins = 0; spins < 64u; ++spins)
    {
      if (hostsum ? pllhip_host_partials_landed(c, ro.seq, &seen) : *word == ro.seq)
      {
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        if (hostsum) pllhip_host_final_sum(c);
        return 0;
      }
      PLLHIP_CPU_RELAX();
    }
    struct timespec t1;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if ((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec) > 200000ll) break;
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  // (the stream is drained: the word has been written -- unless the memory is not coherent, in which case the
  // values are read through the same mapping and are just as late; the copy-back path does not depend on either)
  if (hostsum)
  {
    if (!pllhip_host_partials_landed(c, ro.seq))
    {
      pllhip_set_error("a result-returning kernel has finished but its workgroup sums are not in host memory");
      return -1;
    }
    pllhip_host_final_sum(c);
  }
  return 0;
}

// a shard's result-returning call has been enqueued: remember how its group waits for it
void pllhip_defer_result(pllhip_ctx * c, const ReduceOut & ro, bool stream_work_follows)
{
  c->pending_seq = ro.seq;
  c->pending_hostsum = ro.host_partials != nullptr;
  c->pending_spin = ro.host_seq != nullptr;
  c->pending_stream_work = stream_work_follows;
}

int pllhip_result_wait_pending(pllhip_ctx * c)
{
  ReduceOut ro;
  memset(&ro, 0, sizeof(ro));
  ro.seq = c->pending_seq;
  ro.host_partials = c->pending_hostsum ? c->h_partials_dev : nullptr;
  ro.host_seq = c->pending_spin ? reinterpret_cast<unsigned long long *>(c->h_result_dev + 3) : nullptr;
  return pllhip_result_wait_host(c, ro, c->pending_stream_work);
}

int pllhip_finish_reduce(pllhip_ctx * c, const ReduceOut & ro, unsigned int grid, unsigned int ncomp)
{
  if (ro.fused || ro.host_partials) return 0;
  k_final_sum<<<1, 256, 0, c->stream>>>(ro, grid, ncomp);
  HIP_TRY(hipGetLastError());
  return 0;
}

// Per-(site,rate) kernels.  KIND: EDGE_II / EDGE_TI / ROOT;  S4: 4-state vs 20-state
template <int RC, int KIND, bool S4>
__global__ __launch_bounds__(256) void k_lnl_fast(LnlArgs a)
{
  constexpr unsigned int S = S4 ? 4 : 20;
  extern __shared__ double smem[];
  const unsigned int k = threadIdx.x & (RC - 1);
  const unsigned int fi = a.freqs_indices[k];
  const double * __restrict__ fr = a.freqs + (size_t)fi * S;

  // stage what every lane of the workgroup shares
  //   ii: the R P-matrices;  ti: pi-weighted tip row sums per code;  root: nothing
  double * sm = smem;
  const unsigned int ncodes = S4 ? 16u : a.maxstates;
  if (KIND == EDGE_II)
  {
    for (unsigned int t = threadIdx.x; t < RC * S * S; t += blockDim.x) sm[t] = a.pmat[t];
  }
  else if (KIND == EDGE_TI)
  {
    for (unsigned int t = threadIdx.x; t < ncodes * RC * S; t += blockDim.x)
    {
      const unsigned int code = t / (RC * S), kk = (t / S) % RC, j = t % S;
      const double * row = a.pmat + ((size_t)kk * S + j) * S;
      const double f = a.freqs[(size_t)a.freqs_indices[kk] * S + j];
      // 4 states: freqs * rowsum (core_likelihood_avx.c:291-301);
      // 20 states: rowsum * freqs (core_likelihood_avx2.c:225)
      sm[t] = S4 ? f * masksum4(row, code) : masksum_seq(row, a.tipmap[code], S) * f;
    }
  }
  __syncthreads();

  double acc = 0.0;
  const size_t total = (size_t)a.sites * RC;
  const size_t total_up = (total + 63) & ~(size_t)63;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const unsigned int lane = threadIdx.x & 63u;
  const unsigned int grp0 = lane & ~(unsigned int)(RC - 1);
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total_up; e += stride)
  {
    const bool act = e < total;
    const size_t ec = act ? e : 0;
    const size_t n = ec / RC;
    const double * pc = a.parent + (size_t)S * ec;

    double terma_r;
    if (S4)
    {
      const double2 * P2 = reinterpret_cast<const double2 *>(pc);
      const double2 p01 = P2[0], p23 = P2[1];
      double t0, t1, t2, t3;
      if (KIND == EDGE_II)
      {
        const double2 * C2 = reinterpret_cast<const double2 *>(a.child + (size_t)S * ec);
        const double2 c01 = C2[0], c23 = C2[1];
        const double * m = sm + k * 16;
        // row dot, x pi, x parent (core_likelihood_avx.c:1175-1213)
        t0 = (fr[0] * dot4(m + 0, c01.x, c01.y, c23.x, c23.y)) * p01.x;
        t1 = (fr[1] * dot4(m + 4, c01.x, c01.y, c23.x, c23.y)) * p01.y;
        t2 = (fr[2] * dot4(m + 8, c01.x, c01.y, c23.x, c23.y)) * p23.x;
        t3 = (fr[3] * dot4(m + 12, c01.x, c01.y, c23.x, c23.y)) * p23.y;
      }
      else if (KIND == EDGE_TI)
      {
        const unsigned int code = a.tip[n] & 15u;
        const double * m = sm + (code * RC + k) * 4;
        t0 = m[0] * p01.x; t1 = m[1] * p01.y; t2 = m[2] * p23.x; t3 = m[3] * p23.y;
      }
      else
      {
        t0 = fr[0] * p01.x; t1 = fr[1] * p01.y; t2 = fr[2] * p23.x; t3 = fr[3] * p23.y;
      }
      terma_r = pairsum4(t0, t1, t2, t3);
    }
    else
    {
      double P[20];
      {
        const double2 * q = reinterpret_cast<const double2 *>(pc);
#pragma unroll
        for (int j = 0; j < 10; ++j) { const double2 t = q[j]; P[2 * j] = t.x; P[2 * j + 1] = t.y; }
      }
      const dview Pv{P};
      if (KIND == EDGE_II)
      {
        double C[20];
        const double2 * q = reinterpret_cast<const double2 *>(a.child + (size_t)S * ec);
#pragma unroll
        for (int j = 0; j < 10; ++j) { const double2 t = q[j]; C[2 * j] = t.x; C[2 * j + 1] = t.y; }
        const dview Cv{C};
        const double * m = sm + k * 400;
        // core_likelihood_avx2.c:432-502: chunks of 4 rows, chunk sums added in order
        terma_r = 0.0;
#pragma unroll
        for (int j = 0; j < 20; j += 4)
        {
          double t[4];
#pragma unroll
          for (int r = 0; r < 4; ++r)
            t[r] = (dot_strided4<true>(m + (j + r) * 20, Cv, 20) * fr[j + r]) * P[j + r];
          terma_r += pairsum4(t[0], t[1], t[2], t[3]);
        }
      }
      else if (KIND == EDGE_TI)
      {
        unsigned int code = a.tip[n];
        if (code >= a.maxstates) code = 0;
        // core_likelihood_avx2.c:262-276: fused strided dot of lookup . parent
        terma_r = dot_strided4<true>(sm + ((size_t)code * RC + k) * 20, Pv, 20);
      }
      else
      {
        terma_r = 0.0;
        for (int j = 0; j < 20; ++j) terma_r += P[j] * fr[j];
      }
    }

    // scalers: per-site count, or per-rate counts reduced to min + capped rest
    unsigned int site_scalings = 0, rel = 0;
    if (a.rate_scalers)
    {
      unsigned int mine = 0;
      if (KIND != ROOT)
      {
        if (a.pscaler) mine += a.pscaler[ec];
        if (KIND == EDGE_II && a.cscaler) mine += a.cscaler[ec];
      }
      unsigned int mn = mine;
      for (unsigned int off = 1; off < RC; off <<= 1)
      {
        const unsigned int o = (unsigned int)__shfl_xor((int)mn, (int)off, 64);
        mn = o < mn ? o : mn;
      }
      site_scalings = mn;
      rel = mine - mn;
      if (rel > PLLHIP_SCALE_RATE_MAXDIFF) rel = PLLHIP_SCALE_RATE_MAXDIFF;
      if (KIND == ROOT)
      {
        // the root kernel ignores per-rate scalers and reads scaler[n] only
        // (core_likelihood.c:197-198)
        site_scalings = a.pscaler ? a.pscaler[n] : 0;
        rel = 0;
      }
    }
    else
    {
      if (a.pscaler) site_scalings += a.pscaler[n];
      if (KIND == EDGE_II && a.cscaler) site_scalings += a.cscaler[n];
    }

    const double contrib = (S4 && KIND != ROOT) ? category_term<true>(a, terma_r, k, n, rel)
                                                : category_term<false>(a, terma_r, k, n, rel);
    // category 0's lane adds the RC contributions in category order
    double terma = 0.0;
#pragma unroll
    for (int i = 0; i < RC; ++i) terma += __shfl(contrib, (int)(grp0 + i), 64);
    if (act && k == 0) acc += site_loglk(a, terma, n, site_scalings);
  }
  block_sum_to_partials(acc, a.reduce);
}

// 4 states: one lane per 16 bytes (two states), lane pairs joined by DPP -- the
// same mapping as the CLV kernels (partials.hip), so every load instruction of
// a wave is one contiguous KiB.  W = 2*RC lanes make one site.
//
// A wave works in rounds of 64 sites: W sub-steps of 64/W sites each.  In
// sub-step j the lanes with (lane % W) == j keep their site's category sum, so
// after the round every lane owns ONE distinct site and the expensive tail
// (log, scaler term, weight, optional per-site store) runs once per site
// instead of once per lane -- f64 VALU ops cost 4 cycles per wave on CDNA4 and
// a redundant log on all 8 lanes of a site made the kernel VALU-bound.
template <int RC, int KIND, bool NT, bool GATHER>
__global__ __launch_bounds__(256) void k_lnl_dna(LnlArgs a)
{
  constexpr unsigned int W = 2 * RC;        // lanes per site
  constexpr unsigned int SPS = 64 / W;      // sites per sub-step
  extern __shared__ double smem[]; // EDGE_TI: pi-weighted tip row sums [16][RC][4]
  const unsigned int lane = threadIdx.x & 63u;
  const unsigned int h = lane & 1u;
  const unsigned int k = (lane >> 1) & (RC - 1);
  const unsigned int fi = a.freqs_indices[k];
  const double * __restrict__ frk = a.freqs + (size_t)fi * 4;
  const double fr0 = frk[2 * h], fr1 = frk[2 * h + 1];
  const double wk = a.rate_weights[k];
  const double pinv = a.prop_invar[fi];
  half_rows pm;
  if (KIND == EDGE_II) pm.load(a.pmat, k, h);
  if (KIND == EDGE_TI)
  {
    for (unsigned int t = threadIdx.x; t < 16 * RC * 4; t += blockDim.x)
    {
      const unsigned int code = t / (RC * 4), kk = (t / 4) % RC, j = t % 4;
      // freqs * rowsum (core_likelihood_avx.c:291-301)
      smem[t] = a.freqs[(size_t)a.freqs_indices[kk] * 4 + j] *
                masksum4(a.pmat + ((size_t)kk * 4 + j) * 4, code);
    }
    __syncthreads();
  }
  const unsigned int * ps_rate = a.pscaler ? a.pscaler : a.zero;
  const unsigned int * cs_rate = (KIND == EDGE_II && a.cscaler) ? a.cscaler : a.zero;
  const bool has_ps = a.pscaler != nullptr, has_cs = (KIND == EDGE_II && a.cscaler != nullptr);
  const bool per_rate = a.rate_scalers && KIND != ROOT;

  double acc = 0.0;
  const size_t rounds = ((size_t)a.sites + 63) / 64;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  const unsigned int grp0 = lane & ~(W - 1);
  const double2 * __restrict__ P2 = reinterpret_cast<const double2 *>(a.parent);
  const double2 * __restrict__ C2 = reinterpret_cast<const double2 *>(a.child);
  // per-site counts of the lane's own site; in per-rate mode the minimum over the
  // categories stands in for them and these read the zero word
  const bool site_ps = has_ps && !per_rate, site_cs = has_cs && !per_rate;
  const unsigned int * ps_site = site_ps ? a.pscaler : a.zero;
  const unsigned int * cs_site = site_cs ? a.cscaler : a.zero;
  const int * inv_site = a.invariant ? a.invariant : reinterpret_cast<const int *>(a.zero);
  const bool has_inv = a.invariant != nullptr;
  for (size_t r = wave; r < rounds; r += nwaves)
  {
    double my_terma = 1.0;
    unsigned int my_rate_min = 0;
    // What the lane needs for the ONE site it finishes at the end of the round is
    // requested first, unconditionally (absent arrays read a zero word): inside the
    // `if (n < sites)` tail each of these loads was a separate exposed round trip.
    // (no index clamping anywhere in this kernel: every per-site array has
    // PLLHIP_TAIL_SITES of slack behind it, and a site past the end is never summed)
    const unsigned int own = (lane & (W - 1)) * SPS + (lane / W);
    const size_t n_own = r * 64 + own;
    const unsigned int w_own = a.pattern_weights[n_own];
    // site repeats: the rows of the two CLVs for the 64 sites of this round, one
    // coalesced load each (the maps carry slack); sub-steps take theirs with a __shfl
    unsigned int pi_round = (unsigned int)(r * 64 + lane), ci_round = pi_round;
    if (GATHER && a.pidx) pi_round = a.pidx[r * 64 + lane];
    if (GATHER && a.cidx) ci_round = a.cidx[r * 64 + lane];
    const size_t prow_own = GATHER ? (size_t)(unsigned int)__shfl((int)pi_round, (int)own, 64) : n_own;
    const size_t crow_own = GATHER ? (size_t)(unsigned int)__shfl((int)ci_round, (int)own, 64) : n_own;
    const unsigned int ps_own = ps_site[site_ps ? (a.pscaler_by_site ? n_own : prow_own) : 0];
    const unsigned int cs_own = cs_site[site_cs ? crow_own : 0];
    const int inv_raw = inv_site[has_inv ? n_own : 0];
    const int inv_own = has_inv ? inv_raw : -1;
    // One sub-step = 64/W sites.  The operands of sub-step j+1 are requested before
    // sub-step j is evaluated (rolled loop: 96 VGPRs, 5 waves per SIMD; unrolled, the
    // compiler kept 160 live and still waited for each pair of loads in turn).
    const size_t g0 = r * 64 * W + lane;
    // granule of the parent / child CLV for sub-step j of this lane
    auto granule = [&](unsigned int j, unsigned int rows_of_round) -> size_t {
      if (!GATHER) return g0 + (size_t)j * SPS * W;
      const int src = (int)(j * SPS + lane / W);
      return (size_t)(unsigned int)__shfl((int)rows_of_round, src, 64) * W + (lane & (W - 1));
    };
    // (Round 5 measured TWO sub-steps' operands in flight per lane, and the next round's first two requested before
    // this round's last two are evaluated: at 126 registers -- four waves per SIMD instead of five -- the launch took
    // 61.5 against 55.5 us on BASELINE config 2 with the full grid and the same 55-56 us with a persistent grid of 1024
    // workgroups, profiles/r5_result_calls_ab.txt: bytes in flight are not what holds this kernel at 0.62.  Not kept.
    // Nor is the opposite extreme, every operand of the round requested up front as k_dna_partials does -- 162
    // registers, three waves per SIMD, 192 KB in flight per CU: 58-59 against 55-56 us, r5_lnl_unroll_ab.txt.  The
    // same kernel reads BASELINE config 4's 2.1 GB in 323 us = 6.6 TB/s, 0.83 of the peak: what a 1 M-site launch
    // lacks is length -- ~13 us of its 54 are not streaming.)
    double2 p_next = ld16<NT>(P2 + granule(0, pi_round)), c_next = make_double2(0.0, 0.0);
    if (KIND == EDGE_II) c_next = ld16<NT>(C2 + granule(0, ci_round));
    auto substep = [&](unsigned int j, const double2 p, const double2 c) {
      const size_t gc = g0 + (size_t)j * SPS * W;
      const size_t n = gc / W;
      const size_t ep = granule(j, pi_round) >> 1, ec = granule(j, ci_round) >> 1;
      double t0, t1;
      if (KIND == EDGE_II)
      {
        const double2 cp = make_double2(dpp_pair_swap(c.x), dpp_pair_swap(c.y));
        // row dot, x pi, x parent (core_likelihood_avx.c:1175-1213)
        t0 = (fr0 * pm.dot(0, c, cp)) * p.x;
        t1 = (fr1 * pm.dot(1, c, cp)) * p.y;
      }
      else if (KIND == EDGE_TI)
      {
        const unsigned int code = a.tip[n] & 15u;
        const double2 m = *reinterpret_cast<const double2 *>(smem + (code * RC + k) * 4 + 2 * h);
        t0 = m.x * p.x;
        t1 = m.y * p.y;
      }
      else
      {
        t0 = fr0 * p.x;
        t1 = fr1 * p.y;
      }
      // (t0 + t1) + (t2 + t3): own pair plus the partner's pair
      const double s = t0 + t1;
      double terma_r = s + dpp_pair_swap(s);

      unsigned int mn = 0;
      if (per_rate)
      {
        const unsigned int mine = ps_rate[has_ps ? ep : 0] + cs_rate[has_cs ? ec : 0];
        mn = mine;
        for (unsigned int off = 2; off < W; off <<= 1)
        {
          const unsigned int o = (unsigned int)__shfl_xor((int)mn, (int)off, 64);
          mn = o < mn ? o : mn;
        }
        unsigned int rel = mine - mn;
        if (rel > PLLHIP_SCALE_RATE_MAXDIFF) rel = PLLHIP_SCALE_RATE_MAXDIFF;
        if (rel > 0) terma_r *= scale_minlh(rel);
      }
      // invariant-state index of this sub-step's site: held by the lane that owns it
      const int inv = __shfl(inv_own, (int)(grp0 + j), 64);
      // weighted category term (core_likelihood_avx.c:1225-1240); the 4-state
      // edge kernels skip non-positive terms, the root kernel does not
      double contrib;
      if (KIND != ROOT && !(terma_r > 0.0))
        contrib = 0.0;
      else if (pinv > 0.0)
      {
        const double inv_lk = (inv == -1) ? 0.0 : frk[inv];
        contrib = wk * (terma_r * (1.0 - pinv) + inv_lk * pinv);
      }
      else
        contrib = terma_r * wk;
      double terma = 0.0;
#pragma unroll
      for (int i = 0; i < RC; ++i) terma += __shfl(contrib, (int)(grp0 + 2 * i), 64);
      if ((lane & (W - 1)) == j)
      {
        my_terma = terma;
        my_rate_min = mn;
      }
    };
#pragma unroll 1
    for (unsigned int j = 0; j + 1 < W; ++j)
    {
      const double2 p = p_next, c = c_next;
      p_next = ld16<NT>(P2 + granule(j + 1, pi_round));
      if (KIND == EDGE_II) c_next = ld16<NT>(C2 + granule(j + 1, ci_round));
      substep(j, p, c);
    }
    substep(W - 1, p_next, c_next);
    // lane l now owns site (l % W) * SPS + l / W of this round
    if (n_own < a.sites)
    {
      // per-site counts (ps_own / cs_own are zero in per-rate mode, where the minimum
      // over the categories stands in, core_likelihood_avx.c:1136-1154)
      const unsigned int site_scalings = my_rate_min + ps_own + cs_own;
      double lk = log(my_terma);
      if (site_scalings) lk += (double)site_scalings * log(PLLHIP_SCALE_THRESHOLD);
      lk *= (double)w_own;
      if (a.persite) a.persite[n_own] = lk;
      acc += lk;
    }
  }
  block_sum_to_partials(acc, a.reduce);
}

// any states / any rate_cats: one lane per site
template <int KIND>
__global__ __launch_bounds__(128) void k_lnl_gen(LnlArgs a)
{
  const unsigned int S = a.states, R = a.rate_cats;
  double acc = 0.0;
  for (size_t n = blockIdx.x * (size_t)blockDim.x + threadIdx.x; n < a.sites;
       n += (size_t)gridDim.x * blockDim.x)
  {
    unsigned int rs[PLLHIP_MAX_RATE_CATS];
    unsigned int site_scalings = 0;
    if (a.rate_scalers && KIND != ROOT)
    {
      unsigned int mn = 0xffffffffu;
      for (unsigned int k = 0; k < R; ++k)
      {
        unsigned int v = a.pscaler ? a.pscaler[n * R + k] : 0;
        if (KIND == EDGE_II && a.cscaler) v += a.cscaler[n * R + k];
        rs[k] = v;
        mn = v < mn ? v : mn;
      }
      site_scalings = mn;
      for (unsigned int k = 0; k < R; ++k)
      {
        const unsigned int d = rs[k] - mn;
        rs[k] = d > PLLHIP_SCALE_RATE_MAXDIFF ? PLLHIP_SCALE_RATE_MAXDIFF : d;
      }
    }
    else
    {
      for (unsigned int k = 0; k < R; ++k) rs[k] = 0;
      if (a.pscaler) site_scalings += a.pscaler[n];
      if (KIND == EDGE_II && a.cscaler) site_scalings += a.cscaler[n];
    }
    unsigned int mask = 0;
    if (KIND == EDGE_TI)
    {
      const unsigned int c = a.tip[n];
      mask = (S == 4) ? c : a.tipmap[c];
    }
    double terma = 0.0;
    for (unsigned int k = 0; k < R; ++k)
    {
      const double * fr = a.freqs + (size_t)a.freqs_indices[k] * S;
      const double * pc = a.parent + (n * R + k) * S;
      const double * cc = (KIND == EDGE_II) ? a.child + (n * R + k) * S : nullptr;
      const double * m = a.pmat + (size_t)k * S * S;
      double terma_r = 0.0;
      for (unsigned int j = 0; j < S; ++j)
      {
        if (KIND == ROOT)
          terma_r += pc[j] * fr[j];
        else
        {
          double termb = 0.0;
          if (KIND == EDGE_II)
            for (unsigned int q = 0; q < S; ++q) termb += m[j * S + q] * cc[q];
          else
            for (unsigned int q = 0; q < S; ++q)
              if ((mask >> q) & 1u) termb += m[j * S + q];
          terma_r += pc[j] * fr[j] * termb; // core_likelihood.c:955
        }
      }
      terma += category_term<false>(a, terma_r, k, n, rs[k]);
    }
    acc += site_loglk(a, terma, n, site_scalings);
  }
  block_sum_to_partials(acc, a.reduce);
}

// ---- any other state count up to 64: two passes over fast kernels
//
// The per-state terms of an edge likelihood, (p_i pi_i) * sum_j P[i][j] c_j
// (core_likelihood.c:955), are exactly what a CLV update WITHOUT scaling computes when
// the parent-side "P-matrix" is diag(pi): its row sum 0 + ... + pi_i p_i + ... + 0 is
// exact, and (pi_i p_i) * termb_i is the reference's product.  So the terms are produced
// by the CLV-update kernels of partials_gen_tile.hip (rows-in-registers, LDS-tiled or
// wave-per-row, whichever covers the shape) into a scratch CLV, and k_lnl_rowsum adds
// them per (site, rate) in state order, then forms the site's category sum, logarithm
// and scaler term exactly like k_lnl_gen.  The root likelihood needs no first pass:
// the row sum multiplies by pi itself (core_likelihood.c:80-93).
struct DiagArgs
{
  double * out; // [R][S][S]
  const double * freqs;
  unsigned int states, rate_cats;
  unsigned int freqs_indices[PLLHIP_MAX_RATE_CATS];
};

__global__ void k_diag_freqs(DiagArgs d)
{
  const unsigned int S = d.states, total = d.rate_cats * S * S;
  for (unsigned int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x)
  {
    const unsigned int k = t / (S * S), i = (t / S) % S, j = t % S;
    d.out[t] = (i == j) ? d.freqs[(size_t)d.freqs_indices[k] * S + i] : 0.0;
  }
}

struct RowsumGeom
{
  unsigned int ts;    // sites per tile
  unsigned int pad;   // LDS row stride in doubles (odd: conflict-free for lane-per-row reads)
  unsigned int inv_s; // ceil(2^32 / states)
};

template <bool ROOTK> // true: a.parent is a CLV and the terms are p_i * pi_i; false: a.parent holds the terms
__global__ __launch_bounds__(256) void k_lnl_rowsum(LnlArgs a, RowsumGeom g)
{
  extern __shared__ double smem[];
  const unsigned int S = a.states, R = a.rate_cats, span = S * R, tid = threadIdx.x;
  double * s_tile = smem;                              // [ts * R][pad]
  double * s_term = smem + (size_t)g.ts * R * g.pad;   // [ts * R]
  double acc = 0.0;
  const size_t ntiles = ((size_t)a.sites + g.ts - 1) / g.ts;
  for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x)
  {
    const size_t n0 = tile * g.ts;
    const unsigned int ns = (a.sites - n0 < g.ts) ? (unsigned int)(a.sites - n0) : g.ts;
    __syncthreads();
    for (unsigned int e = tid; e < ns * span; e += 256u)
    {
      const unsigned int row = __umulhi(e, g.inv_s);
      s_tile[row * g.pad + (e - row * S)] = a.parent[n0 * span + e];
    }
    __syncthreads();
    for (unsigned int row = tid; row < ns * R; row += 256u)
    {
      const double * v = s_tile + row * g.pad;
      double t = 0.0;
      if (ROOTK)
      {
        const double * fr = a.freqs + (size_t)a.freqs_indices[row % R] * S;
        for (unsigned int j = 0; j < S; ++j) t += v[j] * fr[j];
      }
      else
        for (unsigned int j = 0; j < S; ++j) t += v[j];
      s_term[row] = t;
    }
    __syncthreads();
    for (unsigned int s = tid; s < ns; s += 256u)
    {
      const size_t n = n0 + s;
      unsigned int rs[PLLHIP_MAX_RATE_CATS];
      unsigned int site_scalings = 0;
      if (a.rate_scalers && !ROOTK)
      {
        unsigned int mn = 0xffffffffu;
        for (unsigned int k = 0; k < R; ++k)
        {
          unsigned int v = a.pscaler ? a.pscaler[n * R + k] : 0;
          if (a.cscaler) v += a.cscaler[n * R + k];
          rs[k] = v;
          mn = v < mn ? v : mn;
        }
        site_scalings = mn;
        for (unsigned int k = 0; k < R; ++k)
        {
          const unsigned int d = rs[k] - mn;
          rs[k] = d > PLLHIP_SCALE_RATE_MAXDIFF ? PLLHIP_SCALE_RATE_MAXDIFF : d;
        }
      }
      else
      {
        for (unsigned int k = 0; k < R; ++k) rs[k] = 0;
        if (a.pscaler) site_scalings += a.pscaler[n];
        if (!ROOTK && a.cscaler) site_scalings += a.cscaler[n];
      }
      double terma = 0.0;
      for (unsigned int k = 0; k < R; ++k) terma += category_term<false>(a, s_term[s * R + k], k, n, rs[k]);
      acc += site_loglk(a, terma, n, site_scalings);
    }
  }
  block_sum_to_partials(acc, a.reduce);
}

// ---- up to 16 states (4 only with an unusual rate_cats): one pass, rows in registers
//
// The mapping of k_gen_rows (partials_gen_tile.hip): one lane per (site, rate) row, both
// rows in registers, P in LDS, loops unrolled over the compile-time state count; the
// category sum and the per-rate scaler minimum travel over the R adjacent lanes of a site
// with __shfl.  Sums in the plain-C order (core_likelihood.c:80-93, :600-640, :940-960).
template <int KIND, int SC>
__global__ __launch_bounds__(256) void k_lnl_rows(LnlArgs a)
{
  extern __shared__ double smem[];
  const unsigned int R = a.rate_cats, tid = threadIdx.x;
  constexpr unsigned int MP = SC * SC + 2;
  if (KIND != ROOT)
  {
    for (unsigned int t = tid; t < R * SC * SC; t += 256u)
      smem[(t / (SC * SC)) * MP + t % (SC * SC)] = a.pmat[t];
    __syncthreads();
  }
  // A wave takes 64 sites at a time, in rounds of spr = 64 / R whole sites (lane = g * R + k
  // is row k of the round's g-th site; all 64 lanes when R is a power of two); after
  // round r the lanes [r * spr, (r+1) * spr) keep the category sums of that round's
  // sites, so that in the end every lane owns ONE site and the logarithm (a hundred
  // 4-cycle f64 instructions) runs once per site instead of once per row.
  const unsigned int lane = tid & 63u;
  const unsigned int spr = 64u / R, nrounds = (64u + spr - 1u) / spr;
  const unsigned int g = lane / R, k = lane - g * R, grp0 = g * R;
  const unsigned int own_round = lane / spr;
  const int own_src = (int)((lane - own_round * spr) * R);
  const double * fr = a.freqs + (size_t)a.freqs_indices[k] * SC;
  const size_t sites_up = ((size_t)a.sites + 63) & ~(size_t)63;
  double acc = 0.0;
  for (size_t sbase = ((size_t)blockIdx.x * 4u + (tid >> 6)) * 64u; sbase < sites_up;
       sbase += (size_t)gridDim.x * 256u)
  {
   double own_terma = 1.0;
   unsigned int own_scalings = 0u;
   for (unsigned int round = 0; round < nrounds; ++round)
   {
    unsigned int koff = k * MP;
    const unsigned int pos = round * spr + g; // the site's place among the wave's 64
    const bool act = g < spr && pos < 64u && sbase + pos < a.sites;
    const size_t n = act ? sbase + pos : 0;
    const size_t ec = act ? n * R + k : 0;
    double p[SC], c[SC];
#pragma unroll
    for (int j = 0; j < SC; ++j) p[j] = a.parent[ec * SC + j];
    unsigned int mask = 0u;
    if (KIND == EDGE_II)
#pragma unroll
      for (int j = 0; j < SC; ++j) c[j] = a.child[ec * SC + j];
    if (KIND == EDGE_TI) mask = (SC == 4) ? a.tip[n] : a.tipmap[a.tip[n]];
    double terma_r = 0.0;
#pragma unroll
    for (int i = 0; i < SC; ++i)
    {
      if (KIND == ROOT)
        terma_r += p[i] * fr[i];
      else
      {
        // (pinned per matrix row: otherwise all S x S reads are issued up front and spilled)
        asm volatile("" : "+v"(koff));
        const double * m = smem + koff;
        double termb = 0.0;
#pragma unroll
        for (int j = 0; j < SC; ++j)
        {
          if (KIND == EDGE_II) termb += m[i * SC + j] * c[j];
          else if ((mask >> j) & 1u) termb += m[i * SC + j];
        }
        terma_r += p[i] * fr[i] * termb;
      }
    }
    unsigned int site_scalings = 0, rel = 0;
    if (a.rate_scalers && KIND != ROOT)
    {
      unsigned int mine = 0;
      if (a.pscaler) mine += a.pscaler[ec];
      if (KIND == EDGE_II && a.cscaler) mine += a.cscaler[ec];
      unsigned int mn = mine;
      for (unsigned int i = 0; i < R; ++i)
      {
        const unsigned int o = (unsigned int)__shfl((int)mine, (int)(grp0 + i), 64);
        mn = o < mn ? o : mn;
      }
      site_scalings = mn;
      rel = mine - mn;
      if (rel > PLLHIP_SCALE_RATE_MAXDIFF) rel = PLLHIP_SCALE_RATE_MAXDIFF;
    }
    else
    {
      if (a.pscaler) site_scalings += a.pscaler[n];
      if (KIND == EDGE_II && a.cscaler) site_scalings += a.cscaler[n];
    }
    const double contrib = category_term<false>(a, terma_r, k, n, rel);
    double terma = 0.0;
    for (unsigned int i = 0; i < R; ++i) terma += __shfl(contrib, (int)(grp0 + i), 64);
    const double t_own = __shfl(terma, own_src, 64);
    const unsigned int s_own = (unsigned int)__shfl((int)site_scalings, own_src, 64);
    if (own_round == round)
    {
      own_terma = t_own;
      own_scalings = s_own;
    }
   }
   if (sbase + lane < a.sites) acc += site_loglk(a, own_terma, sbase + lane, own_scalings);
  }
  block_sum_to_partials(acc, a.reduce);
}

template <int SC>
static void launch_lnl_rows_sc(pllhip_ctx * c, const LnlArgs & a, int kind, unsigned int grid)
{
  const size_t lds = (kind == ROOT) ? 0 : (size_t)a.rate_cats * (SC * SC + 2) * sizeof(double);
  if (kind == EDGE_II) k_lnl_rows<EDGE_II, SC><<<grid, 256, lds, c->stream>>>(a);
  if (kind == EDGE_TI) k_lnl_rows<EDGE_TI, SC><<<grid, 256, lds, c->stream>>>(a);
  if (kind == ROOT) k_lnl_rows<ROOT, SC><<<grid, 256, lds, c->stream>>>(a);
}

// returns 1 if the shape is not covered
static int launch_lnl_rows(pllhip_ctx * c, LnlArgs & a, int kind, unsigned int * grid_out)
{
  const unsigned int S = c->sh.states, R = c->sh.rate_cats;
  // (4 states arrive here only with a rate_cats the dedicated kernels do not cover)
  if (R < 1 || R > 64 || S < 2 || S > 16 || (size_t)R * (S * S + 2) * sizeof(double) > 65536) return 1;
  // a workgroup takes 256 sites per trip
  unsigned int grid = pllhip_stream_grid(c, (size_t)a.sites, 256);
  if (grid > PLLHIP_REDUCE_BLOCKS) grid = PLLHIP_REDUCE_BLOCKS;
  a.reduce = pllhip_reduce_out(c, grid);
  switch (S)
  {
    case 2: launch_lnl_rows_sc<2>(c, a, kind, grid); break;
    case 3: launch_lnl_rows_sc<3>(c, a, kind, grid); break;
    case 4: launch_lnl_rows_sc<4>(c, a, kind, grid); break;
    case 5: launch_lnl_rows_sc<5>(c, a, kind, grid); break;
    case 6: launch_lnl_rows_sc<6>(c, a, kind, grid); break;
    case 7: launch_lnl_rows_sc<7>(c, a, kind, grid); break;
    case 8: launch_lnl_rows_sc<8>(c, a, kind, grid); break;
    case 9: launch_lnl_rows_sc<9>(c, a, kind, grid); break;
    case 10: launch_lnl_rows_sc<10>(c, a, kind, grid); break;
    case 11: launch_lnl_rows_sc<11>(c, a, kind, grid); break;
    case 12: launch_lnl_rows_sc<12>(c, a, kind, grid); break;
    case 13: launch_lnl_rows_sc<13>(c, a, kind, grid); break;
    case 14: launch_lnl_rows_sc<14>(c, a, kind, grid); break;
    case 15: launch_lnl_rows_sc<15>(c, a, kind, grid); break;
    default: launch_lnl_rows_sc<16>(c, a, kind, grid); break;
  }
  *grid_out = grid;
  return 0;
}

// kind as in run_lnl; returns 1 if the shape is not covered
static int launch_lnl_two_pass(pllhip_ctx * c, LnlArgs & a, int kind, unsigned int * grid_out)
{
  // (4 states have their own kernels for every rate_cats.  20 states come here only with a
  // rate_cats no 20-state lnL kernel covers: the first pass then forms the matrix-vector
  // terms in the AVX2-flag order, the products and the state sum associate differently
  // from core_likelihood_avx2.c:432-502 -- last-bit differences, as with k_lnl_gen before.)
  if (c->sh.states == 4 || !pllhip_gen_tile_covers(c)) return 1;
  const unsigned int S = c->sh.states, R = c->sh.rate_cats;
  LnlArgs b = a; // what the row-sum pass sees
  if (kind != ROOT)
  {
    if (!c->lnl_scratch)
      HIP_TRY(hipMalloc((void **)&c->lnl_scratch, (c->clv_elems + PLLHIP_TAIL_SITES * c->span) * sizeof(double)));
    DiagArgs d;
    d.out = (double *)c->d_stage;
    d.freqs = c->freqs;
    d.states = S;
    d.rate_cats = R;
    for (unsigned int k = 0; k < R; ++k) d.freqs_indices[k] = a.freqs_indices[k];
    k_diag_freqs<<<(R * S * S + 255) / 256, 256, 0, c->stream>>>(d);
    HIP_TRY(hipGetLastError());
    PartialsArgs p;
    memset(&p, 0, sizeof(p));
    p.parent = c->lnl_scratch;
    p.tipmap = c->tipmap;
    p.zero = c->d_zero;
    p.sites = a.sites;
    p.rate_cats = R;
    p.states = S;
    p.maxstates = c->maxstates;
    if (kind == EDGE_II)
    {
      p.left = a.parent;
      p.lmat = d.out;
      p.right = a.child;
      p.rmat = a.pmat;
    }
    else
    {
      p.ltip = a.tip;
      p.lmat = a.pmat;
      p.right = a.parent;
      p.rmat = d.out;
    }
    int rc = pllhip_launch_partials(c, p, kind == EDGE_II ? 0 : 1, SCALE_NONE, PLLHIP_PROF_LNL);
    if (rc) return rc;
    b.parent = c->lnl_scratch;
    if (kind == EDGE_TI) b.cscaler = nullptr;
  }
  RowsumGeom g;
  g.pad = S | 1u;
  g.inv_s = (unsigned int)((0x100000000ull + S - 1) / S);
  const size_t per_site = (size_t)R * (g.pad + 1) * sizeof(double);
  size_t ts = 49152 / per_site;
  if (ts > 256) ts = 256;
  if (ts < 1) return 1;
  g.ts = (unsigned int)ts;
  const size_t tiles = ((size_t)a.sites + ts - 1) / ts;
  unsigned int grid = (unsigned int)(tiles < (size_t)c->num_cus * 4 ? tiles : (size_t)c->num_cus * 4);
  if (grid > PLLHIP_REDUCE_BLOCKS) grid = PLLHIP_REDUCE_BLOCKS;
  if (grid < 1) grid = 1;
  b.reduce = a.reduce = pllhip_reduce_out(c, grid);
  const size_t lds = ts * per_site;
  if (kind == ROOT) k_lnl_rowsum<true><<<grid, 256, lds, c->stream>>>(b, g);
  else k_lnl_rowsum<false><<<grid, 256, lds, c->stream>>>(b, g);
  *grid_out = grid;
  return 0;
}

#define LAUNCH_LNL(RCV, KINDV)                                                            \
  do {                                                                                    \
    if (s4 && gather) k_lnl_dna<RCV, KINDV, false, true><<<grid, 256, lds, c->stream>>>(a); \
    else if (s4 && nt) k_lnl_dna<RCV, KINDV, true, false><<<grid, 256, lds, c->stream>>>(a); \
    else if (s4) k_lnl_dna<RCV, KINDV, false, false><<<grid, 256, lds, c->stream>>>(a);   \
    else k_lnl_fast<RCV, KINDV, false><<<grid, 256, lds, c->stream>>>(a);                 \
  } while (0)

#define LAUNCH_LNL_RC(KINDV)                      \
  do {                                            \
    switch (R) {                                  \
      case 1: LAUNCH_LNL(1, KINDV); break;        \
      case 2: LAUNCH_LNL(2, KINDV); break;        \
      case 4: LAUNCH_LNL(4, KINDV); break;        \
      case 8: LAUNCH_LNL(8, KINDV); break;        \
      default: break;                             \
    }                                             \
  } while (0)

static int run_lnl(pllhip_ctx * c, LnlArgs & a, int kind, double * h_persite, double * h_lnl)
{
  HIP_TRY(hipSetDevice(c->sh.device));
  const unsigned int S = c->sh.states, R = c->sh.rate_cats;
  a.freqs = c->freqs;
  a.prop_invar = c->prop_invar;
  a.rate_weights = c->rate_weights;
  a.pattern_weights = c->pattern_weights;
  a.invariant = c->invariant;
  a.tipmap = c->tipmap;
  a.zero = c->d_zero;
  a.sites = c->sh.sites - c->sh.asc_states; // the ascertainment sites are not part of the sum
  a.rate_cats = R;
  a.states = S;
  a.maxstates = c->maxstates;
  a.rate_scalers = c->sh.rate_scalers;
  a.persite = nullptr;
  if (pllhip_asc_lnl(c, a, kind, &c->pending_extra)) return -1;
  if (h_persite)
  {
    if (!c->d_persite) HIP_TRY(hipMalloc((void **)&c->d_persite, (size_t)c->sh.sites * sizeof(double)));
    a.persite = c->d_persite;
  }

  unsigned int grid = 0;
  bool two_pass = false;
  pllhip_prof_scope prof(c, PLLHIP_PROF_LNL);
  const bool mfma = (S == 20 && !c->aa_exact && pllhip_launch_lnl_aa_mfma(c, a, kind, &grid) == 0);
  const bool fast = !mfma && (S == 4 || S == 20) && (R == 1 || R == 2 || R == 4 || R == 8);
  if (fast)
  {
    const bool s4 = (S == 4);
    const bool nt = pllhip_use_nt(c);
    const bool gather = a.pidx || a.cidx;
    // 4 states: a wave consumes 64 sites per round
    grid = s4 ? pllhip_stream_grid(c, ((size_t)a.sites + 63) / 64 * 64, 256)
              : pllhip_stream_grid(c, (size_t)a.sites * R, 256);
    if (grid > PLLHIP_REDUCE_BLOCKS) grid = PLLHIP_REDUCE_BLOCKS;
    // (round 5) no more workgroups than the host adds sums of (PLLHIP_HOSTSUM_MAX): beyond that a one-workgroup
    // k_final_sum launch followed every call -- BASELINE config 4 whole on one GPU, 31,250 workgroups; the waves
    // then walk several rounds each.  4 states: four workgroups per CU at most -- 1 M sites 54.7 -> 50.6 us for the
    // kernel and 64.9 -> 58.3 us for the call (a quarter of the sums for the host to wait for and add), 2 M sites 92 ->
    // 85 / 106 -> 95, 500 k 30.2 -> 27.5 / 39.1 -> 36.2, no difference at 8 M sites or below 250 k
    // (profiles/r5_lnl_grid_cap_ab.txt).  PLLHIP_LNL_GRID: measurements.
    {
      const char * e = pllhip_env("PLLHIP_LNL_GRID");
      unsigned int cap = (unsigned int)PLLHIP_HOSTSUM_MAX;
      if (s4 && (unsigned int)c->num_cus * 4u < cap) cap = (unsigned int)c->num_cus * 4u;
      if (e && atoi(e) > 0) cap = (unsigned int)atoi(e);
      if (grid > cap) grid = cap;
    }
    a.reduce = pllhip_reduce_out(c, grid);
    size_t lds = 0;
    if (kind == EDGE_II && !s4) lds = (size_t)R * S * S * sizeof(double);
    if (kind == EDGE_TI) lds = (size_t)(s4 ? 16u : c->maxstates) * R * S * sizeof(double);
    if (kind == EDGE_II) LAUNCH_LNL_RC(EDGE_II);
    if (kind == EDGE_TI) LAUNCH_LNL_RC(EDGE_TI);
    if (kind == ROOT) LAUNCH_LNL_RC(ROOT);
  }
  else if (!mfma && (two_pass = (launch_lnl_rows(c, a, kind, &grid) == 0 ||
                                 launch_lnl_two_pass(c, a, kind, &grid) == 0)))
  {
  }
  else if (!mfma)
  {
    grid = pllhip_stream_grid(c, a.sites, 128);
    if (grid > PLLHIP_REDUCE_BLOCKS) grid = PLLHIP_REDUCE_BLOCKS;
    a.reduce = pllhip_reduce_out(c, grid, 1, 128);
    if (kind == EDGE_II) k_lnl_gen<EDGE_II><<<grid, 128, 0, c->stream>>>(a);
    if (kind == EDGE_TI) k_lnl_gen<EDGE_TI><<<grid, 128, 0, c->stream>>>(a);
    if (kind == ROOT) k_lnl_gen<ROOT><<<grid, 128, 0, c->stream>>>(a);
  }
  HIP_TRY(hipGetLastError());
  prof.stop();
  {
    int rc = pllhip_finish_reduce(c, a.reduce, grid, 1);
    if (rc) return rc;
  }
  if (c->comm)
  {
    // multi-GPU: sum the per-shard values over xGMI, then fetch
    int rc = pllhip_allreduce_result(c, 1);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->h_result, c->d_result, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  }
  if (h_persite)
    HIP_TRY(hipMemcpyAsync(h_persite, c->d_persite, (size_t)a.sites * sizeof(double),
                           hipMemcpyDeviceToHost, c->stream));
  if (c->defer) // a shard of a group: the group waits for all of them (pllhip_result_wait_pending)
  {
    pllhip_defer_result(c, a.reduce, c->comm != nullptr || h_persite != nullptr);
    return 0;
  }
  {
    int rc = pllhip_result_wait_host(c, a.reduce, c->comm != nullptr || h_persite != nullptr);
    if (rc) return rc;
  }
  *h_lnl = c->h_result[0];
  return 0;
}

static int fill_freqs_indices(pllhip_ctx * c, LnlArgs & a, const unsigned int * h)
{
  for (unsigned int k = 0; k < c->sh.rate_cats; ++k)
  {
    if (h[k] >= c->sh.rate_matrices)
    {
      pllhip_set_error("freqs index %u out of range", h[k]);
      return -1;
    }
    a.freqs_indices[k] = h[k];
  }
  return 0;
}

// The scaling certificate (ctx.hpp) without a bubble: the lnL kernel is enqueued right behind the op list, and the
// list's flag is looked at when the result has arrived -- a word in host memory, written before the lnL kernel
// started.  Raised (rare): the list runs again in the reference's order, and the evaluation once more.  A shard of a
// group (defer) returns without waiting: its group looks at the flags when it has collected (shard.hip).
static int run_lnl_certified(pllhip_ctx * c, LnlArgs & a, int kind, double * h_persite_lnl, double * h_lnl)
{
  // One rank of several (pllhip_comm_init): the evaluation ends in an all-reduce that every rank must enter the same
  // number of times, and a flag is raised on ONE rank -- so the flag is looked at first (one wait for the stream, which
  // this path pays for its result anyway) and the evaluation runs once, on every rank alike.
  if (c->comm && c->cert_pending && !c->defer)
  {
    const int rc = pllhip_cert_resolve(c);
    if (rc) return rc;
  }
  for (;;)
  {
    LnlArgs aa = a;
    int rc = run_lnl(c, aa, kind, h_persite_lnl, h_lnl);
    if (rc || c->defer || !c->cert_pending) return rc;
    bool again = false;
    rc = pllhip_cert_resolve(c, &again, h_persite_lnl == nullptr);
    if (rc || !again) return rc;
  }
}

extern "C" int pllhip_edge_loglikelihood(pllhip_ctx_t * c, unsigned int parent_clv,
                                         int parent_scaler, unsigned int child_clv,
                                         int child_scaler, unsigned int matrix_index,
                                         const unsigned int * h_freqs_indices,
                                         double * h_persite_lnl, double * h_lnl)
{
  if (!c->shards.empty())
    return pllhip_group_edge_loglikelihood(c, parent_clv, parent_scaler, child_clv, child_scaler, matrix_index, h_freqs_indices, h_persite_lnl, h_lnl);
  const unsigned int nodes = (unsigned int)c->clv.size();
  if (parent_clv >= nodes || child_clv >= nodes || matrix_index >= c->sh.prob_matrices ||
      parent_scaler >= (int)c->sh.scale_buffers || child_scaler >= (int)c->sh.scale_buffers)
  {
    pllhip_set_error("pllhip_edge_loglikelihood: index out of range");
    return -1;
  }
  LnlArgs a;
  memset(&a, 0, sizeof(a));
  if (fill_freqs_indices(c, a, h_freqs_indices)) return -1;
  a.pmat = pllhip_pmat_ptr(c, matrix_index);
  const bool tp = pllhip_is_tip(c, parent_clv), tc = pllhip_is_tip(c, child_clv);
  if (tp && tc)
  {
    pllhip_set_error("pllhip_edge_loglikelihood: both ends are pattern tips");
    return -1;
  }
  int kind;
  if (tp || tc)
  {
    // the inner node plays "parent", only its scaler counts (likelihood.c:489-501)
    kind = EDGE_TI;
    a.parent = c->clv[tp ? child_clv : parent_clv];
    a.pscaler = pllhip_scaler_ptr(c, tp ? child_scaler : parent_scaler);
    a.tip = pllhip_tip_ptr(c, tp ? parent_clv : child_clv);
    if (!c->rows.empty() && c->rows[tp ? child_clv : parent_clv].classes)
      a.pidx = c->rows[tp ? child_clv : parent_clv].site_id;
    if (c->sh.states != 4 && c->maxstates == 0)
    {
      pllhip_set_error("pllhip_edge_loglikelihood: tipmap not uploaded");
      return -1;
    }
  }
  else
  {
    kind = EDGE_II;
    a.parent = c->clv[parent_clv];
    a.child = c->clv[child_clv];
    a.pscaler = pllhip_scaler_ptr(c, parent_scaler);
    a.cscaler = pllhip_scaler_ptr(c, child_scaler);
    if (!c->rows.empty())
    {
      if (c->rows[parent_clv].classes) a.pidx = c->rows[parent_clv].site_id;
      if (c->rows[child_clv].classes) a.cidx = c->rows[child_clv].site_id;
    }
  }
  return run_lnl_certified(c, a, kind, h_persite_lnl, h_lnl);
}

extern "C" int pllhip_root_loglikelihood(pllhip_ctx_t * c, unsigned int clv_index,
                                         int scaler_index, const unsigned int * h_freqs_indices,
                                         double * h_persite_lnl, double * h_lnl)
{
  if (!c->shards.empty())
    return pllhip_group_root_loglikelihood(c, clv_index, scaler_index, h_freqs_indices, h_persite_lnl, h_lnl);
  if (clv_index >= c->clv.size() || !c->clv[clv_index] ||
      scaler_index >= (int)c->sh.scale_buffers)
  {
    pllhip_set_error("pllhip_root_loglikelihood: index out of range");
    return -1;
  }
  LnlArgs a;
  memset(&a, 0, sizeof(a));
  if (fill_freqs_indices(c, a, h_freqs_indices)) return -1;
  a.parent = c->clv[clv_index];
  a.pscaler = c->root_scaler_override ? c->root_scaler_override : pllhip_scaler_ptr(c, scaler_index);
  a.pscaler_by_site = c->root_scaler_override != nullptr;
  if (!c->rows.empty() && c->rows[clv_index].classes) a.pidx = c->rows[clv_index].site_id;
  return run_lnl_certified(c, a, ROOT, h_persite_lnl, h_lnl);
}
