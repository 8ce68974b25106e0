// lnl_common.hpp -- internal: argument block and per-site tail shared by the
// log-likelihood kernels (likelihood.hip, likelihood_aa_mfma.hip).
#pragma once
#include "ctx.hpp"
#include "numerics.hpp"

#define PLLHIP_TICKET_GROUP 64u

// Reduction target shared by the reducing kernels: per-workgroup partial sums,
// an arrival counter, and where the final value goes.
struct ReduceOut
{
  double * partials;       // [NCOMP][gridDim.x]
  unsigned int * counter;  // arrival tickets, zero between launches: [0] the groups', [1 + g] group g's workgroups'
  double * result;         // device result [NCOMP] (input of the RCCL all-reduce)
  double * host_result;    // host-mapped copy (nullptr when an all-reduce follows)
  unsigned long long * host_seq; // host-mapped word that takes `seq` once host_result is complete (nullptr: nobody spins)
  unsigned long long seq;
  const double * extra;    // [NCOMP] added to the finished sums (asc-bias correction) or nullptr
  int fused;               // 1: the last-arriving workgroup finishes the sum in this launch (small grids);
                           // 0: a one-workgroup k_final_sum launch follows (measured: pllhip_reduce_out) ...
  double2 * host_partials; // ... unless this is set (round 4): every workgroup stores {its sum, seq} -- ONE 16-byte
                           // store -- into host-mapped memory, [NCOMP][gridDim.x], and the HOST adds them in
                           // k_final_sum's order once every entry carries this call's seq: no second launch
};

// the tag of a host-summed workgroup entry {value, tag}: tag = PLLHIP_SEQ_TAG(seq) ^ bits(value).  The odd
// multiplier makes the tags of any two calls differ in about half of their bits, so a value of one call next to the
// tag half of another would have to differ from its own call's value in exactly those bits to pass.
#define PLLHIP_SEQ_TAG(seq) ((unsigned long long)(seq) * 0x9E3779B97F4A7C15ull)

struct LnlArgs
{
  const double * __restrict__ parent;   // CLV carrying the frequencies side
  const double * __restrict__ child;    // inner child CLV (ii)
  const unsigned char * __restrict__ tip; // tip child codes (ti)
  const unsigned int * __restrict__ pscaler;
  const unsigned int * __restrict__ cscaler;
  const double * __restrict__ pmat;     // [R][S][S]
  const double * __restrict__ freqs;    // [rate_matrices][S]
  const double * __restrict__ prop_invar; // [rate_matrices]
  const double * __restrict__ rate_weights;
  const unsigned int * __restrict__ pattern_weights;
  const int * __restrict__ invariant;   // nullable
  const unsigned int * __restrict__ tipmap;
  const unsigned int * zero;            // device word holding 0
  double * __restrict__ persite;        // nullable
  // site repeats: row of the parent / child CLV (and of their scale buffers) that holds
  // each site; nullptr = the site's own index
  const unsigned int * pidx;
  const unsigned int * cidx;
  ReduceOut reduce;
  unsigned int sites, rate_cats, states, maxstates;
  int rate_scalers;
  // 1: `pscaler` is indexed by the SITE even where the parent CLV is stored by class (pidx) -- a shard's root
  // counts gathered from the whole buffer's entries (shard.hip: gather_root_counts)
  int pscaler_by_site;
  unsigned int freqs_indices[PLLHIP_MAX_RATE_CATS];
};

// ascertainment-bias attribute bits (pll.h:120-124)
#define PLLHIP_AB_LEWIS (1 << 5)
#define PLLHIP_AB_FELSENSTEIN (2 << 5)
#define PLLHIP_AB_STAMATAKIS (3 << 5)
#define PLLHIP_AB_MASK (7 << 5)

enum { EDGE_II = 0, EDGE_TI = 1, ROOT = 2 };

// (2^-256)^d for d = 1..4, exact powers of two (core_likelihood_avx.c:1119-1128)
__device__ __forceinline__ double scale_minlh(unsigned int d)
{
  return d == 1 ? 0x1p-256 : d == 2 ? 0x1p-512 : d == 3 ? 0x1p-768 : 0x1p-1024;
}

// The host does not wait for the STREAM to report the launch complete (10-20 us of runtime and
// interrupt latency per result-returning call -- the floor of a Newton loop): it spins
// on a word of host-mapped memory that the finishing workgroup writes after the result, system scope.
__device__ __forceinline__ void pllhip_publish_seq(const ReduceOut & ro)
{
  if (!ro.host_seq) return;
  __threadfence_system(); // the result before the word
  __hip_atomic_store(ro.host_seq, ro.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Sum `v[0..NCOMP)` over the whole grid, reproducibly: lane sums -> wave
// __shfl_down tree -> LDS -> one value per workgroup in `partials`; the workgroup
// that arrives LAST (write-through stores, ticket, acquire + sc1 loads:
// cdna_hip_programming Guideline 16) adds all workgroup values in index order with a fixed tree and
// publishes the result -- no second kernel launch, same bits every run.
template <int NCOMP>
__device__ __forceinline__ void grid_sum(const double (&v_in)[NCOMP], const ReduceOut & ro)
{
  __shared__ double s_wave[NCOMP][16];
  __shared__ double s_tree[NCOMP][256];
  __shared__ bool s_last;
  const unsigned int wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const unsigned int nparts = gridDim.x;
#pragma unroll
  for (int cidx = 0; cidx < NCOMP; ++cidx)
  {
    double v = v_in[cidx];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) s_wave[cidx][wave] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
#pragma unroll
    for (int cidx = 0; cidx < NCOMP; ++cidx)
    {
      double t = 0.0;
      for (unsigned int w = 0; w < (blockDim.x >> 6); ++w) t += s_wave[cidx][w];
      if (!ro.fused)
      {
        if (ro.host_partials)
        {
          // value and tag travel in ONE 16-byte store, and nobody waits for anybody on the device.  Nothing promises
          // the host that the two halves become visible together, so the tag does not depend on it: it is
          // the call's sequence number, spread over all 64 bits, XOR the value's own bits -- an entry whose halves
          // are of different calls matches neither call's tag (PLLHIP_SEQ_TAG; the host: pllhip_host_partials_landed)
          const pll_v2d e = {t, __longlong_as_double((long long)(PLLHIP_SEQ_TAG(ro.seq) ^ (unsigned long long)__double_as_longlong(t)))};
          *reinterpret_cast<pll_v2d *>(ro.host_partials + (size_t)cidx * nparts + blockIdx.x) = e;
        }
        else
          ro.partials[(size_t)cidx * nparts + blockIdx.x] = t;
        continue;
      }
      // write-through (sc1) store: visible at device scope without an L2 write-back
      // fence, which every workgroup would otherwise pay (measured +10 us per launch)
      __hip_atomic_store(ro.partials + (size_t)cidx * nparts + blockIdx.x, t, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    s_last = false;
    if (ro.fused)
    {
      // Tickets in two levels: groups of PLLHIP_TICKET_GROUP consecutive workgroups share a counter,
      // the last of a group takes a ticket of the top counter, the last of those finishes the sum.
      // One counter for everybody serialises its atomics (~30 ns each across the XCDs: +125 us on a
      // 3907-workgroup grid, which is why grids above 512 workgroups used to launch k_final_sum
      // instead); with 64 per counter the longest chain is 64 + grid / 64 atomics.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the stores have left before the ticket
      const unsigned int g = blockIdx.x / PLLHIP_TICKET_GROUP, ngroups = (nparts + PLLHIP_TICKET_GROUP - 1) / PLLHIP_TICKET_GROUP;
      const unsigned int gsize = (g + 1 == ngroups) ? nparts - g * PLLHIP_TICKET_GROUP : PLLHIP_TICKET_GROUP;
      if (atomicAdd(ro.counter + 1 + g, 1u) == gsize - 1)
        s_last = (atomicAdd(ro.counter, 1u) == ngroups - 1);
    }
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence(); // acquire: drop stale L1 lines before reading the other workgroups' partials
#pragma unroll
  for (int cidx = 0; cidx < NCOMP; ++cidx)
  {
    double v = 0.0;
    for (unsigned int i = threadIdx.x; i < nparts; i += blockDim.x)
      v += __hip_atomic_load(ro.partials + (size_t)cidx * nparts + i, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
    s_tree[cidx][threadIdx.x] = v;
  }
  __syncthreads();
  for (unsigned int w = blockDim.x >> 1; w > 0; w >>= 1)
  {
    if (threadIdx.x < w)
#pragma unroll
      for (int cidx = 0; cidx < NCOMP; ++cidx) s_tree[cidx][threadIdx.x] += s_tree[cidx][threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0)
  {
#pragma unroll
    for (int cidx = 0; cidx < NCOMP; ++cidx)
    {
      const double total = ro.extra ? s_tree[cidx][0] + ro.extra[cidx] : s_tree[cidx][0];
      ro.result[cidx] = total;
      if (ro.host_result) ro.host_result[cidx] = total;
    }
    pllhip_publish_seq(ro);
  }
  // ready for the next launch (stream order separates launches)
  for (unsigned int i = threadIdx.x; i <= (nparts + PLLHIP_TICKET_GROUP - 1) / PLLHIP_TICKET_GROUP; i += blockDim.x) ro.counter[i] = 0u;
}

__device__ __forceinline__ void block_sum_to_partials(double v, const ReduceOut & ro)
{
  const double one[1] = {v};
  grid_sum<1>(one, ro);
}

// category term -> weighted contribution (core_likelihood_avx.c:1219-1240)
template <bool GUARD_POSITIVE>
__device__ __forceinline__ double category_term(const LnlArgs & a, double terma_r,
                                                unsigned int k, size_t n, unsigned int rel_scale)
{
  if (rel_scale > 0) terma_r *= scale_minlh(rel_scale);
  if (GUARD_POSITIVE && !(terma_r > 0.0)) return 0.0;
  const unsigned int fi = a.freqs_indices[k];
  const double pinv = a.prop_invar[fi];
  const double w = a.rate_weights[k];
  if (pinv > 0.0)
  {
    const int inv = a.invariant ? a.invariant[n] : -1;
    const double inv_lk = (inv == -1) ? 0.0 : a.freqs[(size_t)fi * a.states + inv];
    return w * (terma_r * (1.0 - pinv) + inv_lk * pinv);
  }
  return terma_r * w;
}

__device__ __forceinline__ double site_loglk(const LnlArgs & a, double terma, size_t n,
                                             unsigned int site_scalings)
{
  double lk = log(terma);
  if (site_scalings) lk += (double)site_scalings * log(PLLHIP_SCALE_THRESHOLD);
  lk *= (double)a.pattern_weights[n];
  if (a.persite) a.persite[n] = lk;
  return lk;
}


// fills a ReduceOut from the context (host_result only when no all-reduce follows);
// pllhip_finish_reduce launches the final pass when the kernel did not fuse it
ReduceOut pllhip_reduce_out(pllhip_ctx * c, unsigned int grid, unsigned int ncomp = 1, unsigned int block = 256);
int pllhip_finish_reduce(pllhip_ctx * c, const ReduceOut & ro, unsigned int grid, unsigned int ncomp);
// after the launches of a result-returning call: wait until h_result holds this call's values
// (`stream_work_follows`: copies or a collective were enqueued behind the kernel -- wait for the stream)
int pllhip_result_wait_host(pllhip_ctx * c, const ReduceOut & ro, bool stream_work_follows);
// a shard of a group: remember the enqueued call (pllhip_defer_result), wait for it later (ctx.hpp: pending_*)
void pllhip_defer_result(pllhip_ctx * c, const ReduceOut & ro, bool stream_work_follows);

// asc_bias.hip: launch the correction kernel (if a correction type is set) ahead of the
// site kernel; *extra = what that kernel's final sum must add
int pllhip_asc_lnl(pllhip_ctx * c, const LnlArgs & a, int kind, const double ** extra);
int pllhip_asc_derivatives(pllhip_ctx * c, const double * sumtable, const double * d_diagp,
                           size_t ordinary_sites, int parent_scaler, int child_scaler,
                           const double ** extra);

// 20-state kernels on the matrix cores; returns 1 if the case is not covered
int pllhip_launch_lnl_aa_mfma(pllhip_ctx * c, LnlArgs & a, int kind, unsigned int * grid_out);
