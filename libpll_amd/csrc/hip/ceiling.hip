// ceiling.hip -- the box's own ceiling for the whole-list kernels, measured in the run that is being judged.
//
// The whole-list kernels (partials_fused.hip, partials_aa_fused.hip) are write streams: a workgroup takes a tile of
// sites through the whole op list and stores that tile of every parent CLV (and of its scale buffer) op after op --
// K interleaved output streams, one tile-sized piece of each in turn.  What such a stream can reach differs from box
// to box of one part number (round 5: four boxes, 40.5-43.0 G site-updates/s for one library; a bare stream with
// these addresses 5.5-6.9 TB/s), so a fraction of the nameplate 8 TB/s cannot tell a slow box from a slow kernel.
// pllhip_write_ceiling runs NOTHING BUT the stores of an op list -- same parents, same scale buffers, same tile walk
// (a wave's tile of every parent in turn, a fixed stride over the tiles), same cache policy, no arithmetic, no loads
// -- and times it with HIP events on the context's stream: the bytes per second the list kernel's own address pattern
// gets on THIS device now.  bench.py reports it as roofline.box_ceiling beside the nameplate fraction.
//
// It overwrites the CLVs and scale buffers of the list's parents (with ones and zeros): the caller runs the list again
// before it reads anything.  Measurement infrastructure: not part of any pll_* call.
#include "ctx.hpp"
#include "numerics.hpp"

namespace
{
struct CeilStream
{
  unsigned long long clv, counts; // device addresses; counts 0: no scale buffer
};

// tile_b: bytes of one wave's tile of a CLV (a multiple of 1 KB: 64 lanes x 16 bytes per store instruction);
// count_b: bytes of its tile of a scale buffer (4 per site, or per (site, rate)).
// The tile walk is the list kernels': a wave's first rounds by fixed stride, its last third from eight ticket counters
// (the XCDs of a box do not write at the same rate, DESIGN 2.0 "Tile order": without the tickets this kernel ended
// behind the kernel it is the ceiling of -- profiles/r6_bench_c2.json: 1.009).  counters: eight words, 128 bytes apart,
// zero when the launch starts; group g of eight consecutive workgroups draws from counter g % 8, which hands out
// tiles g % 8, g % 8 + 8, ... of the dynamic region.
template <bool NT>
__global__ __launch_bounds__(256) void k_write_ceiling(const CeilStream * __restrict__ streams, unsigned int K, size_t tiles,
                                                       unsigned int tile_b, unsigned int count_b, unsigned int static_rounds,
                                                       unsigned int * counters)
{
  const unsigned int lane = threadIdx.x & 63u;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
  const pll_v2d one = {1.0, 1.0};
  const unsigned int group = (blockIdx.x >> 3) & 7u;
  for (unsigned int round = 0;; ++round)
  {
    size_t t;
    if (round < static_rounds) t = wave + (size_t)round * nwaves;
    else
    {
      unsigned int ticket = 0;
      if (lane == 0) ticket = atomicAdd(counters + group * 32u, 1u);
      ticket = (unsigned int)__builtin_amdgcn_readfirstlane((int)ticket);
      t = (size_t)static_rounds * nwaves + group + 8u * (size_t)ticket;
    }
    if (t >= tiles) break;
    for (unsigned int k = 0; k < K; ++k)
    {
      const unsigned long long clv = streams[k].clv, cnt = streams[k].counts; // (wave-uniform: scalar loads)
      if (cnt && lane * 4u < count_b) *reinterpret_cast<unsigned int *>(cnt + t * count_b + lane * 4u) = 0u;
      char * out = reinterpret_cast<char *>(clv + t * (size_t)tile_b);
      for (unsigned int off = lane * 16u; off < tile_b; off += 1024u)
      {
        pll_v2d * dst = reinterpret_cast<pll_v2d *>(out + off);
        if (NT) __builtin_nontemporal_store(one, dst);
        else *dst = one;
      }
    }
  }
}
} // namespace

// The same stores over a candidate place for a partition's CLVs (ctx.hip "Where an arena lies"): n streams of
// `stride_b` bytes from `base` on, every CLV a stream, no scale buffers; GB/s of the better of two passes behind an
// untimed one.  Returns 1 if this shape has no tile (the caller then times a plain zeroing pass instead).
int pllhip_probe_clv_streams(pllhip_ctx * c, char * base, size_t n, size_t stride_b, double * gbs)
{
  const size_t site_b = c->span * sizeof(double);
  size_t tile_sites = c->sh.states == 20 ? 8 : (c->sh.states == 4 && c->sh.rate_cats == 4) ? 16 : std::max<size_t>(1, 4096 / site_b);
  while ((tile_sites * site_b) % 1024 && tile_sites <= PLLHIP_TAIL_SITES) ++tile_sites;
  if (tile_sites > PLLHIP_TAIL_SITES || !n) return 1;
  std::vector<CeilStream> h(n);
  for (size_t i = 0; i < n; ++i)
  {
    h[i].clv = (unsigned long long)(uintptr_t)(base + i * stride_b);
    h[i].counts = 0ull;
  }
  CeilStream * d = nullptr;
  unsigned int * counters = nullptr;
  const size_t set_words = 8 * 32;
  HIP_TRY(hipMalloc((void **)&d, n * sizeof(CeilStream)));
  hipError_t e = hipMemcpyAsync(d, h.data(), n * sizeof(CeilStream), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMalloc((void **)&counters, 3 * set_words * sizeof(unsigned int));
  if (e == hipSuccess) e = hipMemsetAsync(counters, 0, 3 * set_words * sizeof(unsigned int), c->stream);
  const size_t tiles = ((size_t)c->sh.sites + tile_sites - 1) / tile_sites;
  const unsigned int grid = (unsigned int)std::min<size_t>((tiles + 3) / 4, (size_t)c->num_cus * (c->sh.states == 20 ? 2 : 3));
  const size_t rounds = tiles / ((size_t)grid * 4);
  const unsigned int dynamic_rounds = (unsigned int)std::max<size_t>(2, rounds / 3);
  const unsigned int static_rounds = rounds > dynamic_rounds ? (unsigned int)(rounds - dynamic_rounds) : 0u;
  double best = 0.0;
  for (unsigned int pass = 0; pass < 3 && e == hipSuccess; ++pass)
  {
    if (pass) e = hipEventRecord(c->ev0, c->stream);
    k_write_ceiling<true><<<grid, 256, 0, c->stream>>>(d, (unsigned int)n, tiles, (unsigned int)(tile_sites * site_b), 0u, static_rounds,
                                                       counters + pass * set_words);
    if (e == hipSuccess) e = hipGetLastError();
    if (!pass || e != hipSuccess) continue;
    e = hipEventRecord(c->ev1, c->stream);
    if (e == hipSuccess) e = hipEventSynchronize(c->ev1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev0, c->ev1);
    if (e == hipSuccess && ms > 0.f) best = std::max(best, (double)n * c->sh.sites * site_b / (ms * 1e6));
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  if (counters) (void)hipFree(counters);
  if (e != hipSuccess) { pllhip_set_error("pllhip_probe_clv_streams: %s", hipGetErrorString(e)); return (int)e; }
  *gbs = best;
  return 0;
}

extern "C" int pllhip_write_ceiling(pllhip_ctx_t * c, const pllhip_op_t * ops, unsigned int count, unsigned int reps,
                                    float * ms_per_pass, double * bytes_per_pass)
{
  if (!c->shards.empty()) { pllhip_set_error("pllhip_write_ceiling: not for a sharded context"); return -1; }
  if (!count || !reps) { pllhip_set_error("pllhip_write_ceiling: nothing to do"); return -1; }
  HIP_TRY(hipSetDevice(c->sh.device));
  PLLHIP_CERT_FIRST(c);
  // the tile of a wave: 2 KB of a 4-state CLV (16 sites x 4 categories), 5 KB of a 20-state one (8 sites) -- what the
  // list kernels store per wave and op; other shapes: the largest whole number of sites in about 4 KB
  const size_t site_b = c->span * sizeof(double);
  size_t tile_sites = c->sh.states == 20 ? 8 : (c->sh.states == 4 && c->sh.rate_cats == 4) ? 16 : std::max<size_t>(1, 4096 / site_b);
  while ((tile_sites * site_b) % 1024) ++tile_sites; // (whole store instructions; the arenas carry 64 sites of slack)
  if (tile_sites > PLLHIP_TAIL_SITES) { pllhip_set_error("pllhip_write_ceiling: no tile for this shape"); return -1; }
  const size_t per_count = c->sh.rate_scalers ? c->sh.rate_cats : 1;
  std::vector<CeilStream> h(count);
  double bytes = 0.0;
  for (unsigned int i = 0; i < count; ++i)
  {
    const pllhip_op_t & op = ops[i];
    if (op.parent_clv >= c->clv.size() || !c->clv[op.parent_clv] || op.parent_scaler >= (int)c->sh.scale_buffers)
    {
      pllhip_set_error("pllhip_write_ceiling: op %u names no CLV of this partition", i);
      return -1;
    }
    h[i].clv = (unsigned long long)(uintptr_t)c->clv[op.parent_clv];
    h[i].counts = (unsigned long long)(uintptr_t)pllhip_scaler_ptr(c, op.parent_scaler);
    bytes += (double)c->sh.sites * (site_b + (h[i].counts ? 4.0 * per_count : 0.0));
    pllhip_cert_mark_clv(c, op.parent_clv, 0.0);
  }
  CeilStream * d = nullptr;
  HIP_TRY(hipMalloc((void **)&d, count * sizeof(CeilStream)));
  hipError_t e = hipMemcpyAsync(d, h.data(), count * sizeof(CeilStream), hipMemcpyHostToDevice, c->stream);
  const size_t tiles = ((size_t)c->sh.sites + tile_sites - 1) / tile_sites;
  const unsigned int grid = (unsigned int)std::min<size_t>((tiles + 3) / 4, (size_t)c->num_cus * (c->sh.states == 20 ? 2 : 3));
  const bool nt = pllhip_use_nt(c);
  // the last third of a wave's rounds by ticket (two at least), as the list kernels do; a set of counters per pass
  const size_t rounds = tiles / ((size_t)grid * 4);
  const unsigned int dynamic_rounds = (unsigned int)std::max<size_t>(2, rounds / 3);
  const unsigned int static_rounds = rounds > dynamic_rounds ? (unsigned int)(rounds - dynamic_rounds) : 0u;
  unsigned int * counters = nullptr;
  const size_t set_words = 8 * 32;
  if (e == hipSuccess) e = hipMalloc((void **)&counters, (size_t)(reps + 1) * set_words * sizeof(unsigned int));
  if (e == hipSuccess) e = hipMemsetAsync(counters, 0, (size_t)(reps + 1) * set_words * sizeof(unsigned int), c->stream);
  unsigned int pass = 0;
  auto launch = [&]() {
    unsigned int * ctr = counters + (size_t)pass++ * set_words;
    if (nt) k_write_ceiling<true><<<grid, 256, 0, c->stream>>>(d, count, tiles, (unsigned int)(tile_sites * site_b), (unsigned int)(tile_sites * 4 * per_count), static_rounds, ctr);
    else k_write_ceiling<false><<<grid, 256, 0, c->stream>>>(d, count, tiles, (unsigned int)(tile_sites * site_b), (unsigned int)(tile_sites * 4 * per_count), static_rounds, ctr);
  };
  float ms = 0.f;
  if (e == hipSuccess)
  {
    launch(); // (once untimed)
    e = hipEventRecord(c->ev0, c->stream);
    for (unsigned int r = 0; r < reps && e == hipSuccess; ++r) launch();
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipEventRecord(c->ev1, c->stream);
    if (e == hipSuccess) e = hipEventSynchronize(c->ev1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev0, c->ev1);
  }
  (void)hipFree(d);
  if (counters) (void)hipFree(counters);
  if (e != hipSuccess) { pllhip_set_error("pllhip_write_ceiling: %s", hipGetErrorString(e)); return (int)e; }
  // (kept plans hold addresses only; the CLVs they would relaunch over are simply stale until the list runs again)
  if (ms_per_pass) *ms_per_pass = ms / reps;
  if (bytes_per_pass) *bytes_per_pass = bytes;
  return 0;
}
