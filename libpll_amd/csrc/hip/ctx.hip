// ctx.hip -- device context of one partition: HBM allocation, host<->device
// transfers, the HIP-event stopwatch and the RCCL communicator.
//
// HBM layout (sized for 288 GB parts; clv_arena, tipchars and scaler_arena are ONE allocation, placed where the
// device writes fast -- "Where an arena lies" below):
//   clv_arena     [n_clv][sites][rate_cats][states] f64   site-major, state fastest --
//                 the reference layout (pll.c:527-541), which is also the coalesced
//                 one for "one lane per (site,rate)" kernels: a wave touches
//                 64 * states * 8 contiguous bytes.
//   scaler_arena  [scale_buffers][sites (* rate_cats)] u32
//   tipchars      [tips][sites rounded up to 256 B] u8
//   pmatrix       [prob_matrices][rate_cats][states][states] f64  (KBs; L2-resident)
#include <dlfcn.h>
#include <unistd.h>
#include <sched.h>
#include <time.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <rccl/rccl.h>

#include <algorithm>
#include <string>
#include <vector>

#include <atomic>
#include "ctx.hpp"
#include "numerics.hpp"

static thread_local char g_err[512] = "";

void pllhip_set_error(const char * fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char * pllhip_last_error(void) { return g_err; }

// the switches a client may set (everything else: PLLHIP_DEVELOPER=1)
static const char * const pllhip_user_switches[] = {
  "PLLHIP_AA_EXACT",         // 20 states: the bit-exact vector kernels everywhere (no matrix cores, no whole-list kernel)
  "PLLHIP_AA_TI_MFMA",       // 20 states, 0: tip-inner mat-vecs of the whole-list kernel on the vector unit in the reference's order
  "PLLHIP_FUSED",            // 0: one launch per tree level instead of the whole-list kernels
  "PLLHIP_HOSTSUM",          // 0: workgroup sums added on the device (k_final_sum / tickets) instead of by the host
  "PLLHIP_FUSE_REDUCE",      // 0 / 1: force the ticketed in-kernel final sum off / on
  "PLLHIP_SPIN",             // 0: wait for the stream instead of polling host-mapped result words
  "PLLHIP_SHARD_THREADS",    // 0: a sharded partition is driven by the calling thread alone
  "PLLHIP_SHARD_POLL",       // 0: a sharded partition waits for its shards' streams one after another
  "PLLHIP_SHARD_PIN",        // 0: the shards' threads are not bound to the cores next to their devices
  "PLLHIP_PLACEMENT_TRIES",  // how many places in device memory a partition's CLV arena may try (1: take the first)
  "PLLHIP_FUSED_DEBUG",      // diagnostics on stderr
  "PLLHIP_RCCL_DEBUG",       // diagnostics on stderr
  "PLLHIP_DEVELOPER",
};

static bool pllhip_is_user_switch(const char * name)
{
  for (const char * u : pllhip_user_switches)
    if (!strcmp(u, name)) return true;
  return false;
}

// PLLHIP_DEVELOPER is read once (and again by pllhip_env_reload): a developer's switch costs one flag test per read in
// a production run -- several are read per launch -- and the environment is searched for ignored ones only then.
// (atomics: partitions may be created by several threads at once, and the shards' worker threads look switches up on
// every launch -- ADVICE r5)
static std::atomic<int> g_developer{-1};
extern char ** environ;

extern "C" void pllhip_env_reload(void)
{
  const char * dev = getenv("PLLHIP_DEVELOPER");
  const int on = dev && atoi(dev) != 0;
  static std::atomic<bool> said{false};
  if (!on && !said.load(std::memory_order_relaxed))
    for (char ** e = environ; e && *e; ++e)
    {
      if (strncmp(*e, "PLLHIP_", 7)) continue;
      const char * eq = strchr(*e, '=');
      const std::string name(*e, eq ? (size_t)(eq - *e) : strlen(*e));
      if (pllhip_is_user_switch(name.c_str())) continue;
      if (said.exchange(true)) break; // (another thread has said it meanwhile)
      fprintf(stderr, "libpll_amd: %s is a developer's switch and is ignored without PLLHIP_DEVELOPER=1\n", name.c_str());
      break;
    }
  g_developer.store(on, std::memory_order_release);
}

const char * pllhip_env(const char * name)
{
  if (pllhip_is_user_switch(name)) return getenv(name);
  if (g_developer.load(std::memory_order_acquire) < 0) pllhip_env_reload();
  return g_developer.load(std::memory_order_acquire) ? getenv(name) : nullptr;
}

// what the library sees of a variable right now: 1 set and honoured, 0 unset or ignored (tests/test_host.py)
extern "C" int pllhip_env_is_honoured(const char * name) { return pllhip_env(name) != nullptr; }

// 1: the variable is read as it stands; 0: only under PLLHIP_DEVELOPER=1 (tests/test_host.py)
extern "C" int pllhip_env_is_user_switch(const char * name) { return pllhip_is_user_switch(name) ? 1 : 0; }

extern "C" int pllhip_device_count(int * count)
{
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess)
  {
    *count = 0;
    pllhip_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return (int)e;
  }
  *count = n;
  return 0;
}

template <typename T>
static int dev_alloc(T ** p, size_t count, bool zero, hipStream_t s)
{
  *p = nullptr;
  if (!count) return 0;
  HIP_TRY(hipMalloc((void **)p, count * sizeof(T)));
  if (zero) HIP_TRY(hipMemsetAsync(*p, 0, count * sizeof(T), s));
  return 0;
}

// ---- Where an arena lies (round 6).  The speed of a partition's stores depends on WHERE in device memory it was
// placed: ten partitions of BASELINE config 2 created one after the other in one process store their list at
// 5.75-7.28 TB/s and run it in 1,398-1,484 us, each the same every time it is measured, and a new partition that gets
// a slow one's memory back is slow again (tools/placement_probe.py, profiles/r6_placement_probe.txt) -- what rounds
// 2-5 called "slow boxes".  k_fill_zero is the zeroing every arena gets anyway, as a kernel that can be timed:
// contiguous non-temporal 16-byte stores, eight workgroups per CU.
__global__ __launch_bounds__(256) void k_fill_zero(pll_v2d * __restrict__ p, size_t n16)
{
  const pll_v2d z = {0.0, 0.0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
    __builtin_nontemporal_store(z, p + i);
}

// one timed zeroing pass over [p, p + bytes) on stream s (bytes a multiple of 16): GB/s
static int fill_bandwidth(void * p, size_t bytes, hipStream_t s, hipEvent_t e0, hipEvent_t e1, int cus, double * gbs)
{
  HIP_TRY(hipEventRecord(e0, s));
  k_fill_zero<<<(unsigned int)cus * 8u, 256, 0, s>>>(static_cast<pll_v2d *>(p), bytes / 16);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(e1, s));
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  *gbs = ms > 0.f ? (double)bytes / (ms * 1e6) : 0.0;
  return 0;
}

// (tool: the CLV arena zeroed once more, timed -- OVERWRITES every CLV; tools/placement_probe.py)
extern "C" int pllhip_arena_fill_bandwidth(pllhip_ctx_t * c, double * gbs)
{
  if (!c->shards.empty()) { pllhip_set_error("pllhip_arena_fill_bandwidth: not for a sharded context"); return -1; }
  HIP_TRY(hipSetDevice(c->sh.device));
  PLLHIP_CERT_FIRST(c);
  HIP_TRY(hipStreamSynchronize(c->stream));
  const size_t bytes = (c->clv_arena_alloc_bytes / 16) * 16;
  return fill_bandwidth(c->clv_arena, bytes, c->stream, c->ev0, c->ev1, c->num_cus, gbs);
}

// The partition's per-site memory -- CLVs, tip characters, scale buffers: ONE allocation --, placed: up to `tries`
// allocations, each written with the list kernels' own store pattern (every CLV a stream, a wave's tile of each in
// turn: pllhip_probe_clv_streams -- a plain zeroing pass tells slow places from fast ones for arenas of config 2's size
// but not for twice that: profiles/r6_placement_probe.txt, last section) with the clock running; the first fast one or
// else the fastest is kept, the others are held until the choice is made -- a freed allocation is what the next one
// gets -- and then freed; the one kept is zeroed.  Never with less than another arena's worth (+ 4 GB) of device
// memory left free, never for arenas below 384 MB (4 states x 62,500 sites x 64 taxa = 0.5 GB: +7 %; 31,250 sites:
// nothing -- the write stream does not bound it) -- config 4 whole (133 GB) takes what it gets, and is an average over
// the device anyway.  Costs 3-4 ms and 8 GB of transient memory per try at config 2's size.
#define PLLHIP_PLACEMENT_MIN_BYTES ((size_t)384 << 20)
#define PLLHIP_PLACEMENT_GOOD_GBS 6500.0       // (the list pattern: 6.4-7.3 TB/s on the fast places, 5.5-5.9 on the slow ones)
#define PLLHIP_PLACEMENT_GOOD_FILL_GBS 5150.0  // (shapes without a tile: k_fill_zero, 5.2-5.7 against 4.6-4.9 at 8 GB)
static int alloc_arena_placed(pllhip_ctx * c, char ** out, size_t bytes, size_t first_clv, size_t n_clv, size_t clv_stride_b)
{
  *out = nullptr;
  c->placement_tries = 0;
  c->placement_gbs.clear();
  int tries = 12; // (a fresh process has been handed six slow places in a row: profiles/r6_bench_driver_command.json)
  if (const char * e = pllhip_env("PLLHIP_PLACEMENT_TRIES")) tries = atoi(e);
  size_t min_bytes = PLLHIP_PLACEMENT_MIN_BYTES;
  if (const char * e = pllhip_env("PLLHIP_PLACEMENT_MIN_MB")) min_bytes = (size_t)atoi(e) << 20; // (tool switch)
  if (bytes < min_bytes || tries <= 1) return dev_alloc(out, bytes, true, c->stream);
  std::vector<void *> held;
  size_t best = 0;
  for (int t = 0; t < tries; ++t)
  {
    if (t > 0)
    {
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); break; }
      if (free_b < 2 * bytes + ((size_t)4 << 30)) break;
    }
    void * p = nullptr;
    const hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess)
    {
      if (t == 0) { pllhip_set_error("hipMalloc of the partition's per-site memory (%zu bytes): %s", bytes, hipGetErrorString(e)); return (int)e; }
      (void)hipGetLastError();
      break;
    }
    held.push_back(p);
    double gbs = 0.0, good = PLLHIP_PLACEMENT_GOOD_GBS;
    // (the CLVs the lists will write: the inner nodes' -- tip CLVs, if the partition has any, lie ahead of them)
    int rc = pllhip_probe_clv_streams(c, static_cast<char *>(p) + first_clv * clv_stride_b, n_clv, clv_stride_b, &gbs);
    if (rc == 1)
    {
      // (no tile for this shape: the better of two zeroing passes behind a first one)
      double g[3] = {0.0, 0.0, 0.0};
      rc = 0;
      for (int pass = 0; pass < 3 && !rc; ++pass) rc = fill_bandwidth(p, (bytes / 16) * 16, c->stream, c->ev0, c->ev1, c->num_cus, &g[pass]);
      gbs = std::max(g[1], g[2]);
      good = PLLHIP_PLACEMENT_GOOD_FILL_GBS;
    }
    if (rc)
    {
      for (void * h : held) (void)hipFree(h);
      return rc;
    }
    c->placement_gbs.push_back(gbs);
    if (gbs > c->placement_gbs[best]) best = held.size() - 1;
    if (gbs >= good) break;
  }
  for (size_t i = 0; i < held.size(); ++i)
    if (i != best) (void)hipFree(held[i]);
  *out = static_cast<char *>(held[best]);
  c->placement_tries = (int)held.size();
  c->placement_best = (int)best;
  // (zeroed like the reference's, pll.c:525-542, 800-815: the probe wrote ones)
  double unused = 0.0;
  int rc = fill_bandwidth(*out, (bytes / 16) * 16, c->stream, c->ev0, c->ev1, c->num_cus, &unused);
  if (!rc && (bytes & 15)) rc = hipMemsetAsync(*out + (bytes / 16) * 16, 0, bytes & 15, c->stream) == hipSuccess ? 0 : 1;
  if (rc)
  {
    (void)hipFree(*out);
    *out = nullptr;
  }
  return rc;
}

// what the search found: the write rate (GB/s of k_fill_zero) of every place tried, in order; which one was kept.
// Returns the number of places tried (0: no search -- a small arena, or PLLHIP_PLACEMENT_TRIES <= 1).
extern "C" int pllhip_placement_info(pllhip_ctx_t * c, double * gbs, unsigned int cap, int * kept)
{
  const pllhip_ctx * x = c->shards.empty() ? c : c->shards[0];
  for (unsigned int i = 0; i < cap && i < x->placement_gbs.size(); ++i) gbs[i] = x->placement_gbs[i];
  if (kept) *kept = x->placement_best;
  return x->placement_tries;
}

// ---- Quiescing the HIP runtime before anything is torn down (round 4; the crash hunt of DESIGN.md section 3).
// Since round 3 a result-returning call does not wait for the STREAM: the host spins on a word the kernel writes
// (pllhip_result_wait_host).  The host thread then runs AHEAD of the runtime's own completion processing: nobody
// waits on the command's signal, so the runtime retires the launch on its asynchronous signal-handler thread,
// whenever that thread gets to it.  A client that destroys its partition and exits right after its last call tears
// streams, queues and finally the runtime down under that thread: use-after-free INSIDE the runtime -- 1 run in
// 2,500 of the reference's derivatives programs died with SIGABRT (std::system_error from a mutex in freed memory,
// thrown on the handler thread; round 3 saw the same race once as SIGSEGV); 0 of 18,000 with PLLHIP_SPIN=0,
// 0 of 19,400 with a pause before exit (profiles/r4_crash_soak_*.log).
//
// The fix orders teardown behind the handler thread instead of pausing: a host function is enqueued on the stream
// and waited for.  Host functions run on that very thread, in stream order, so when it has run every earlier
// launch of the stream has been retired there.  (The wait is bounded: a runtime that never calls back costs 50 ms
// per destroy, not a hang.)  And an exit handler, registered after the runtime's own, drains the device once more.
static void pllhip_fence_mark(void * flag)
{
  __atomic_store_n(static_cast<int *>(flag), 1, __ATOMIC_RELEASE);
}

static void pllhip_stream_quiesce(hipStream_t stream)
{
  static const bool off = pllhip_env("PLLHIP_QUIESCE") && atoi(pllhip_env("PLLHIP_QUIESCE")) == 0; // (A/B of the crash hunt)
  (void)hipStreamSynchronize(stream);
  if (off) return;
  // (the flag must outlive a callback that fires after the timeout: leaked on that path only)
  int * flag = new int(0);
  if (hipLaunchHostFunc(stream, pllhip_fence_mark, flag) != hipSuccess)
  {
    (void)hipGetLastError();
    delete flag;
    return;
  }
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  bool done = false;
  for (unsigned int spins = 0; !(done = __atomic_load_n(flag, __ATOMIC_ACQUIRE) != 0); ++spins)
  {
    if ((spins & 63u) == 63u)
    {
      clock_gettime(CLOCK_MONOTONIC, &t1);
      if ((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec) > 50000000ll) break;
      sched_yield(); // (the handler thread may want this core)
    }
  }
  (void)hipStreamSynchronize(stream); // the host function's own command
  if (done) delete flag;
}

static void pllhip_exit_drain()
{
  (void)hipDeviceSynchronize();
  if (const char * e = pllhip_env("PLLHIP_EXIT_GRACE_US")) // (experiment knob of the crash hunt)
    if (atoi(e) > 0) usleep((useconds_t)atoi(e));
}

extern "C" int pllhip_ctx_create(const pllhip_shape_t * shape, pllhip_ctx_t ** out)
{
  *out = nullptr;
  if (!shape || shape->states < 2 || !shape->rate_cats || !shape->sites)
  {
    pllhip_set_error("pllhip_ctx_create: bad shape");
    return -1;
  }
  if (shape->rate_cats > PLLHIP_MAX_RATE_CATS)
  {
    pllhip_set_error("pllhip_ctx_create: rate_cats %u > %d unsupported",
                     shape->rate_cats, PLLHIP_MAX_RATE_CATS);
    return -1;
  }
  if (shape->states > 64)
  {
    pllhip_set_error("pllhip_ctx_create: states %u > 64 unsupported", shape->states);
    return -1;
  }
  int ndev = 0;
  int rc = pllhip_device_count(&ndev);
  if (rc) return rc;
  {
    // registered after the HIP runtime's own exit handlers (it is initialised by now): runs before them
    static const bool once = (atexit(pllhip_exit_drain), true);
    (void)once;
  }
  if (shape->device < 0 || shape->device >= ndev)
  {
    pllhip_set_error("pllhip_ctx_create: device %d not present (%d visible)",
                     shape->device, ndev);
    return (int)hipErrorInvalidDevice;
  }
  HIP_TRY(hipSetDevice(shape->device));

  pllhip_ctx * c = new pllhip_ctx();
  c->sh = *shape;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, shape->device));
  c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (const char * e = pllhip_env("PLLHIP_AA_EXACT")) c->aa_exact = atoi(e) != 0;
  if (const char * e = pllhip_env("PLLHIP_FUSED_DEBUG")) c->fused_debug = atoi(e) ? atoi(e) : 1;
  if (const char * e = pllhip_env("PLLHIP_AA_TI_MFMA")) c->aa_ti_mfma = atoi(e) != 0;
  if (const char * e = pllhip_env("PLLHIP_SPIN")) c->no_spin = atoi(e) == 0;
  if (const char * e = pllhip_env("PLLHIP_HOSTSUM")) c->no_hostsum = atoi(e) == 0;
  if (const char * e = pllhip_env("PLLHIP_FUSE_REDUCE")) c->fuse_forced = atoi(e) ? 1 : 0;
  if (const char * e = pllhip_env("PLLHIP_FUSE_MAX_GRID")) c->fuse_max_grid = (unsigned int)atoi(e);
  if (const char * e = pllhip_env("PLLHIP_NT")) c->nt_override = atoi(e); // 0 / 1; 2: the whole-list kernel's counts too
  if (const char * e = pllhip_env("PLLHIP_NO_BATCH")) c->no_batch = atoi(e) != 0;
  if (const char * e = pllhip_env("PLLHIP_FUSED"))
  {
    c->no_fused = atoi(e) == 0;
    c->force_fused = atoi(e) == 2;
  }
  if (const char * e = pllhip_env("PLLHIP_BLOCKS_PER_CU"))
    if (atoi(e) > 0) c->blocks_per_cu = atoi(e);

  HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIP_TRY(hipEventCreate(&c->ev0));
  HIP_TRY(hipEventCreate(&c->ev1));

  const size_t S = shape->states, R = shape->rate_cats, N = shape->sites;
  c->span = S * R;
  c->clv_elems = N * c->span;
  c->scaler_elems = shape->rate_scalers ? N * R : N;
  c->tip_stride = (N + 255) & ~(size_t)255;
  c->clv_stride = c->clv_elems + (size_t)PLLHIP_TAIL_SITES * c->span;
  // (tool switch: extra sites of slack per CLV -- what the distance between the list kernels' output streams does to
  // their stores, tools/ceiling_by_size.sh)
  if (const char * e = pllhip_env("PLLHIP_CLV_PAD_SITES"))
    if (atoi(e) > 0) c->clv_stride += (size_t)atoi(e) * c->span;
  c->scaler_stride = c->scaler_elems + (size_t)PLLHIP_TAIL_SITES * (shape->rate_scalers ? R : 1);
  c->pmat_elems = R * S * S;

  const unsigned int nodes = shape->tips + shape->clv_buffers;
  const unsigned int first = shape->pattern_tip ? shape->tips : 0;
  const size_t n_clv = nodes - first;

  // CLVs are zeroed like the reference's (pll.c:525-542); scalers calloc'd (pll.c:800-815).  Everything the list
  // kernels stream per site -- CLVs, tip characters, scale buffers -- is ONE allocation, so that all of it lies where
  // the search above found the device fast (the buffers allocated next would get the rejected places back).
  {
    const size_t clv_b = n_clv * c->clv_stride * sizeof(double);
    const size_t tip_b = shape->pattern_tip ? shape->tips * c->tip_stride + PLLHIP_TAIL_SITES : 0;
    const size_t sc_b = ((size_t)shape->scale_buffers * c->scaler_stride + PLLHIP_TAIL_SITES * R) * sizeof(unsigned int);
    auto up = [](size_t b) { return (b + 4095) & ~(size_t)4095; };
    char * base = nullptr;
    if ((rc = alloc_arena_placed(c, &base, up(clv_b) + up(tip_b) + up(sc_b), shape->pattern_tip ? 0 : shape->tips,
                                 shape->clv_buffers, c->clv_stride * sizeof(double)))) goto fail;
    c->clv_arena = reinterpret_cast<double *>(base);
    if (tip_b) c->tipchars = reinterpret_cast<unsigned char *>(base + up(clv_b));
    c->scaler_arena = reinterpret_cast<unsigned int *>(base + up(clv_b) + up(tip_b));
    c->clv_arena_alloc_bytes = clv_b;
  }
  c->clv_arena_bytes = n_clv * c->clv_elems * sizeof(double);
  c->clv.assign(nodes, nullptr);
  for (unsigned int i = first; i < nodes; ++i)
    c->clv[i] = c->clv_arena + (size_t)(i - first) * c->clv_stride;
  if ((rc = dev_alloc(&c->pmatrix, (size_t)shape->prob_matrices * c->pmat_elems, true,
                      c->stream))) goto fail;
  if ((rc = dev_alloc(&c->eigenvals, (size_t)shape->rate_matrices * S, true, c->stream))) goto fail;
  if ((rc = dev_alloc(&c->eigenvecs, (size_t)shape->rate_matrices * S * S, true, c->stream))) goto fail;
  if ((rc = dev_alloc(&c->inv_eigenvecs, (size_t)shape->rate_matrices * S * S, true, c->stream))) goto fail;
  if ((rc = dev_alloc(&c->freqs, (size_t)shape->rate_matrices * S, true, c->stream))) goto fail;
  if ((rc = dev_alloc(&c->prop_invar, (size_t)shape->rate_matrices, true, c->stream))) goto fail;
  c->h_prop_invar.assign(shape->rate_matrices, 0.0);
  if ((rc = dev_alloc(&c->rates, R, true, c->stream))) goto fail;
  if ((rc = dev_alloc(&c->rate_weights, R, true, c->stream))) goto fail;
  if ((rc = dev_alloc(&c->pattern_weights, N + PLLHIP_TAIL_SITES, true, c->stream))) goto fail;
  if ((rc = dev_alloc(&c->tipmap, (size_t)256, true, c->stream))) goto fail;
  if ((rc = dev_alloc(&c->block_partials, (size_t)PLLHIP_REDUCE_BLOCKS * 2, true, c->stream))) goto fail;
  if ((rc = dev_alloc(&c->d_result, (size_t)4, true, c->stream))) goto fail;
  if ((rc = dev_alloc(&c->d_zero, (size_t)64, true, c->stream))) goto fail; // a zero word; 16-byte dummy loads read it too
  HIP_TRY(hipHostMalloc((void **)&c->h_result, 4 * sizeof(double), hipHostMallocMapped));
  HIP_TRY(hipHostGetDevicePointer((void **)&c->h_result_dev, c->h_result, 0));
  memset(c->h_result, 0, 4 * sizeof(double));
  HIP_TRY(hipHostMalloc((void **)&c->h_partials, PLLHIP_HOSTSUM_MAX * sizeof(double2), hipHostMallocMapped));
  HIP_TRY(hipHostGetDevicePointer((void **)&c->h_partials_dev, c->h_partials, 0));
  memset(c->h_partials, 0, PLLHIP_HOSTSUM_MAX * sizeof(double2));
  // (the scaling certificate's flag: a word the list kernels raise, read by the host -- ctx.hpp)
  HIP_TRY(hipHostMalloc((void **)&c->h_cert, 64, hipHostMallocMapped));
  HIP_TRY(hipHostGetDevicePointer((void **)&c->h_cert_dev, c->h_cert, 0));
  memset(c->h_cert, 0, 64);
  // (arrival tickets of the reducing kernels: one word per group of 64 workgroups + one)
  if ((rc = dev_alloc(&c->d_counter, (size_t)(PLLHIP_REDUCE_BLOCKS / 64 + 4) * sizeof(unsigned int), true, c->stream))) goto fail;

  // (the sumtable's two matrix sets are built in the device half: 2 x one P-matrix set)
  c->stage_bytes = 64 * 1024 + (size_t)shape->prob_matrices * 16 +
                   (size_t)(shape->tips + shape->clv_buffers) * sizeof(pllhip_op_t) +
                   2 * (size_t)shape->rate_cats * shape->states * shape->states * sizeof(double);
  HIP_TRY(hipHostMalloc(&c->h_stage, c->stage_bytes, hipHostMallocDefault));
  HIP_TRY(hipMalloc(&c->d_stage, c->stage_bytes));

  {
    // pattern weights default to 1 (pll.c:773-786)
    std::vector<unsigned int> ones(N, 1u);
    HIP_TRY(hipMemcpyAsync(c->pattern_weights, ones.data(), N * sizeof(unsigned int),
                           hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  *out = c;
  return 0;

fail:
  pllhip_ctx_destroy(c);
  return rc;
}

// ---- RCCL, bound lazily so single-GPU users never load the library ----
struct rccl_api
{
  void * handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t,
                            ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char * (*GetErrorString)(ncclResult_t) = nullptr;
};
static rccl_api g_rccl;

// ONE RCCL per process.  A host program that has already mapped an RCCL -- PyTorch brings its own
// copy, torch/lib/librccl.so, which is not the one of /opt/rocm/lib -- must not get a second instance
// next to it (two copies keep two sets of device state and transports): the copy ALREADY in the process
// is bound first (RTLD_NOLOAD does not load anything), and only a process without one loads the
// system's.  pllhip_rccl_path() says which file it was (bench.py prints it).
static char g_rccl_path[512] = "";
extern "C" const char * pllhip_rccl_path(void) { return g_rccl_path; }

static int rccl_load()
{
  if (g_rccl.handle) return 0;
  static const char * names[] = {"librccl.so.1", "librccl.so"};
  void * h = nullptr;
  const char * how = "already in the process";
  for (const char * n : names)
    if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
  if (!h)
  {
    how = "loaded by libpll_amd";
    for (const char * n : names)
      if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
  }
  if (!h)
  {
    pllhip_set_error("cannot load librccl: %s", dlerror());
    return -1;
  }
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
  g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy)
  {
    pllhip_set_error("librccl lacks an expected symbol");
    dlclose(h);
    return -1;
  }
  Dl_info info;
  if (dladdr((void *)g_rccl.AllReduce, &info) && info.dli_fname)
    snprintf(g_rccl_path, sizeof(g_rccl_path), "%s (%s)", info.dli_fname, how);
  else
    snprintf(g_rccl_path, sizeof(g_rccl_path), "? (%s)", how);
  if (pllhip_env("PLLHIP_RCCL_DEBUG")) fprintf(stderr, "pllhip: RCCL bound to %s\n", g_rccl_path);
  g_rccl.handle = h;
  return 0;
}

#define NCCL_TRY(expr)                                                       \
  do {                                                                       \
    ncclResult_t r_ = (expr);                                                \
    if (r_ != ncclSuccess) {                                                 \
      pllhip_set_error("%s failed: %s", #expr,                               \
                       g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); \
      return 1000 + (int)r_;                                                 \
    }                                                                        \
  } while (0)

extern "C" int pllhip_comm_unique_id(void * id128)
{
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  if (rccl_load()) return -1;
  ncclUniqueId id;
  NCCL_TRY(g_rccl.GetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return 0;
}

extern "C" int pllhip_comm_init(pllhip_ctx_t * c, int rank, int nranks, const void * id128)
{
  if (!c->shards.empty())
  {
    pllhip_set_error("pllhip_comm_init: this context is already sharded over the devices of the process");
    return -1;
  }
  if (nranks < 1 || rank < 0 || rank >= nranks)
  {
    pllhip_set_error("pllhip_comm_init: bad rank %d of %d", rank, nranks);
    return -1;
  }
  if (rccl_load()) return -1;
  HIP_TRY(hipSetDevice(c->sh.device));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t comm;
  NCCL_TRY(g_rccl.CommInitRank(&comm, nranks, id, rank));
  c->comm = (ncclComm *)comm;
  c->nranks = nranks;
  return 0;
}

// sum d_result[0..count) over all ranks, in place, on the context's stream
int pllhip_allreduce_result(pllhip_ctx * c, unsigned int count)
{
  if (!c->comm) return 0;
  ++c->comm_reduces;
  NCCL_TRY(g_rccl.AllReduce(c->d_result, c->d_result, count, ncclDouble, ncclSum,
                            (ncclComm_t)c->comm, c->stream));
  return 0;
}

extern "C" unsigned long long pllhip_comm_reduces(pllhip_ctx_t * c) { return c->comm_reduces; }

extern "C" void pllhip_ctx_destroy(pllhip_ctx_t * c)
{
  if (c && !c->shards.empty()) { pllhip_group_destroy(c); return; }
  if (!c) return;
  (void)hipSetDevice(c->sh.device);
  if (c->stream) pllhip_stream_quiesce(c->stream);
  if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy((ncclComm_t)c->comm);
  // (c->tipchars and c->scaler_arena lie in c->clv_arena's allocation)
  void * bufs[] = {c->clv_arena, c->pmatrix, c->eigenvals,
                   c->eigenvecs, c->inv_eigenvecs, c->freqs, c->prop_invar, c->rates,
                   c->rate_weights, c->pattern_weights, c->invariant, c->tipmap,
                   c->block_partials, c->d_result, c->d_counter, c->d_zero, c->d_tiptab, c->d_persite, c->d_stage, c->d_asc,
                   c->lnl_scratch};
  for (double * t : c->sumtable)
    if (t) (void)hipFree(t);
  for (void * p : bufs)
    if (p) (void)hipFree(p);
  if (c->d_plan) (void)hipFree(c->d_plan);
  if (c->d_sink) (void)hipFree(c->d_sink);
  if (c->fused_zero_row) (void)hipFree(c->fused_zero_row);
  if (c->d_tile_counter) (void)hipFree(c->d_tile_counter);
  if (c->d_pairtab) (void)hipFree(c->d_pairtab);
  if (c->cherry_pool) (void)hipFree(c->cherry_pool);
  if (c->cherry_codes) (void)hipFree(c->cherry_codes);
  if (c->cherry_zero) (void)hipFree(c->cherry_zero);
  if (c->cherry_pool_all) (void)hipFree(c->cherry_pool_all);
  if (c->split_verdicts) (void)hipFree(c->split_verdicts);
  if (c->root_counts) (void)hipFree(c->root_counts);
  pllhip_aa_fused_free(c);
  for (int b = 0; b < 2; ++b)
  {
    if (c->h_plan[b]) (void)hipHostFree(c->h_plan[b]);
    if (c->plan_done[b]) (void)hipEventDestroy(c->plan_done[b]);
  }
  pllhip_rep_work_free(c);
  pllhip_level_cache_free(c);
  for (pllhip_ctx::node_rows & r : c->rows)
    for (void * p : {(void *)r.site_id, (void *)r.perm, (void *)r.perm_class, (void *)r.lrow, (void *)r.rrow})
      if (p) (void)hipFree(p);
  if (c->h_result) (void)hipHostFree(c->h_result);
  if (c->h_partials) (void)hipHostFree(c->h_partials);
  if (c->h_cert) (void)hipHostFree(c->h_cert);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  for (hipEvent_t e : c->prof_events) (void)hipEventDestroy(e);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  if (const char * e = pllhip_env("PLLHIP_DESTROY_GRACE_US"))
    if (atoi(e) > 0) usleep((useconds_t)atoi(e));
}

extern "C" int pllhip_wait(pllhip_ctx_t * c)
{
  PLLHIP_ALL_SHARDS(c, pllhip_wait(s));
  PLLHIP_CERT_FIRST(c);
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// ---- transfers.  Host buffers are pageable caller memory, so every copy is
// followed by a stream sync: the caller may reuse the buffer on return. ----
static int h2d(pllhip_ctx * c, void * dst, const void * src, size_t bytes)
{
  HIP_TRY(hipSetDevice(c->sh.device));
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

static int d2h(pllhip_ctx * c, void * dst, const void * src, size_t bytes)
{
  HIP_TRY(hipSetDevice(c->sh.device));
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int pllhip_put_tipchars(pllhip_ctx_t * c, unsigned int tip, const unsigned char * h)
{
  PLLHIP_ALL_SHARDS(c, pllhip_put_tipchars(s, tip, h + lo));
  PLLHIP_CERT_FIRST(c);
  if (!c->sh.pattern_tip || tip >= c->sh.tips)
  {
    pllhip_set_error("pllhip_put_tipchars: tip %u invalid", tip);
    return -1;
  }
  return h2d(c, c->tipchars + (size_t)tip * c->tip_stride, h, c->sh.sites);
}

extern "C" int pllhip_put_tipmap(pllhip_ctx_t * c, const unsigned int * h, unsigned int maxstates)
{
  if (!c->shards.empty()) c->maxstates = maxstates;
  PLLHIP_ALL_SHARDS(c, pllhip_put_tipmap(s, h, maxstates));
  PLLHIP_CERT_FIRST(c);
  if (maxstates > 256) { pllhip_set_error("tipmap too large"); return -1; }
  c->maxstates = maxstates;
  return h2d(c, c->tipmap, h, maxstates * sizeof(unsigned int));
}

extern "C" int pllhip_put_clv(pllhip_ctx_t * c, unsigned int idx, const double * h)
{
  PLLHIP_ALL_SHARDS(c, pllhip_put_clv(s, idx, h + lo * c->span));
  PLLHIP_CERT_FIRST(c);
  if (idx >= c->clv.size() || !c->clv[idx])
  {
    pllhip_set_error("pllhip_put_clv: index %u has no CLV", idx);
    return -1;
  }
  pllhip_cert_mark_clv(c, idx, 0.0); // (the caller's values: nothing of ours in them)
  return h2d(c, c->clv[idx], h, c->clv_elems * sizeof(double));
}

__global__ void k_replicate_tip_clv(double * __restrict__ clv, const double * __restrict__ v,
                                    unsigned int sites, unsigned int rate_cats,
                                    unsigned int states)
{
  const size_t total = (size_t)sites * rate_cats * states;
  for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total;
       t += (size_t)gridDim.x * blockDim.x)
  {
    const size_t n = t / ((size_t)rate_cats * states);
    const unsigned int i = (unsigned int)(t % states);
    clv[t] = v[n * states + i];
  }
}

extern "C" int pllhip_put_tip_clv_persite(pllhip_ctx_t * c, unsigned int idx,
                                          const double * h, unsigned int stride)
{
  PLLHIP_ALL_SHARDS(c, pllhip_put_tip_clv_persite(s, idx, h + lo * stride, stride));
  PLLHIP_CERT_FIRST(c);
  if (idx >= c->clv.size() || !c->clv[idx])
  {
    pllhip_set_error("pllhip_put_tip_clv_persite: index %u has no CLV", idx);
    return -1;
  }
  pllhip_cert_mark_clv(c, idx, 0.0);
  const size_t S = c->sh.states, N = c->sh.sites;
  // stage the compact [sites][states] vectors in the tail of the parent CLV's
  // own storage?  No: use a temporary so partially written CLVs never alias.
  double * tmp = nullptr;
  HIP_TRY(hipSetDevice(c->sh.device));
  HIP_TRY(hipMalloc((void **)&tmp, N * S * sizeof(double)));
  int rc = 0;
  if (stride == S)
    rc = h2d(c, tmp, h, N * S * sizeof(double));
  else
  {
    hipError_t e = hipMemcpy2DAsync(tmp, S * sizeof(double), h, stride * sizeof(double),
                                    S * sizeof(double), N, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) { pllhip_set_error("hipMemcpy2DAsync: %s", hipGetErrorString(e)); rc = (int)e; }
  }
  if (!rc)
  {
    const size_t total = c->clv_elems;
    k_replicate_tip_clv<<<pllhip_stream_grid(c, total, 256), 256, 0, c->stream>>>(
        c->clv[idx], tmp, c->sh.sites, c->sh.rate_cats, c->sh.states);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { pllhip_set_error("replicate: %s", hipGetErrorString(e)); rc = (int)e; }
  }
  (void)hipFree(tmp);
  return rc;
}

extern "C" int pllhip_put_pattern_weights(pllhip_ctx_t * c, const unsigned int * h)
{
  PLLHIP_ALL_SHARDS(c, pllhip_put_pattern_weights(s, h + lo));
  return h2d(c, c->pattern_weights, h, (size_t)c->sh.sites * sizeof(unsigned int));
}

extern "C" int pllhip_put_invariant(pllhip_ctx_t * c, const int * h)
{
  PLLHIP_ALL_SHARDS(c, pllhip_put_invariant(s, h ? h + lo : nullptr));
  HIP_TRY(hipSetDevice(c->sh.device));
  if (!h)
  {
    if (c->invariant) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->invariant)); }
    c->invariant = nullptr;
    return 0;
  }
  if (!c->invariant)
  {
    const size_t bytes = ((size_t)c->sh.sites + PLLHIP_TAIL_SITES) * sizeof(int);
    HIP_TRY(hipMalloc((void **)&c->invariant, bytes));
    HIP_TRY(hipMemsetAsync(c->invariant, 0xff, bytes, c->stream)); // -1 everywhere
  }
  return h2d(c, c->invariant, h, (size_t)c->sh.sites * sizeof(int));
}

extern "C" int pllhip_put_rates(pllhip_ctx_t * c, const double * r, const double * w)
{
  PLLHIP_ALL_SHARDS(c, pllhip_put_rates(s, r, w));
  int rc = 0;
  if (r) rc = h2d(c, c->rates, r, c->sh.rate_cats * sizeof(double));
  if (!rc && w) rc = h2d(c, c->rate_weights, w, c->sh.rate_cats * sizeof(double));
  return rc;
}

extern "C" int pllhip_put_model(pllhip_ctx_t * c, unsigned int pi, const double * evals,
                                const double * evecs, const double * inv_evecs,
                                const double * freqs, double prop_invar)
{
  PLLHIP_ALL_SHARDS(c, pllhip_put_model(s, pi, evals, evecs, inv_evecs, freqs, prop_invar));
  if (pi >= c->sh.rate_matrices) { pllhip_set_error("pllhip_put_model: index %u", pi); return -1; }
  const size_t S = c->sh.states;
  int rc = 0;
  if (evals) rc = h2d(c, c->eigenvals + pi * S, evals, S * sizeof(double));
  if (!rc && evecs) rc = h2d(c, c->eigenvecs + pi * S * S, evecs, S * S * sizeof(double));
  if (!rc && inv_evecs) rc = h2d(c, c->inv_eigenvecs + pi * S * S, inv_evecs, S * S * sizeof(double));
  if (!rc && freqs) rc = h2d(c, c->freqs + pi * S, freqs, S * sizeof(double));
  if (!rc) rc = h2d(c, c->prop_invar + pi, &prop_invar, sizeof(double));
  c->h_prop_invar[pi] = prop_invar;
  c->any_prop_invar = false;
  for (double p : c->h_prop_invar)
    if (p > 0) c->any_prop_invar = true;
  return rc;
}

extern "C" int pllhip_put_pmatrix(pllhip_ctx_t * c, unsigned int idx, const double * h)
{
  PLLHIP_ALL_SHARDS(c, pllhip_put_pmatrix(s, idx, h));
  PLLHIP_CERT_FIRST(c);
  if (idx >= c->sh.prob_matrices) { pllhip_set_error("pllhip_put_pmatrix: index %u", idx); return -1; }
  return h2d(c, pllhip_pmat_ptr(c, idx), h, c->pmat_elems * sizeof(double));
}

extern "C" int pllhip_put_scaler(pllhip_ctx_t * c, unsigned int idx, const unsigned int * h)
{
  PLLHIP_ALL_SHARDS(c, pllhip_put_scaler(s, idx, h + lo * (c->sh.rate_scalers ? c->sh.rate_cats : 1)));
  PLLHIP_CERT_FIRST(c);
  if (idx >= c->sh.scale_buffers) { pllhip_set_error("pllhip_put_scaler: index %u", idx); return -1; }
  return h2d(c, pllhip_scaler_ptr(c, (int)idx), h, c->scaler_elems * sizeof(unsigned int));
}

// parent[n] = lookup[(code1[n] << shift) + code2[n]]: one lane per 16 bytes of the parent row
// (T = double2; T = double for rows of an odd number of doubles)
template <typename T>
__global__ void k_rows_from_lookup(T * __restrict__ parent, const T * __restrict__ lookup,
                                   const unsigned char * __restrict__ c1, const unsigned char * __restrict__ c2,
                                   size_t sites, unsigned int span2, unsigned int shift, size_t rows)
{
  const size_t total = sites * span2;
  for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x)
  {
    const size_t n = t / span2;
    const unsigned int g = (unsigned int)(t - n * span2);
    size_t row = ((size_t)c1[n] << shift) + c2[n];
    if (row >= rows) row = 0; // (characters outside the table: the reference would read past it)
    parent[t] = lookup[row * span2 + g];
  }
}

extern "C" int pllhip_partial_tt_from_lookup(pllhip_ctx_t * c, unsigned int parent_clv, int parent_scaler,
                                             unsigned int tip1, unsigned int tip2, const double * h_lookup,
                                             size_t rows, unsigned int log2_maxstates)
{
  if (!c->shards.empty()) { pllhip_set_error("pllhip_partial_tt_from_lookup: not for a sharded context"); return -1; }
  PLLHIP_CERT_FIRST(c);
  if (parent_clv >= c->clv.size() || !c->clv[parent_clv] || !c->sh.pattern_tip || tip1 >= c->sh.tips ||
      tip2 >= c->sh.tips || parent_scaler >= (int)c->sh.scale_buffers)
  {
    pllhip_set_error("pllhip_partial_tt_from_lookup: bad arguments");
    return -1;
  }
  pllhip_cert_mark_clv(c, parent_clv, 0.0);
  HIP_TRY(hipSetDevice(c->sh.device));
  double * d_lookup = nullptr;
  HIP_TRY(hipMalloc((void **)&d_lookup, rows * c->span * sizeof(double)));
  int rc = h2d(c, d_lookup, h_lookup, rows * c->span * sizeof(double));
  if (!rc)
  {
    if (c->span & 1)
      k_rows_from_lookup<double><<<pllhip_stream_grid(c, (size_t)c->sh.sites * c->span, 256), 256, 0, c->stream>>>(
          c->clv[parent_clv], d_lookup, pllhip_tip_ptr(c, tip1), pllhip_tip_ptr(c, tip2), c->sh.sites,
          (unsigned int)c->span, log2_maxstates, rows);
    else
      k_rows_from_lookup<double2><<<pllhip_stream_grid(c, (size_t)c->sh.sites * (c->span / 2), 256), 256, 0, c->stream>>>(
          reinterpret_cast<double2 *>(c->clv[parent_clv]), reinterpret_cast<const double2 *>(d_lookup),
          pllhip_tip_ptr(c, tip1), pllhip_tip_ptr(c, tip2), c->sh.sites, (unsigned int)(c->span / 2),
          log2_maxstates, rows);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && parent_scaler >= 0)
      e = hipMemsetAsync(pllhip_scaler_ptr(c, parent_scaler), 0, c->scaler_elems * sizeof(unsigned int), c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { pllhip_set_error("rows from lookup: %s", hipGetErrorString(e)); rc = (int)e; }
  }
  (void)hipFree(d_lookup);
  return rc;
}

// ---- a SHARD's CLV / scale buffer as one row per site, however it is stored.  A partition over several devices
// with PLL_ATTRIB_SITE_REPEATS (round 4) identifies its classes per shard -- repeats are a property of a site range --
// and the host layer sees such a group as "stored per site": the expansion happens here, shard by shard.
template <typename T>
static int expand_rows(pllhip_ctx * s, unsigned int owner_clv, T * h, size_t per)
{
  const unsigned int classes = s->rows[owner_clv].classes, sites = s->sh.sites;
  std::vector<unsigned int> sid(sites);
  const int rc = pllhip_get_site_id(s, owner_clv, sid.data());
  if (rc) return rc;
  const std::vector<T> rows(h, h + (size_t)classes * per);
  for (size_t n = 0; n < sites; ++n) memcpy(h + n * per, rows.data() + (size_t)sid[n] * per, per * sizeof(T));
  return 0;
}

static int shard_clv_per_site(pllhip_ctx * s, unsigned int idx, double * h)
{
  const int rc = pllhip_get_clv(s, idx, h);
  if (rc || s->rows.empty() || idx >= s->rows.size() || !s->rows[idx].classes) return rc;
  return expand_rows(s, idx, h, s->span);
}

static int shard_scaler_per_site(pllhip_ctx * s, unsigned int idx, unsigned int * h)
{
  const int rc = pllhip_get_scaler(s, idx, h);
  if (rc || s->rows.empty() || idx >= s->scaler_owner.size() || s->scaler_owner[idx] < 0) return rc;
  const unsigned int owner = (unsigned int)s->scaler_owner[idx];
  if (owner >= s->rows.size() || !s->rows[owner].classes) return 0;
  return expand_rows(s, owner, h, (size_t)(s->sh.rate_scalers ? s->sh.rate_cats : 1));
}

extern "C" int pllhip_get_clv(pllhip_ctx_t * c, unsigned int idx, double * h)
{
  PLLHIP_ALL_SHARDS(c, shard_clv_per_site(s, idx, h + lo * c->span));
  PLLHIP_CERT_FIRST(c);
  if (idx >= c->clv.size() || !c->clv[idx])
  {
    pllhip_set_error("pllhip_get_clv: index %u has no CLV", idx);
    return -1;
  }
  return d2h(c, h, c->clv[idx], c->clv_elems * sizeof(double));
}

extern "C" int pllhip_get_scaler(pllhip_ctx_t * c, unsigned int idx, unsigned int * h)
{
  PLLHIP_ALL_SHARDS(c, shard_scaler_per_site(s, idx, h + lo * (c->sh.rate_scalers ? c->sh.rate_cats : 1)));
  PLLHIP_CERT_FIRST(c);
  if (idx >= c->sh.scale_buffers) { pllhip_set_error("pllhip_get_scaler: index %u", idx); return -1; }
  return d2h(c, h, pllhip_scaler_ptr(c, (int)idx), c->scaler_elems * sizeof(unsigned int));
}

// ---- several buffers to pinned host memory in one launch (pllhip.h: pllhip_mirror_batch)
#define PLLHIP_MIRROR_BATCH 48
struct MirrorCopies
{
  const unsigned int * src[PLLHIP_MIRROR_BATCH];
  unsigned int * dst[PLLHIP_MIRROR_BATCH];
  unsigned int words[PLLHIP_MIRROR_BATCH];
};
// blockIdx.y = buffer; 16 bytes per lane where the alignment allows, words for the tail
__global__ __launch_bounds__(256) void k_mirror_copy(MirrorCopies m)
{
  const unsigned int * __restrict__ src = m.src[blockIdx.y];
  unsigned int * __restrict__ dst = m.dst[blockIdx.y];
  const size_t words = m.words[blockIdx.y], quads = words / 4;
  const uint4 * s4 = reinterpret_cast<const uint4 *>(src);
  uint4 * d4 = reinterpret_cast<uint4 *>(dst);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (size_t)gridDim.x * blockDim.x) d4[i] = s4[i];
  if (blockIdx.x == 0 && threadIdx.x < words - 4 * quads) dst[4 * quads + threadIdx.x] = src[4 * quads + threadIdx.x];
}

extern "C" void * pllhip_host_alloc(size_t bytes)
{
  void * p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) return nullptr;
  return p;
}

extern "C" void pllhip_host_free(void * p)
{
  if (p) (void)hipHostFree(p);
}

extern "C" int pllhip_mirror_batch(pllhip_ctx_t * c, const pllhip_mirror_job_t * jobs, unsigned int count)
{
  if (!c->shards.empty()) { pllhip_set_error("pllhip_mirror_batch: not for a sharded context"); return -1; }
  HIP_TRY(hipSetDevice(c->sh.device));
  PLLHIP_CERT_FIRST(c);
  for (unsigned int first = 0; first < count; first += PLLHIP_MIRROR_BATCH)
  {
    const unsigned int n = std::min<unsigned int>(PLLHIP_MIRROR_BATCH, count - first);
    MirrorCopies m;
    size_t most = 0;
    for (unsigned int k = 0; k < n; ++k)
    {
      const pllhip_mirror_job_t & j = jobs[first + k];
      void * dev = nullptr;
      HIP_TRY(hipHostGetDevicePointer(&dev, j.h, 0));
      if (j.kind == 0)
      {
        if (j.index >= c->clv.size() || !c->clv[j.index] || (!c->rows.empty() && c->rows[j.index].classes))
        {
          pllhip_set_error("pllhip_mirror_batch: CLV %u", j.index);
          return -1;
        }
        m.src[k] = reinterpret_cast<const unsigned int *>(c->clv[j.index]);
        m.words[k] = (unsigned int)(c->clv_elems * 2);
        if (c->clv_elems * 2 > 0xffffffffull) { pllhip_set_error("pllhip_mirror_batch: CLV too large"); return -1; }
      }
      else
      {
        if (j.index >= c->sh.scale_buffers) { pllhip_set_error("pllhip_mirror_batch: scale buffer %u", j.index); return -1; }
        m.src[k] = pllhip_scaler_ptr(c, (int)j.index);
        m.words[k] = (unsigned int)c->scaler_elems;
      }
      m.dst[k] = static_cast<unsigned int *>(dev);
      most = std::max<size_t>(most, m.words[k]);
    }
    const unsigned int gx = (unsigned int)std::min<size_t>(std::max<size_t>(1, (most / 4 + 255) / 256), 64);
    k_mirror_copy<<<dim3(gx, n), 256, 0, c->stream>>>(m);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int pllhip_get_pmatrix(pllhip_ctx_t * c, unsigned int idx, double * h)
{
  if (!c->shards.empty()) return pllhip_get_pmatrix(c->shards[0], idx, h); // (replicated: identical bits on every shard)
  if (idx >= c->sh.prob_matrices) { pllhip_set_error("pllhip_get_pmatrix: index %u", idx); return -1; }
  return d2h(c, h, pllhip_pmat_ptr(c, idx), c->pmat_elems * sizeof(double));
}

extern "C" int pllhip_get_pmatrices(pllhip_ctx_t * c, unsigned int first, unsigned int count, double * h)
{
  if (!c->shards.empty()) return pllhip_get_pmatrices(c->shards[0], first, count, h); // (replicated)
  if (!count || first >= c->sh.prob_matrices || count > c->sh.prob_matrices - first)
  {
    pllhip_set_error("pllhip_get_pmatrices: %u matrices from %u", count, first);
    return -1;
  }
  return d2h(c, h, pllhip_pmat_ptr(c, first), (size_t)count * c->pmat_elems * sizeof(double));
}

extern "C" int pllhip_put_sumtable(pllhip_ctx_t * c, unsigned int slot, const double * h)
{
  PLLHIP_ALL_SHARDS(c, pllhip_put_sumtable(s, slot, h + lo * c->span));
  if (slot >= PLLHIP_SUMTABLE_MAX_SLOTS) { pllhip_set_error("sumtable slot %u", slot); return -1; }
  HIP_TRY(hipSetDevice(c->sh.device));
  if (!c->sumtable[slot]) HIP_TRY(hipMalloc((void **)&c->sumtable[slot], (c->clv_elems + PLLHIP_TAIL_SITES * c->span) * sizeof(double)));
  return h2d(c, c->sumtable[slot], h, c->clv_elems * sizeof(double));
}

extern "C" int pllhip_get_sumtable(pllhip_ctx_t * c, unsigned int slot, double * h)
{
  PLLHIP_ALL_SHARDS(c, pllhip_get_sumtable(s, slot, h + lo * c->span));
  if (slot >= PLLHIP_SUMTABLE_MAX_SLOTS || !c->sumtable[slot])
  {
    pllhip_set_error("sumtable slot %u empty", slot);
    return -1;
  }
  return d2h(c, h, c->sumtable[slot], c->clv_elems * sizeof(double));
}

// Slots the host layer keeps alive before it recycles the least recently used one: what
// fits 32 GiB (a table has the size of a CLV: 128 MB at 1 M sites x 4 x 4), at least 4, at
// most all of them; env PLL_AMD_SUMTABLE_SLOTS overrides.
extern "C" unsigned int pllhip_sumtable_budget(pllhip_ctx_t * c)
{
  if (!c->shards.empty())
  {
    unsigned int n = PLLHIP_SUMTABLE_MAX_SLOTS;
    for (pllhip_ctx * s : c->shards) n = std::min(n, pllhip_sumtable_budget(s));
    return n;
  }
  if (const char * e = getenv("PLL_AMD_SUMTABLE_SLOTS"))
  {
    const int n = atoi(e);
    if (n >= 1) return n > PLLHIP_SUMTABLE_MAX_SLOTS ? PLLHIP_SUMTABLE_MAX_SLOTS : (unsigned int)n;
  }
  const size_t bytes = c->clv_elems * sizeof(double);
  size_t n = ((size_t)32 << 30) / (bytes ? bytes : 1);
  if (n < 4) n = 4;
  if (n > PLLHIP_SUMTABLE_MAX_SLOTS) n = PLLHIP_SUMTABLE_MAX_SLOTS;
  return (unsigned int)n;
}

extern "C" int pllhip_release_sumtable(pllhip_ctx_t * c, unsigned int slot)
{
  PLLHIP_ALL_SHARDS(c, pllhip_release_sumtable(s, slot));
  if (slot >= PLLHIP_SUMTABLE_MAX_SLOTS) { pllhip_set_error("sumtable slot %u", slot); return -1; }
  if (!c->sumtable[slot]) return 0;
  HIP_TRY(hipSetDevice(c->sh.device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipFree(c->sumtable[slot]));
  c->sumtable[slot] = nullptr;
  return 0;
}

extern "C" void * pllhip_dev_clv(pllhip_ctx_t * c, unsigned int idx)
{
  if (!c->shards.empty()) return nullptr; // (one CLV lives on several devices)
  if (c->cert_pending && pllhip_cert_resolve(c)) return nullptr;
  return idx < c->clv.size() ? (void *)c->clv[idx] : nullptr;
}

// ---- per-launch profiling ----
pllhip_prof_scope::pllhip_prof_scope(pllhip_ctx * ctx, int kind) : c(ctx), slot(0), on(false)
{
  if (!c->profiling) return;
  if (c->prof_used * 2 + 2 > c->prof_events.size())
  {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
    c->prof_events.push_back(a);
    c->prof_events.push_back(b);
    c->prof_kind.push_back(kind);
  }
  slot = c->prof_used++;
  c->prof_kind[slot] = kind;
  on = (hipEventRecord(c->prof_events[2 * slot], c->stream) == hipSuccess);
}

void pllhip_prof_scope::stop()
{
  if (on) (void)hipEventRecord(c->prof_events[2 * slot + 1], c->stream);
  on = false;
}

extern "C" int pllhip_profile_enable(pllhip_ctx_t * c, int on)
{
  PLLHIP_ALL_SHARDS(c, pllhip_profile_enable(s, on));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->profiling = on != 0;
  c->prof_used = 0;
  return 0;
}

extern "C" int pllhip_profile_read(pllhip_ctx_t * c, unsigned int * launches, double * total_ms)
{
  if (!c->shards.empty())
  {
    // the launches of the first shard (all shards make the same ones); the others are drained
    for (size_t i = c->shards.size(); i-- > 0;)
    {
      const int rc = pllhip_profile_read(c->shards[i], launches, total_ms);
      if (rc) return rc;
    }
    return 0;
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (int k = 0; k < PLLHIP_PROF_KINDS; ++k) { launches[k] = 0; total_ms[k] = 0.0; }
  for (size_t i = 0; i < c->prof_used; ++i)
  {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->prof_events[2 * i], c->prof_events[2 * i + 1]));
    const int k = c->prof_kind[i];
    launches[k] += 1;
    total_ms[k] += ms;
  }
  c->prof_used = 0;
  return 0;
}

extern "C" int pllhip_timer_start(pllhip_ctx_t * c)
{
  PLLHIP_ALL_SHARDS(c, pllhip_timer_start(s));
  HIP_TRY(hipEventRecord(c->ev0, c->stream));
  return 0;
}

extern "C" int pllhip_timer_stop_ms(pllhip_ctx_t * c, float * ms)
{
  if (!c->shards.empty())
  {
    // the slowest shard (they run side by side)
    *ms = 0.f;
    for (pllhip_ctx * s : c->shards)
    {
      float t = 0.f;
      const int rc = pllhip_timer_stop_ms(s, &t);
      if (rc) return rc;
      if (t > *ms) *ms = t;
    }
    return 0;
  }
  HIP_TRY(hipEventRecord(c->ev1, c->stream));
  HIP_TRY(hipEventSynchronize(c->ev1));
  HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
  c->last_timer_ms = *ms;
  return 0;
}

// what the last pllhip_timer_stop_ms measured on every shard's own stream (one value for an unsharded
// context): a straggling device shows here, the stopwatch itself reports the slowest
extern "C" unsigned int pllhip_timer_shard_ms(pllhip_ctx_t * c, float * ms, unsigned int cap)
{
  if (c->shards.empty())
  {
    if (cap) ms[0] = c->last_timer_ms;
    return 1;
  }
  unsigned int n = 0;
  for (pllhip_ctx * s : c->shards)
  {
    if (n < cap) ms[n] = s->last_timer_ms;
    ++n;
  }
  return n;
}
