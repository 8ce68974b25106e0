// shard.hip -- ONE partition over several devices of one process (north_star: "sites
// of one partition shard naturally across the 8 GPUs of a single node").
//
// The reference has no counterpart (it is single-threaded, one address space): a client calls
// pll_partition_create (pll.h:530) once and gets one partition.  Here that one partition may be
// a GROUP: one ordinary device context per entry of the device list, each holding a contiguous
// range of the sites (boundaries on multiples of 256 sites, every per-site array stays 16-byte
// aligned).  What is per site is split (CLVs, scale buffers, tip characters, pattern weights,
// invariant-site indices, sumtables, per-site lnL); what is not is replicated (P-matrices: every
// device computes them itself from the same host eigen data -> identical bits; model, rates,
// tipmap); an op list is enqueued on every device in turn -- with the whole-list kernel that is
// three launches per device per pll_update_partials, so one host thread keeps eight devices busy.
// The only cross-device operation is the sum of the per-shard lnL (or d, dd): each shard's
// final-sum kernel writes its value into host-mapped memory, the host adds the (at most 8)
// doubles in shard order -- deterministic, and cheaper than a collective for 8 bytes.  (The
// one-process-per-GPU mode, pllhip_comm_init, sums the same values with one RCCL all-reduce.)
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <ctype.h>
#include <pthread.h>
#include <sched.h>
#include <signal.h>
#include <string>
#include <system_error>
#include <thread>

#include "ctx.hpp"

// ---- one host thread per shard (round 5) ----
// The calling thread takes shard 0 itself; shard i > 0 has a thread of its own that sleeps on a condition variable
// between calls (after a short spin: the calls of a likelihood evaluation come in bursts), binds its device once per
// job and runs the same function on its shard.  Errors come back as each shard's return code and message (the
// library's error text is per thread); the first failing shard's is the call's.  PLLHIP_SHARD_THREADS=0: the calling
// thread visits the shards in turn, as until round 4.
struct pllhip_shard_pool
{
  std::vector<std::thread> threads;
  std::mutex m;
  std::condition_variable cv;
  const std::function<int(pllhip_ctx *, size_t)> * job = nullptr;
  std::atomic<unsigned long long> generation{0};
  std::atomic<unsigned int> remaining{0};
  std::vector<int> rc;
  std::vector<std::string> msg;
  bool stop = false;
};

// (round 6) A shard's thread runs on the cores next to its device: the launches of a shard are a
// few microseconds of host work per call, and a thread on the other socket pays the interconnect for every doorbell
// and every poll of the shard's host-mapped result words.  The device's PCI address names its NUMA node
// (/sys/bus/pci/devices/<address>/numa_node), the node its cores (/sys/devices/system/node/node<n>/cpulist); the thread's
// affinity becomes those cores that the PROCESS may use at all (a cgroup or taskset mask is never widened; nothing in
// common, no node, no sysfs: the thread stays where it is).  Only with more than one distinct device, and not with
// PLLHIP_SHARD_PIN=0.  The calling thread (shard 0) is the client's: never touched.
static void pin_near_device(int device)
{
  char bus[64] = "";
  if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess || !bus[0]) return;
  for (char * q = bus; *q; ++q) *q = (char)tolower((unsigned char)*q);
  char path[160];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
  int node = -1;
  if (FILE * f = fopen(path, "r"))
  {
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
  }
  if (node < 0) return;
  snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
  char list[4096] = "";
  if (FILE * f = fopen(path, "r"))
  {
    if (!fgets(list, (int)sizeof(list), f)) list[0] = 0;
    fclose(f);
  }
  cpu_set_t allowed, want;
  if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return;
  CPU_ZERO(&want);
  int any = 0;
  for (char * q = list; *q;)
  {
    char * end = nullptr;
    const long a = strtol(q, &end, 10);
    if (end == q) break;
    long b = a;
    q = end;
    if (*q == '-')
    {
      b = strtol(q + 1, &end, 10);
      q = end;
    }
    for (long cpu = a; cpu <= b && cpu < CPU_SETSIZE; ++cpu)
      if (cpu >= 0 && CPU_ISSET((int)cpu, &allowed))
      {
        CPU_SET((int)cpu, &want);
        any = 1;
      }
    while (*q == ',' || *q == ' ' || *q == '\n') ++q;
  }
  if (any) (void)pthread_setaffinity_np(pthread_self(), sizeof(want), &want);
}

static void shard_worker(pllhip_ctx * g, size_t i)
{
  pllhip_shard_pool & p = *g->pool;
  unsigned long long seen = 0;
  if (g->shard_pin) pin_near_device(g->shards[i]->sh.device);
  for (;;)
  {
    // a short spin (the next call of a burst), then sleep
    bool got = false;
    for (int spin = 0; spin < 2000 && !got; ++spin)
    {
      got = p.generation.load(std::memory_order_acquire) != seen;
#if defined(__x86_64__) || defined(__i386__)
      if (!got) __builtin_ia32_pause();
#endif
    }
    if (!got)
    {
      std::unique_lock<std::mutex> lk(p.m);
      p.cv.wait(lk, [&] { return p.stop || p.generation.load(std::memory_order_acquire) != seen; });
      if (p.stop) return;
    }
    {
      std::lock_guard<std::mutex> lk(p.m); // (the job pointer and the generation were published under the lock)
      if (p.stop) return;
      seen = p.generation.load(std::memory_order_acquire);
    }
    int rc = 0;
    if (hipSetDevice(g->shards[i]->sh.device) != hipSuccess)
    {
      pllhip_set_error("hipSetDevice(%d) on a shard's thread", g->shards[i]->sh.device);
      rc = -1;
    }
    else rc = (*p.job)(g->shards[i], g->shard_lo[i]);
    p.rc[i] = rc;
    if (rc) p.msg[i] = pllhip_last_error();
    p.remaining.fetch_sub(1, std::memory_order_acq_rel);
  }
}

static void pool_destroy(pllhip_ctx * g);

int pllhip_group_parallel(pllhip_ctx * g, const std::function<int(pllhip_ctx *, size_t)> & fn)
{
  pllhip_device_guard guard; // (the caller's current device is put back on return)
  const size_t n = g->shards.size();
  if (n < 2 || !g->shard_threads)
  {
    for (size_t i = 0; i < n; ++i)
    {
      if (hipSetDevice(g->shards[i]->sh.device) != hipSuccess) { pllhip_set_error("hipSetDevice"); return -1; }
      const int rc = fn(g->shards[i], g->shard_lo[i]);
      if (rc) return rc;
    }
    return 0;
  }
  if (!g->pool)
  {
    // (the workers inherit a mask with every signal blocked: a client's handlers run on the client's threads.  A
    // thread that cannot be created must not end the process inside a C API: the calling thread then visits the
    // shards itself, now and from now on -- ADVICE r5)
    sigset_t all, old;
    sigfillset(&all);
    pthread_sigmask(SIG_BLOCK, &all, &old);
    g->pool = new pllhip_shard_pool();
    g->pool->rc.assign(n, 0);
    g->pool->msg.assign(n, std::string());
    bool ok = true;
    try
    {
      for (size_t i = 1; i < n; ++i) g->pool->threads.emplace_back(shard_worker, g, i);
    }
    catch (const std::system_error &)
    {
      ok = false;
    }
    pthread_sigmask(SIG_SETMASK, &old, nullptr);
    if (!ok)
    {
      pool_destroy(g);
      g->shard_threads = false;
      return pllhip_group_parallel(g, fn);
    }
  }
  pllhip_shard_pool & p = *g->pool;
  {
    std::lock_guard<std::mutex> lk(p.m);
    p.job = &fn;
    p.remaining.store((unsigned int)(n - 1), std::memory_order_release);
    p.generation.fetch_add(1, std::memory_order_acq_rel);
  }
  p.cv.notify_all();
  int rc0 = 0;
  if (hipSetDevice(g->shards[0]->sh.device) != hipSuccess) { pllhip_set_error("hipSetDevice"); rc0 = -1; }
  else rc0 = fn(g->shards[0], g->shard_lo[0]);
  // (everybody is through before `fn` -- the caller's lambda, its captures on the caller's stack -- goes away)
  for (unsigned int spins = 0; p.remaining.load(std::memory_order_acquire) != 0; ++spins)
  {
    // (the others enqueue a few launches each: microseconds; a shard that takes longer -- a first call that allocates,
    // a stream it must drain -- is waited for without a core)
    if (spins < 20000u)
    {
#if defined(__x86_64__) || defined(__i386__)
      __builtin_ia32_pause();
#endif
    }
    else std::this_thread::yield();
  }
  if (rc0) return rc0;
  for (size_t i = 1; i < n; ++i)
    if (p.rc[i])
    {
      pllhip_set_error("%s", p.msg[i].c_str());
      return p.rc[i];
    }
  return 0;
}

static void pool_destroy(pllhip_ctx * g)
{
  if (!g->pool) return;
  {
    std::lock_guard<std::mutex> lk(g->pool->m);
    g->pool->stop = true;
  }
  g->pool->cv.notify_all();
  for (std::thread & t : g->pool->threads) t.join();
  delete g->pool;
  g->pool = nullptr;
}

extern "C" int pllhip_ctx_create_sharded(const pllhip_shape_t * shape, const int * devices,
                                         unsigned int ndevices, pllhip_ctx_t ** out)
{
  *out = nullptr;
  if (!shape || !devices || ndevices < 1 || shape->sites <= shape->asc_states)
  {
    pllhip_set_error("pllhip_ctx_create_sharded: bad arguments");
    return -1;
  }
  // contiguous ranges of the ordinary sites, boundaries on multiples of 256 sites; fewer
  // shards than devices when the partition is too small to give everyone a range
  const size_t ordinary = shape->sites - shape->asc_states;
  size_t per = (ordinary + ndevices - 1) / ndevices;
  per = (per + 255) / 256 * 256;
  std::vector<size_t> lo;
  for (size_t b = 0; b < ordinary; b += per) lo.push_back(b);
  pllhip_ctx * g = new pllhip_ctx();
  // (read when the group is created, like every other switch)
  g->shard_threads = !(pllhip_env("PLLHIP_SHARD_THREADS") && atoi(pllhip_env("PLLHIP_SHARD_THREADS")) == 0);
  const bool shard_poll = !(pllhip_env("PLLHIP_SHARD_POLL") && atoi(pllhip_env("PLLHIP_SHARD_POLL")) == 0);
  g->sh = *shape;
  g->sh.device = devices[0];
  g->span = (size_t)shape->states * shape->rate_cats;
  g->clv_elems = (size_t)shape->sites * g->span;
  g->scaler_elems = shape->rate_scalers ? (size_t)shape->sites * shape->rate_cats : shape->sites;
  for (size_t i = 0; i < lo.size(); ++i)
  {
    const bool last = i + 1 == lo.size();
    pllhip_shape_t sh = *shape;
    sh.device = devices[i];
    sh.sites = (unsigned int)((last ? ordinary : lo[i + 1]) - lo[i]) + (last ? shape->asc_states : 0u);
    sh.asc_states = last ? shape->asc_states : 0u;
    pllhip_ctx * s = nullptr;
    const int rc = pllhip_ctx_create(&sh, &s);
    if (rc)
    {
      pllhip_group_destroy(g);
      return rc;
    }
    s->shard_poll = shard_poll;
    g->shards.push_back(s);
    g->shard_lo.push_back(lo[i]);
  }
  g->shard_lo.push_back(shape->sites);
  {
    // (threads next to their devices: only when there is more than one device to be next to)
    bool distinct = false;
    for (pllhip_ctx * s : g->shards) distinct = distinct || s->sh.device != g->shards[0]->sh.device;
    g->shard_pin = distinct && !(pllhip_env("PLLHIP_SHARD_PIN") && atoi(pllhip_env("PLLHIP_SHARD_PIN")) == 0);
  }
  *out = g;
  return 0;
}

void pllhip_group_destroy(pllhip_ctx * g)
{
  pool_destroy(g);
  pllhip_device_guard guard;
  for (pllhip_ctx * s : g->shards) pllhip_ctx_destroy(s);
  g->shards.clear();
  delete g;
}

extern "C" unsigned int pllhip_shard_count(pllhip_ctx_t * c)
{
  return c->shards.empty() ? 1u : (unsigned int)c->shards.size();
}

extern "C" unsigned int pllhip_shard_first_site(pllhip_ctx_t * c, unsigned int shard)
{
  return (c->shards.empty() || shard >= c->shards.size()) ? 0u : (unsigned int)c->shard_lo[shard];
}

// Wait for a shard's enqueued result-returning call and take its sums.  Round 5: the way an unsharded context waits
// for its own -- polling the shard's word / workgroup sums in host-mapped memory (pllhip_result_wait_host; the stream
// itself only after 200 us, or when copies follow the kernel) -- instead of a hipStreamSynchronize per shard, one
// after the other, each 10-20 us of runtime whether the shard had finished or not: eight shards of 125,000 sites
// are 0.2 ms each.  PLLHIP_SHARD_POLL=0: the stream waits again.
int pllhip_result_wait_pending(pllhip_ctx * c);
static int result_wait(pllhip_ctx * s, unsigned int count, double * out)
{
  HIP_TRY(hipSetDevice(s->sh.device));
  if (s->shard_poll)
  {
    const int rc = pllhip_result_wait_pending(s);
    if (rc) return rc;
  }
  else
  {
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->pending_hostsum)
    {
      const int rc = pllhip_result_wait_pending(s); // (the stream is drained: this only adds the sums)
      if (rc) return rc;
    }
  }
  for (unsigned int i = 0; i < count; ++i) out[i] = s->h_result[i];
  return 0;
}

namespace
{
// enqueue on every shard (no waiting in between: the devices run side by side), then collect
template <typename Enqueue>
int fan_out_and_sum(pllhip_ctx * g, unsigned int ncomp, double * sums, Enqueue enqueue)
{
  pllhip_device_guard guard; // (the caller's current device is put back on return)
  for (;;)
  {
    const size_t n = g->shards.size();
    std::vector<int> enq(n, 0);
    std::vector<std::string> enq_msg(n);
    int rc = pllhip_group_parallel(g, [&](pllhip_ctx * s, size_t lo) -> int {
      // (a shard that fails before it has enqueued a result must not be waited for as if it had: the previous
      // call's sequence number and "host adds the sums" would be polled -- ADVICE r5)
      s->pending_seq = 0;
      s->pending_hostsum = s->pending_spin = s->pending_stream_work = false;
      s->defer = true;
      const int r = enqueue(s, lo);
      s->defer = false;
      size_t i = 0;
      while (i + 1 < n && g->shards[i] != s) ++i;
      enq[i] = r;
      if (r) enq_msg[i] = pllhip_last_error();
      return r;
    });
    for (unsigned int k = 0; k < ncomp; ++k) sums[k] = 0.0;
    // (also after a failure: nothing may still be writing into the caller's per-site buffer)
    for (size_t i = 0; i < n; ++i)
    {
      pllhip_ctx * s = g->shards[i];
      if (enq[i])
      {
        // nothing of this call's to collect from this shard: its stream only, and the first error's text is kept
        if (hipSetDevice(s->sh.device) == hipSuccess) (void)hipStreamSynchronize(s->stream);
        if (!rc) rc = enq[i];
        continue;
      }
      double v[2] = {0.0, 0.0};
      const int rw = result_wait(s, ncomp, v);
      if (!rc) rc = rw;
      for (unsigned int k = 0; k < ncomp; ++k) sums[k] += v[k];
    }
    if (rc)
    {
      for (size_t i = 0; i < n; ++i)
        if (enq[i] && !enq_msg[i].empty()) { pllhip_set_error("%s", enq_msg[i].c_str()); break; }
      return rc;
    }
    // the scaling certificate (ctx.hpp): every shard's result is in, so every shard's list is through and its flag can
    // be read; a shard that runs its list again makes the whole evaluation stale
    bool again = false;
    for (pllhip_ctx * s : g->shards)
      if (s->cert_pending)
      {
        bool r = false;
        const int rr = pllhip_cert_resolve(s, &r);
        if (rr) return rr;
        again = again || r;
      }
    if (!again) return 0;
  }
}
}

int pllhip_group_edge_loglikelihood(pllhip_ctx * g, unsigned int parent_clv, int parent_scaler,
                                    unsigned int child_clv, int child_scaler, unsigned int matrix_index,
                                    const unsigned int * h_freqs_indices, double * h_persite_lnl, double * h_lnl)
{
  double unused = 0.0;
  return fan_out_and_sum(g, 1, h_lnl, [&](pllhip_ctx * s, size_t lo) {
    return pllhip_edge_loglikelihood(s, parent_clv, parent_scaler, child_clv, child_scaler, matrix_index,
                                     h_freqs_indices, h_persite_lnl ? h_persite_lnl + lo : nullptr, &unused);
  });
}

// The reference's root kernel takes the count of site i from ENTRY i of the scale buffer (core_likelihood.c:197-198)
// -- with per-rate scale buffers (sites x rate_cats entries) that is the count of site i / rate_cats, category
// i % rate_cats: not site i's, but what the reference computes and what an unsharded partition reproduces.  A shard
// holding sites [lo, hi) therefore needs entries [lo, hi) of the WHOLE buffer, which belong to the shards holding
// sites lo / rate_cats ... hi / rate_cats: copied together here (rare call, rare attribute: every shard is waited for
// first).  Found by the sharded test on a tree that scales (round 4): the lnL had silently differed from the unsharded
// partition's whenever a scaling event had happened.
static int gather_root_counts(pllhip_ctx * g, int scaler_index)
{
  const size_t R = g->sh.rate_cats, n = g->shards.size();
  for (pllhip_ctx * s : g->shards)
  {
    HIP_TRY(hipSetDevice(s->sh.device));
    HIP_TRY(hipStreamSynchronize(s->stream));
  }
  // A shard that stores the buffer's owning CLV by class (PLL_ATTRIB_SITE_REPEATS) stores the buffer by class too:
  // its entries are not the whole buffer's then.  Such a buffer goes through the host, expanded shard by shard the
  // way pllhip_get_scaler hands it to the client (ctx.hip: shard_scaler_per_site).
  bool by_class = false;
  for (pllhip_ctx * s : g->shards)
  {
    const int owner = (size_t)scaler_index < s->scaler_owner.size() ? s->scaler_owner[scaler_index] : -1;
    if (owner >= 0 && (size_t)owner < s->rows.size() && s->rows[owner].classes) by_class = true;
  }
  std::vector<unsigned int> whole;
  if (by_class)
  {
    whole.resize((size_t)g->sh.sites * R);
    if (pllhip_get_scaler(g, (unsigned int)scaler_index, whole.data())) return -1;
  }
  for (size_t i = 0; i < n; ++i)
  {
    pllhip_ctx * s = g->shards[i];
    const size_t lo = g->shard_lo[i], hi = g->shard_lo[i + 1];
    HIP_TRY(hipSetDevice(s->sh.device));
    if (!s->root_counts)
    {
      HIP_TRY(hipMalloc((void **)&s->root_counts, ((size_t)s->sh.sites + PLLHIP_TAIL_SITES) * sizeof(unsigned int)));
      HIP_TRY(hipMemsetAsync(s->root_counts, 0, ((size_t)s->sh.sites + PLLHIP_TAIL_SITES) * sizeof(unsigned int), s->stream));
    }
    if (by_class)
    {
      HIP_TRY(hipMemcpyAsync(s->root_counts, whole.data() + lo, (hi - lo) * sizeof(unsigned int),
                             hipMemcpyHostToDevice, s->stream));
      HIP_TRY(hipStreamSynchronize(s->stream)); // (`whole` goes at the end of this function; rare call)
      s->root_scaler_override = s->root_counts;
      continue;
    }
    for (size_t t = 0; t < n; ++t)
    {
      const size_t glo = g->shard_lo[t] * R, ghi = g->shard_lo[t + 1] * R; // entries of the whole buffer shard t holds
      const size_t a = lo > glo ? lo : glo, b = hi < ghi ? hi : ghi;
      if (a >= b) continue;
      const unsigned int * src = pllhip_scaler_ptr(g->shards[t], scaler_index);
      if (!src) continue;
      HIP_TRY(hipMemcpyAsync(s->root_counts + (a - lo), src + (a - glo), (b - a) * sizeof(unsigned int),
                             hipMemcpyDeviceToDevice, s->stream));
    }
    s->root_scaler_override = s->root_counts;
  }
  return 0;
}

int pllhip_group_root_loglikelihood(pllhip_ctx * g, unsigned int clv_index, int scaler_index,
                                    const unsigned int * h_freqs_indices, double * h_persite_lnl, double * h_lnl)
{
  double unused = 0.0;
  int rc = 0;
  if (g->sh.rate_scalers && scaler_index >= 0 && scaler_index < (int)g->sh.scale_buffers)
  {
    pllhip_device_guard guard;
    // (the counts gathered below must be final: the scaling certificate of every shard's last list first)
    for (pllhip_ctx * s : g->shards)
      if (s->cert_pending && !rc) rc = pllhip_cert_resolve(s);
    if (!rc) rc = gather_root_counts(g, scaler_index);
  }
  if (!rc)
    rc = fan_out_and_sum(g, 1, h_lnl, [&](pllhip_ctx * s, size_t lo) {
      return pllhip_root_loglikelihood(s, clv_index, scaler_index, h_freqs_indices,
                                       h_persite_lnl ? h_persite_lnl + lo : nullptr, &unused);
    });
  for (pllhip_ctx * s : g->shards) s->root_scaler_override = nullptr;
  return rc;
}

int pllhip_group_likelihood_derivatives(pllhip_ctx * g, unsigned int slot, int parent_scaler, int child_scaler,
                                        const unsigned int * h_params_indices, const double * h_diagptable,
                                        double * h_d_f, double * h_dd_f)
{
  double sums[2] = {0.0, 0.0}, u0 = 0.0, u1 = 0.0;
  const int rc = fan_out_and_sum(g, 2, sums, [&](pllhip_ctx * s, size_t) {
    return pllhip_likelihood_derivatives(s, slot, parent_scaler, child_scaler, h_params_indices, h_diagptable, &u0, &u1);
  });
  *h_d_f = sums[0];
  *h_dd_f = sums[1];
  return rc;
}
