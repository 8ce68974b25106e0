// repeats.hip -- site-repeat classes of a CLV slot, identified on the device.
//
// (An extension: the reference snapshot has no site repeats; see host/repeats.c for the
// definition and DESIGN.md 2.7.)  Two sites are in the same class at a node iff their
// rows at both children agree.  With the children's site -> row maps (or tip characters)
// already in HBM the distinct pairs are found without leaving the device:
//
//   key[i]  = major[n_i] << bits(minor rows) | minor[n_i]   (n_i: the sites in the major child's class order)
//   stable radix sort of (key, n) over the minor bits -- the upper part is sorted already
//   head[i] = key[i] != key[i-1];  class[i] = heads up to i, less one
//   site_id[n_i] = class[i];  perm[i] = n_i;  at heads: the two children's rows = the two parts of the key
//
// Classes are numbered in (minor, major) order, which is as good as any: every result is per site.
// (Round 5 built the identification WITHOUT the sort -- a hash table of (key, smallest site that shows it): 64-bit
// compare-and-swap with linear probing, atomicMin of the site, classes numbered by first occurrence with an inclusive
// sum in site order; correct (tests/test_gpu_repeats.py) and twice as slow: 2.4 against 1.1 ms for a four-op partial
// traversal after a subtree swap at 1 M sites, 154 against 55 ms for the first evaluation of BASELINE config 5's
// shape (profiles/r5_repeats_hash_ab.txt).  Where repeats pay there are few keys for many sites -- a million atomics
// on a few hundred addresses -- and where they do not, a million atomics on a million random lines run at a
// seventeenth of the streaming rate (MI355X_MICROARCH.md, global atomics); the radix sort streams.  Not kept.)
// The class count goes back to the host (a tagged word in host-mapped memory, polled), which needs it to size the
// launches and to decide whether the node is worth storing by class at all; nothing on the stream waits for that.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "ctx.hpp"
#include "lnl_common.hpp"

// The sort is rocPRIM's radix sort with its algorithm choice and kernel shape named here (round 5).  The library's
// default picks a MERGE sort up to 2^20 items, which compares whole keys whatever bits were asked for -- at 1 M sites
// a block sort and twenty merge passes, 155-200 us per CLV slot however few bits matter -- and its Onesweep kernels
// carry no tuning for gfx950: 35.6 us per 8-bit digit at 1 M pairs; workgroups of 1024 x 4 items 25.3 us, about the
// floor of the algorithm at this size (a chain of ~250 workgroups each waiting for its predecessor's prefix, two
// fills of the chain's state per digit).  Below 2^17 pairs the merge sort is the faster one (42 against 49 us at
// 50 k pairs).  tools/sort_config_bench.hip, profiles/r5_repeats_sort_config.txt.
using rep_sort_config = rocprim::radix_sort_config<
    rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 4>, rocprim::kernel_config<1024, 4>, 8,
                                        rocprim::block_radix_rank_algorithm::match>,
    131072>;

struct pllhip_rep_work
{
  void * keys[2] = {nullptr, nullptr};        // 8 bytes per site each (4 used while the key space fits 32 bits)
  unsigned int * vals[2] = {nullptr, nullptr};
  unsigned int * wave_heads = nullptr;       // per wave of the class-number kernels: the heads in / before its span
  void * temp = nullptr;
  size_t temp_bytes = 0;
  // the class count comes back through a host-mapped word the host polls: PLLHIP_SEQ_TAG(seq) ^ count
  unsigned long long * h_count = nullptr, * h_count_dev = nullptr;
  unsigned long long seq = 0;
};

// key of the site at position i of the starting order: (row at the MAJOR child) << minor_bits | row at the minor child.
// The starting order is the major child's sites in class order when it has one (an inner node identified before):
// the keys are then already sorted by their upper part, and a stable sort of the minor bits alone finishes the job.
template <typename K>
__global__ __launch_bounds__(256) void k_rep_keys(const unsigned int * __restrict__ start,
                                                  const unsigned int * __restrict__ start_class,
                                                  const unsigned int * __restrict__ id_major,
                                                  const unsigned char * __restrict__ tip_major,
                                                  const unsigned int * __restrict__ id_minor,
                                                  const unsigned char * __restrict__ tip_minor,
                                                  unsigned int minor_bits, unsigned int tip_mask,
                                                  unsigned int sites,
                                                  K * __restrict__ keys,
                                                  unsigned int * __restrict__ vals)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < sites;
       i += (size_t)gridDim.x * blockDim.x)
  {
    const unsigned int n = start ? start[i] : (unsigned int)i;
    const unsigned int a = start ? start_class[i] : tip_major ? (tip_major[n] & tip_mask) : id_major[n];
    const unsigned int b = tip_minor ? (tip_minor[n] & tip_mask) : id_minor[n];
    keys[i] = ((K)a << minor_bits) | b;
    vals[i] = n;
  }
}

// Class numbers from the sorted keys, in three small launches (round 5; a library scan of a flag array, a one-thread
// kernel for the count and the scatter before: 17 + 4 + 13 us per CLV slot at 1 M sites, now 4 + 3 + 13).  A head is
// a sorted position whose key differs from the one before; the class of a position is the number of heads up to it.
// Every wave owns REP_SPAN consecutive positions: k_rep_heads counts its heads, k_rep_offsets turns the counts into
// exclusive prefixes (one workgroup) and hands the total -- the class count -- to the host, k_rep_scatter walks the
// span again with a running count.  No flag array, no array of class numbers.
constexpr unsigned int REP_SPAN = 512;

template <typename K>
__global__ __launch_bounds__(256) void k_rep_heads(const K * __restrict__ keys, unsigned int sites,
                                                   unsigned int * __restrict__ wave_heads)
{
  const unsigned int lane = threadIdx.x & 63u, wave = blockIdx.x * 4u + (threadIdx.x >> 6);
  const size_t base = (size_t)wave * REP_SPAN;
  if (base >= sites) return;
  unsigned int count = 0;
#pragma unroll
  for (unsigned int r = 0; r < REP_SPAN / 64u; ++r)
  {
    const size_t i = base + r * 64u + lane;
    const bool head = i < sites && (i == 0 || keys[i] != keys[i - 1]);
    count += (unsigned int)__popcll(__ballot(head));
  }
  if (lane == 0) wave_heads[wave] = count;
}

// in place: wave_heads[w] = heads before wave w's span; the total goes to wave_heads[nwaves] and, tagged, to the
// host-mapped word
__global__ __launch_bounds__(1024) void k_rep_offsets(unsigned int * __restrict__ wave_heads, unsigned int nwaves,
                                                      unsigned long long seq, unsigned long long * __restrict__ host_word)
{
  __shared__ unsigned int s_wave[16];
  __shared__ unsigned int s_carry;
  const unsigned int lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (unsigned int first = 0; first < nwaves; first += 1024u)
  {
    const unsigned int idx = first + threadIdx.x;
    const unsigned int v = idx < nwaves ? wave_heads[idx] : 0u;
    unsigned int incl = v;
#pragma unroll
    for (unsigned int d = 1; d < 64u; d <<= 1)
    {
      const unsigned int up = __shfl_up(incl, d);
      if (lane >= d) incl += up;
    }
    if (lane == 63u) s_wave[w] = incl;
    __syncthreads();
    unsigned int before = s_carry;
    for (unsigned int k = 0; k < w; ++k) before += s_wave[k];
    if (idx < nwaves) wave_heads[idx] = before + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023u) s_carry = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0)
  {
    wave_heads[nwaves] = s_carry; // (the scatter behind this launch reads it: it does not wait for the host)
    *host_word = PLLHIP_SEQ_TAG(seq) ^ (unsigned long long)s_carry;
  }
}

// site -> class, the sites in class order with their classes (what a parent's identification starts from: read in
// order there, 16 -> 10 us for its key kernel against gathering the class of every site), per class the rows of the two
// children = the two parts of the key.  Also zeroes the `slack` entries behind the row lists, which lanes past the
// last row gather (two memsets per list before: four launches of a microsecond, 35 us apart).
template <typename K>
__global__ __launch_bounds__(256) void k_rep_scatter(const K * __restrict__ keys,
                                                     const unsigned int * __restrict__ vals,
                                                     const unsigned int * __restrict__ wave_heads,
                                                     unsigned int minor_bits, unsigned int sites,
                                                     unsigned int nwaves, unsigned int max_classes, unsigned int slack,
                                                     unsigned int * __restrict__ site_id,
                                                     unsigned int * __restrict__ perm,
                                                     unsigned int * __restrict__ perm_class,
                                                     unsigned int * __restrict__ row_major,
                                                     unsigned int * __restrict__ row_minor)
{
  // (enqueued before the host has seen the class count: a node with too many classes is given up by the host, its
  // row lists -- sized for max_classes -- are left alone here)
  const unsigned int classes = wave_heads[nwaves];
  if (classes > max_classes) return;
  const size_t first = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (first < slack)
  {
    row_major[classes + first] = 0u;
    row_minor[classes + first] = 0u;
  }
  const unsigned int lane = threadIdx.x & 63u, wave = blockIdx.x * 4u + (threadIdx.x >> 6);
  const size_t base = (size_t)wave * REP_SPAN;
  if (base >= sites) return;
  unsigned int run = wave_heads[wave];
#pragma unroll
  for (unsigned int r = 0; r < REP_SPAN / 64u; ++r)
  {
    const size_t i = base + r * 64u + lane;
    const bool in = i < sites;
    const K key = in ? keys[i] : (K)0;
    const bool head = in && (i == 0 || key != keys[i - 1]);
    const unsigned long long heads = __ballot(head);
    const unsigned int c = run + (unsigned int)__popcll(heads & ((2ull << lane) - 1ull)) - 1u;
    run += (unsigned int)__popcll(heads);
    if (in)
    {
      const unsigned int n = vals[i];
      site_id[n] = c;
      perm[i] = n;
      perm_class[i] = c;
      if (head)
      {
        row_major[c] = (unsigned int)(key >> minor_bits);
        row_minor[c] = (unsigned int)(key & (((K)1 << minor_bits) - 1));
      }
    }
  }
}

template <typename K>
static hipError_t rep_sort(pllhip_rep_work * w, size_t & bytes, bool query, unsigned int N, int bits, hipStream_t stream,
                           int * sorted_in)
{
  rocprim::double_buffer<K> keys(static_cast<K *>(w->keys[0]), static_cast<K *>(w->keys[1]));
  rocprim::double_buffer<unsigned int> vals(w->vals[0], w->vals[1]);
  const hipError_t e = rocprim::radix_sort_pairs<rep_sort_config>(query ? nullptr : w->temp, bytes, keys, vals, N, 0u,
                                                                  (unsigned int)bits, stream, false);
  if (sorted_in) *sorted_in = (keys.current() == static_cast<K *>(w->keys[1])) ? 1 : 0;
  return e;
}

static int rep_work(pllhip_ctx * c, pllhip_rep_work ** out)
{
  if (c->rep_work)
  {
    *out = c->rep_work;
    return 0;
  }
  pllhip_rep_work * w = new pllhip_rep_work();
  c->rep_work = w; // (freed with the context also when an allocation below fails)
  const size_t N = c->sh.sites;
  for (int i = 0; i < 2; ++i)
  {
    HIP_TRY(hipMalloc(&w->keys[i], N * sizeof(unsigned long long)));
    HIP_TRY(hipMalloc((void **)&w->vals[i], N * sizeof(unsigned int)));
  }
  HIP_TRY(hipMalloc((void **)&w->wave_heads, ((N + REP_SPAN - 1) / REP_SPAN + 4) * sizeof(unsigned int))); // (+ the total)
  size_t most = 0, bytes = 0;
  HIP_TRY(rep_sort<unsigned long long>(w, bytes, true, (unsigned int)N, 64, c->stream, nullptr));
  most = bytes;
  HIP_TRY(rep_sort<unsigned int>(w, bytes, true, (unsigned int)N, 32, c->stream, nullptr));
  if (bytes > most) most = bytes;
  w->temp_bytes = most;
  HIP_TRY(hipMalloc(&w->temp, w->temp_bytes));
  HIP_TRY(hipHostMalloc((void **)&w->h_count, sizeof(unsigned long long), hipHostMallocMapped));
  HIP_TRY(hipHostGetDevicePointer((void **)&w->h_count_dev, w->h_count, 0));
  *w->h_count = 0;
  *out = w;
  return 0;
}

void pllhip_rep_work_free(pllhip_ctx * c)
{
  pllhip_rep_work * w = c->rep_work;
  if (!w) return;
  for (void * p : {w->keys[0], w->keys[1], (void *)w->vals[0], (void *)w->vals[1], (void *)w->wave_heads, w->temp})
    if (p) (void)hipFree(p);
  if (w->h_count) (void)hipHostFree(w->h_count);
  delete w;
  c->rep_work = nullptr;
}

// the class count of the identification just enqueued: polled from the mapped word for a while (the chain before it
// is ~100 us of kernels at a million sites), then the stream is waited for
static int rep_wait_count(pllhip_ctx * c, pllhip_rep_work * w, unsigned int * count)
{
  const volatile unsigned long long * word = w->h_count;
  const unsigned long long tag = PLLHIP_SEQ_TAG(w->seq);
  if (!c->no_spin)
  {
    struct timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;)
    {
      for (unsigned int spins = 0; spins < 64u; ++spins)
      {
        const unsigned long long got = *word ^ tag;
        if (!(got >> 32))
        {
          __atomic_thread_fence(__ATOMIC_ACQUIRE);
          *count = (unsigned int)got;
          return 0;
        }
        PLLHIP_CPU_RELAX();
      }
      struct timespec t1;
      clock_gettime(CLOCK_MONOTONIC, &t1);
      if ((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec) > 2000000ll) break;
    }
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  const unsigned long long got = *word ^ tag;
  if (got >> 32)
  {
    pllhip_set_error("pllhip_identify_repeats: the class count has not reached host memory");
    return -1;
  }
  *count = (unsigned int)got;
  return 0;
}

// keys -> sorted -> class numbers -> count on the host; then (unless the node is given up) the maps.
// `major` is child 1 or 2 (0 / 1); the sort covers `sort_bits` low bits of the key.
template <typename K>
static int rep_identify(pllhip_ctx * c, pllhip_rep_work * w, pllhip_ctx::node_rows & r, const unsigned int child[2],
                        const bool tip[2], int major, const unsigned int * start, unsigned int minor_bits, int sort_bits,
                        unsigned int tip_rows, unsigned int max_classes, unsigned int * classes_out)
{
  const unsigned int N = c->sh.sites;
  const unsigned int grid = pllhip_stream_grid(c, N, 256);
  const int minor = 1 - major;
  k_rep_keys<K><<<grid, 256, 0, c->stream>>>(start, start ? c->rows[child[major]].perm_class : nullptr,
                                             tip[major] ? nullptr : c->rows[child[major]].site_id,
                                             tip[major] ? pllhip_tip_ptr(c, child[major]) : nullptr,
                                             tip[minor] ? nullptr : c->rows[child[minor]].site_id,
                                             tip[minor] ? pllhip_tip_ptr(c, child[minor]) : nullptr, minor_bits,
                                             tip_rows - 1u, N, static_cast<K *>(w->keys[0]), w->vals[0]);
  HIP_TRY(hipGetLastError());
  size_t tb = w->temp_bytes;
  int in = 0;
  HIP_TRY(rep_sort<K>(w, tb, false, N, sort_bits, c->stream, &in));
  const K * sorted = static_cast<const K *>(w->keys[in]);
  const unsigned int * sites_sorted = w->vals[in];
  const unsigned int nwaves = (N + REP_SPAN - 1) / REP_SPAN, wave_grid = (nwaves + 3) / 4;
  k_rep_heads<K><<<wave_grid, 256, 0, c->stream>>>(sorted, N, w->wave_heads);
  ++w->seq;
  k_rep_offsets<<<1, 1024, 0, c->stream>>>(w->wave_heads, nwaves, w->seq, w->h_count_dev);
  HIP_TRY(hipGetLastError());
  // The maps are written before the host has the class count (the wait used to sit between the two launches: 13-18 us
  // of idle stream per CLV slot), so the row lists are sized for the most classes the node may keep.
  const size_t slack = PLLHIP_TAIL_SITES;
  if (!r.site_id)
  {
    HIP_TRY(hipMalloc((void **)&r.site_id, ((size_t)N + slack) * sizeof(unsigned int)));
    HIP_TRY(hipMemsetAsync(r.site_id, 0, ((size_t)N + slack) * sizeof(unsigned int), c->stream));
    HIP_TRY(hipMalloc((void **)&r.perm, (size_t)N * sizeof(unsigned int)));
    HIP_TRY(hipMalloc((void **)&r.perm_class, (size_t)N * sizeof(unsigned int)));
  }
  if (r.row_cap < max_classes)
  {
    // (hipFree waits for the device: nothing enqueued still reads the old lists)
    if (r.lrow) HIP_TRY(hipFree(r.lrow));
    if (r.rrow) HIP_TRY(hipFree(r.rrow));
    r.lrow = r.rrow = nullptr;
    r.row_cap = 0;
    HIP_TRY(hipMalloc((void **)&r.lrow, (max_classes + slack) * sizeof(unsigned int)));
    HIP_TRY(hipMalloc((void **)&r.rrow, (max_classes + slack) * sizeof(unsigned int)));
    r.row_cap = max_classes;
  }
  // the slack behind the last row is zeroed by the scatter, so that lanes past the last row gather row 0
  k_rep_scatter<K><<<wave_grid, 256, 0, c->stream>>>(sorted, sites_sorted, w->wave_heads, minor_bits, N, nwaves, max_classes,
                                                     (unsigned int)slack, r.site_id, r.perm, r.perm_class, major == 0 ? r.lrow : r.rrow,
                                                     major == 0 ? r.rrow : r.lrow);
  HIP_TRY(hipGetLastError());
  unsigned int classes = 0;
  if (rep_wait_count(c, w, &classes)) return -1;
  if (classes > max_classes) return 0;
  r.classes = classes;
  *classes_out = classes;
  return 0;
}

extern "C" int pllhip_identify_repeats(pllhip_ctx_t * c, unsigned int parent, unsigned int child1,
                                       unsigned int child2, unsigned int max_classes,
                                       unsigned int * classes_out)
{
  if (!c->shards.empty())
  {
    // One partition over several devices (round 4): every shard identifies the classes of ITS site range -- two
    // sites are in one class iff their classes at both children agree, which never looks beyond the range -- with
    // its own limit (half its sites).  The group reports 0: the host layer treats its CLVs as stored per site (the
    // mirrors are expanded shard by shard, pllhip_get_clv); pllhip_repeats_rows counts the rows really stored.
    pllhip_device_guard guard;
    for (pllhip_ctx * s : c->shards)
    {
      unsigned int unused = 0;
      const int rc = pllhip_identify_repeats(s, parent, child1, child2, s->sh.sites / 2, &unused);
      if (rc) return rc;
    }
    *classes_out = 0;
    return 0;
  }
  HIP_TRY(hipSetDevice(c->sh.device));
  *classes_out = 0;
  const unsigned int nodes = (unsigned int)c->clv.size();
  if (parent >= nodes || child1 >= nodes || child2 >= nodes || !c->clv[parent])
  {
    pllhip_set_error("pllhip_identify_repeats: CLV index out of range");
    return -1;
  }
  if ((c->sh.states != 4 && c->sh.states != 20) || !c->sh.pattern_tip || c->sh.asc_states)
  {
    pllhip_set_error("pllhip_identify_repeats: site repeats need 4 or 20 states, pattern tips, no asc-bias sites");
    return -1;
  }
  // 20 states: only the matrix-core kernels follow row maps; on the bit-exact vector
  // kernels (PLLHIP_AA_EXACT=1, or a rate_cats / tip alphabet they do not cover) every
  // CLV simply stays stored per site
  // (category counts other than 1, 2, 4 run as several chunk launches per op, which do not gather rows either)
  if (c->sh.states == 20 &&
      (!(c->sh.rate_cats == 1 || c->sh.rate_cats == 2 || c->sh.rate_cats == 4) || !(pllhip_aa_fast_covers(c, 0) && pllhip_aa_fast_covers(c, 2) && c->maxstates <= 32)))
  {
    if (c->rows.empty()) c->rows.resize(nodes);
    c->rows[parent].classes = 0;
    return 0;
  }
  const unsigned int tip_rows = (c->sh.states == 4) ? 16u : 32u; // codes a tip can show
  if (c->rows.empty()) c->rows.resize(nodes);
  pllhip_ctx::node_rows & r = c->rows[parent];
  r.classes = 0;

  // rows of the two children: 16 (4 states) or 32 (20 states) characters at a tip, the class count of an inner node
  // stored by class; an inner node stored per site cannot carry a compression
  const bool t1 = pllhip_is_tip(c, child1), t2 = pllhip_is_tip(c, child2);
  if ((!t1 && !c->rows[child1].classes) || (!t2 && !c->rows[child2].classes)) return 0;
  const unsigned int na = t1 ? tip_rows : c->rows[child1].classes;
  const unsigned int nb = t2 ? tip_rows : c->rows[child2].classes;
  const unsigned int N = c->sh.sites;
  if (max_classes > N) max_classes = N;

  pllhip_rep_work * w;
  if (rep_work(c, &w)) return -1;
  // The key is (row at the major child) << minor_bits | (row at the minor child).  The major child is the inner one
  // with more rows: its sites in class order (kept from its own identification) are the starting order, so only the
  // minor bits are left to sort -- one 8-bit digit for an inner node and a tip, two or three for two inner nodes
  // near the root where the whole key has 35-38 bits.  A cherry sorts its whole key: 8 bits (4 states), 10 (20).
  auto bits_of = [](unsigned int rows)
  {
    int bits = 1;
    while (bits < 32 && ((rows - 1u) >> bits)) ++bits;
    return bits;
  };
  const unsigned int child[2] = {child1, child2};
  const bool tip[2] = {t1, t2};
  const unsigned int rows_of[2] = {na, nb};
  const int major = (t1 && !t2) ? 1 : (!t1 && !t2 && nb > na) ? 1 : 0;
  const unsigned int * start = tip[major] ? nullptr : c->rows[child[major]].perm;
  const int minor_bits = bits_of(rows_of[1 - major]), key_bits = minor_bits + bits_of(rows_of[major]);
  const int sort_bits = start ? minor_bits : key_bits;
  if (key_bits <= 32)
    return rep_identify<unsigned int>(c, w, r, child, tip, major, start, (unsigned int)minor_bits, sort_bits, tip_rows,
                                      max_classes, classes_out);
  return rep_identify<unsigned long long>(c, w, r, child, tip, major, start, (unsigned int)minor_bits, sort_bits,
                                          tip_rows, max_classes, classes_out);
}

// rows CLV slot `idx` is stored in, over all shards (its sites where a shard stores it per site); 0: nowhere by class
extern "C" unsigned int pllhip_repeats_rows(pllhip_ctx_t * c, unsigned int idx)
{
  if (c->shards.empty()) return (c->rows.empty() || idx >= c->rows.size()) ? 0u : c->rows[idx].classes;
  unsigned long long total = 0;
  bool any = false;
  for (pllhip_ctx * s : c->shards)
  {
    const unsigned int k = (s->rows.empty() || idx >= s->rows.size()) ? 0u : s->rows[idx].classes;
    any = any || k;
    total += k ? k : s->sh.sites;
  }
  return any ? (unsigned int)total : 0u;
}

extern "C" int pllhip_get_site_id(pllhip_ctx_t * c, unsigned int idx, unsigned int * h_site_id)
{
  if (!c->shards.empty())
  {
    pllhip_set_error("site repeats are not available to a partition sharded over several devices");
    return -1;
  }
  if (c->rows.empty() || idx >= c->rows.size() || !c->rows[idx].classes)
  {
    pllhip_set_error("pllhip_get_site_id: CLV %u is not stored by class", idx);
    return -1;
  }
  HIP_TRY(hipSetDevice(c->sh.device));
  HIP_TRY(hipMemcpyAsync(h_site_id, c->rows[idx].site_id, (size_t)c->sh.sites * sizeof(unsigned int),
                         hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}
