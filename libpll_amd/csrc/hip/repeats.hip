// repeats.hip -- site-repeat classes of a CLV slot, identified on the device.
//
// (An extension: the reference snapshot has no site repeats; see host/repeats.c for the
// definition and DESIGN.md 2.7.)  Two sites are in the same class at a node iff their
// rows at both children agree.  With the children's site -> row maps (or tip characters)
// already in HBM the distinct pairs are found without leaving the device:
//
//   key[n]  = row1[n] * nb + row2[n]                      (nb = rows of child 2)
//   radix sort of (key, n) over the bits the key space needs
//   head[i] = key[i] != key[i-1];  class[i] = inclusive_sum(head) - 1
//   site_id[n_i] = class[i];  at heads: lrow[class] = key / nb, rrow[class] = key % nb
//
// Classes are numbered in key order, which is as good as any: every result is per site.
// (Round 5 built the identification WITHOUT the sort -- a hash table of (key, smallest site that shows it): 64-bit
// compare-and-swap with linear probing, atomicMin of the site, classes numbered by first occurrence with an inclusive
// sum in site order; correct (tests/test_gpu_repeats.py) and twice as slow: 2.4 against 1.1 ms for a four-op partial
// traversal after a subtree swap at 1 M sites, 154 against 55 ms for the first evaluation of BASELINE config 5's
// shape (profiles/r5_repeats_hash_ab.txt).  Where repeats pay there are few keys for many sites -- a million atomics
// on a few hundred addresses -- and where they do not, a million atomics on a million random lines run at a
// seventeenth of the streaming rate (MI355X_MICROARCH.md, global atomics); the radix sort streams.  Not kept.)
// The class count goes back to the host (one 4-byte copy), which needs it to size the
// launches and to decide whether the node is worth storing by class at all.
#include <hipcub/hipcub.hpp>

#include "ctx.hpp"

struct pllhip_rep_work
{
  unsigned long long * keys_in = nullptr, * keys_out = nullptr;
  unsigned int * vals_in = nullptr, * vals_out = nullptr, * cls = nullptr;
  void * temp = nullptr;
  size_t temp_bytes = 0;
  unsigned int * h_count = nullptr; // pinned
};

__global__ __launch_bounds__(256) void k_rep_keys(const unsigned int * __restrict__ id1,
                                                  const unsigned char * __restrict__ tip1,
                                                  const unsigned int * __restrict__ id2,
                                                  const unsigned char * __restrict__ tip2,
                                                  unsigned int nb, unsigned int tip_mask,
                                                  unsigned int sites,
                                                  unsigned long long * __restrict__ keys,
                                                  unsigned int * __restrict__ vals)
{
  for (size_t n = blockIdx.x * (size_t)blockDim.x + threadIdx.x; n < sites;
       n += (size_t)gridDim.x * blockDim.x)
  {
    const unsigned int a = tip1 ? (tip1[n] & tip_mask) : id1[n];
    const unsigned int b = tip2 ? (tip2[n] & tip_mask) : id2[n];
    keys[n] = (unsigned long long)a * nb + b;
    vals[n] = (unsigned int)n;
  }
}

__global__ __launch_bounds__(256) void k_rep_heads(const unsigned long long * __restrict__ keys,
                                                   unsigned int sites, unsigned int * __restrict__ head)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < sites;
       i += (size_t)gridDim.x * blockDim.x)
    head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

// cls holds the inclusive sums of the head flags
__global__ __launch_bounds__(256) void k_rep_scatter(const unsigned long long * __restrict__ keys,
                                                     const unsigned int * __restrict__ vals,
                                                     const unsigned int * __restrict__ cls,
                                                     unsigned int nb, unsigned int sites,
                                                     unsigned int * __restrict__ site_id,
                                                     unsigned int * __restrict__ lrow,
                                                     unsigned int * __restrict__ rrow)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < sites;
       i += (size_t)gridDim.x * blockDim.x)
  {
    const unsigned int c = cls[i] - 1u;
    site_id[vals[i]] = c;
    if (i == 0 || keys[i] != keys[i - 1])
    {
      lrow[c] = (unsigned int)(keys[i] / nb);
      rrow[c] = (unsigned int)(keys[i] % nb);
    }
  }
}

static int rep_work(pllhip_ctx * c, pllhip_rep_work ** out)
{
  if (c->rep_work)
  {
    *out = c->rep_work;
    return 0;
  }
  pllhip_rep_work * w = new pllhip_rep_work();
  const size_t N = c->sh.sites;
  HIP_TRY(hipMalloc((void **)&w->keys_in, N * sizeof(unsigned long long)));
  HIP_TRY(hipMalloc((void **)&w->keys_out, N * sizeof(unsigned long long)));
  HIP_TRY(hipMalloc((void **)&w->vals_in, N * sizeof(unsigned int)));
  HIP_TRY(hipMalloc((void **)&w->vals_out, N * sizeof(unsigned int)));
  HIP_TRY(hipMalloc((void **)&w->cls, N * sizeof(unsigned int)));
  size_t sort_bytes = 0, scan_bytes = 0;
  HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, w->keys_in, w->keys_out, w->vals_in,
                                             w->vals_out, (int)N, 0, 64, c->stream));
  HIP_TRY(hipcub::DeviceScan::InclusiveSum(nullptr, scan_bytes, w->cls, w->cls, (int)N, c->stream));
  w->temp_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
  HIP_TRY(hipMalloc(&w->temp, w->temp_bytes));
  HIP_TRY(hipHostMalloc((void **)&w->h_count, sizeof(unsigned int), hipHostMallocDefault));
  c->rep_work = w;
  *out = w;
  return 0;
}

void pllhip_rep_work_free(pllhip_ctx * c)
{
  pllhip_rep_work * w = c->rep_work;
  if (!w) return;
  for (void * p : {(void *)w->keys_in, (void *)w->keys_out, (void *)w->vals_in, (void *)w->vals_out,
                   (void *)w->cls, w->temp})
    if (p) (void)hipFree(p);
  if (w->h_count) (void)hipHostFree(w->h_count);
  delete w;
  c->rep_work = nullptr;
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
  const unsigned int grid = pllhip_stream_grid(c, N, 256);
  k_rep_keys<<<grid, 256, 0, c->stream>>>(t1 ? nullptr : c->rows[child1].site_id,
                                          t1 ? pllhip_tip_ptr(c, child1) : nullptr,
                                          t2 ? nullptr : c->rows[child2].site_id,
                                          t2 ? pllhip_tip_ptr(c, child2) : nullptr, nb, tip_rows - 1u, N,
                                          w->keys_in, w->vals_in);
  HIP_TRY(hipGetLastError());
  // only the bits the key space needs are sorted: a cherry (16 x 16) is one 8-bit pass
  const unsigned long long space = (unsigned long long)na * nb;
  int bits = 1;
  while (bits < 64 && (space - 1) >> bits) ++bits;
  size_t tb = w->temp_bytes;
  HIP_TRY(hipcub::DeviceRadixSort::SortPairs(w->temp, tb, w->keys_in, w->keys_out, w->vals_in,
                                             w->vals_out, (int)N, 0, bits, c->stream));
  k_rep_heads<<<grid, 256, 0, c->stream>>>(w->keys_out, N, w->cls);
  HIP_TRY(hipGetLastError());
  tb = w->temp_bytes;
  HIP_TRY(hipcub::DeviceScan::InclusiveSum(w->temp, tb, w->cls, w->cls, (int)N, c->stream));
  HIP_TRY(hipMemcpyAsync(w->h_count, w->cls + (N - 1), sizeof(unsigned int), hipMemcpyDeviceToHost,
                         c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  const unsigned int classes = *w->h_count;
  if (classes > max_classes) return 0;

  const size_t slack = PLLHIP_TAIL_SITES;
  if (!r.site_id)
  {
    HIP_TRY(hipMalloc((void **)&r.site_id, ((size_t)N + slack) * sizeof(unsigned int)));
    HIP_TRY(hipMemsetAsync(r.site_id, 0, ((size_t)N + slack) * sizeof(unsigned int), c->stream));
  }
  if (r.row_cap < classes)
  {
    if (r.lrow) HIP_TRY(hipFree(r.lrow)); // (the stream is idle: synchronised above)
    if (r.rrow) HIP_TRY(hipFree(r.rrow));
    r.lrow = r.rrow = nullptr;
    HIP_TRY(hipMalloc((void **)&r.lrow, (classes + slack) * sizeof(unsigned int)));
    HIP_TRY(hipMalloc((void **)&r.rrow, (classes + slack) * sizeof(unsigned int)));
    r.row_cap = classes;
  }
  // the slack behind the row lists stays zero so that lanes past the last row gather row 0
  HIP_TRY(hipMemsetAsync(r.lrow, 0, (r.row_cap + slack) * sizeof(unsigned int), c->stream));
  HIP_TRY(hipMemsetAsync(r.rrow, 0, (r.row_cap + slack) * sizeof(unsigned int), c->stream));
  k_rep_scatter<<<grid, 256, 0, c->stream>>>(w->keys_out, w->vals_out, w->cls, nb, N, r.site_id,
                                             r.lrow, r.rrow);
  HIP_TRY(hipGetLastError());
  r.classes = classes;
  *classes_out = classes;
  return 0;
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
