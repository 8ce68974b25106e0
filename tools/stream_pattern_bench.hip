// stream_pattern_bench.hip -- the whole-list kernel's WRITE PATTERN on its own: K output streams
// (CLVs) of `bytes` each, one allocation; a wave takes a 2 KB tile position and writes that tile of
// every stream in turn (as k_dna_fused does, op after op), 12 waves per CU walking tiles with a
// fixed stride.  Question: does the achievable write rate depend on K x bytes (the footprint)?
//   hipcc --offload-arch=gfx950 -O3 tools/stream_pattern_bench.hip -o tools/stream_pattern_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

// READS: 0 = none; 1 = every op also gathers 16 bytes per lane from a small (L2-resident) table,
// requested one op ahead and folded into the next op's stores -- the kernel's pair-table gathers;
// PREFETCH: the first workgroup of every XCD inside a 2 MB page touches, for every stream, the
// page its XCD will write in the NEXT round (translation prefetch: the page walk happens off
// the critical path)
template <bool NT, int READS, bool PREFETCH>
__global__ __launch_bounds__(256, 3) void k_pattern(v2d * base, size_t stride_g, unsigned int K, size_t tiles, int work,
                                                    const v2d * table, unsigned int blk)
{
  const unsigned int lane = threadIdx.x & 63u;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
  for (size_t t = wave; t < tiles; t += nwaves)
  {
    v2d acc = {(double)t, (double)lane};
    v2d g_next = {0.0, 0.0};
    unsigned int sacc = 0;
    if (READS) g_next = table[((t * 37 + lane * 11) & 4095u) * 8 + (lane & 7u)];
    const bool leader = PREFETCH && (t & 1023u) < 32u && (threadIdx.x >> 6) == 0 && t + nwaves < tiles;
    for (unsigned int k = 0; k < K; ++k)
    {
      if (leader)
      {
        // a scalar load (its own path to the shared translation cache; SGPR destination, counted by the compiler)
        typedef const unsigned int __attribute__((address_space(4))) * cptr;
        const unsigned long long a = (unsigned long long)(base + (size_t)k * stride_g + (t + nwaves) * 128);
        sacc += *(cptr)(((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((unsigned int)(a >> 32)) << 32) |
                        (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((unsigned int)a));
      }
      v2d g = g_next;
      if (READS) g_next = table[(((t + k + 1) * 37 + lane * 11) & 4095u) * 8 + (lane & 7u)];
      // some arithmetic per op, like the kernel's ~250 VALU instructions
      for (int w = 0; w < work; ++w) acc = acc * 1.0000001 + 0.5;
      if (READS) acc += g;
      if (PREFETCH && sacc == 0x7fffffffu) acc += 1.0;
      // blk: the streams interleaved in chunks of blk tiles ([chunk][stream][tile in chunk]) instead of
      // one stream after the other -- what a wave writes for one tile, op after op, then lies within
      // K x blk x 2 KB (one 2 MB page for 126 streams and blk = 8)
      v2d * out = blk ? base + (((t / blk) * K + k) * blk + t % blk) * 128 : base + (size_t)k * stride_g + t * 128;
      if (NT) { __builtin_nontemporal_store(acc, out + lane); __builtin_nontemporal_store(acc, out + 64 + lane); }
      else { out[lane] = acc; out[64 + lane] = acc; }
    }
  }
}

template <bool NT>
__global__ __launch_bounds__(256) void k_linear(v2d * base, size_t n)
{
  const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x);
  for (size_t i = i0; i < n; i += (size_t)gridDim.x * 256)
  {
    const v2d v = {(double)i, 1.0};
    if (NT) __builtin_nontemporal_store(v, base + i); else base[i] = v;
  }
}

static double time_ms(void (*f)(void *), void * arg, int reps)
{
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(arg); f(arg);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) f(arg);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

struct Case { v2d * d; size_t stride_g; unsigned int K; size_t tiles; int work; bool nt; size_t n; const v2d * table; int reads, prefetch; unsigned int blk; };
static void run_pattern(void * p) { Case * c = (Case *)p; const unsigned int grid = 256 * 3;
  if (c->reads && c->prefetch) k_pattern<true, 1, true><<<grid, 256>>>(c->d, c->stride_g, c->K, c->tiles, c->work, c->table, c->blk);
  else if (c->reads) k_pattern<true, 1, false><<<grid, 256>>>(c->d, c->stride_g, c->K, c->tiles, c->work, c->table, c->blk);
  else if (c->prefetch) k_pattern<true, 0, true><<<grid, 256>>>(c->d, c->stride_g, c->K, c->tiles, c->work, c->table, c->blk);
  else if (c->nt) k_pattern<true, 0, false><<<grid, 256>>>(c->d, c->stride_g, c->K, c->tiles, c->work, c->table, c->blk);
  else k_pattern<false, 0, false><<<grid, 256>>>(c->d, c->stride_g, c->K, c->tiles, c->work, c->table, c->blk); }
static void run_linear(void * p) { Case * c = (Case *)p;
  if (c->nt) k_linear<true><<<256 * 16, 256>>>(c->d, c->n); else k_linear<false><<<256 * 16, 256>>>(c->d, c->n); }

int main(int argc, char ** argv)
{
  const int contiguous = argc > 1 && !strcmp(argv[1], "contiguous");
  // `blocked`: the big shapes only, plain layout against the streams interleaved in chunks of 8 / 64 / 512 tiles
  const int blocked = argc > 1 && !strcmp(argv[1], "blocked");
  struct { unsigned int K; size_t sites; } shapes_all[] = {{62, 1000000}, {126, 500000}, {126, 1000000}, {198, 500000}, {126, 4000000}};
  struct { unsigned int K; size_t sites; } shapes_big[] = {{126, 1000000}, {126, 4000000}, {126, 8000000}, {254, 2000000}};
  auto & shapes_ref = shapes_all;
  if (blocked)
  {
    v2d * table = nullptr;
    CK(hipMalloc((void **)&table, 4096 * 8 * 16 + 4096));
    CK(hipMemset(table, 0, 4096 * 8 * 16 + 4096));
    for (auto & s : shapes_big)
    {
      const size_t stride_g = (s.sites + 64) * 8, total = stride_g * s.K * 16;
      v2d * d = nullptr;
      if (hipMalloc((void **)&d, total + (64 << 20)) != hipSuccess) { printf("K %u sites %zu: allocation failed\n", s.K, s.sites); continue; }
      CK(hipMemset(d, 0, total));
      const double bytes = (double)s.K * (s.sites / 16) * 2048;
      for (unsigned int blk : {0u, 8u, 64u, 512u})
        for (int reads = 0; reads < 2; ++reads)
        {
          Case c = {d, stride_g, s.K, s.sites / 16 / 512 * 512, 30, true, stride_g * s.K, table, reads, 0, blk};
          const double ms = time_ms(run_pattern, &c, 5);
          printf("K %3u x %8zu sites (%5.1f GB) chunk %3u tiles reads %d: %8.1f us  %5.2f TB/s\n", s.K, s.sites, total / 1e9, blk, reads,
                 ms * 1e3, bytes / (ms * 1e-3) / 1e12);
          fflush(stdout);
        }
      CK(hipFree(d));
    }
    return 0;
  }
  auto & shapes = shapes_ref;
  v2d * table = nullptr;
  CK(hipMalloc((void **)&table, 4096 * 8 * 16 + 4096));
  CK(hipMemset(table, 0, 4096 * 8 * 16 + 4096));
  printf("allocation: %s\n", contiguous ? "hipDeviceMallocContiguous" : "hipMalloc");
  for (auto & s : shapes)
  {
    const size_t stride_g = (s.sites + 64) * 8; // 16-byte granules per stream, incl. the library's 64 sites of slack
    const size_t total = stride_g * s.K * 16;
    v2d * d = nullptr;
    hipError_t e = contiguous ? hipExtMallocWithFlags((void **)&d, total + (4 << 20), hipDeviceMallocContiguous) : hipMalloc((void **)&d, total + (4 << 20));
    if (e != hipSuccess) { printf("K %u sites %zu: allocation of %.1f GB failed: %s\n", s.K, s.sites, total / 1e9, hipGetErrorString(e)); continue; }
    CK(hipMemset(d, 0, total));
    Case c = {d, stride_g, s.K, s.sites / 16, 0, true, stride_g * s.K, table, 0, 0, 0};
    const double bytes = (double)s.K * (s.sites / 16) * 2048;
    for (int reads = 0; reads < 2; ++reads)
      for (int prefetch = 0; prefetch < 2; ++prefetch)
        for (int work : {0, 30})
        {
          c.work = work; c.reads = reads; c.prefetch = prefetch;
          const double ms = time_ms(run_pattern, &c, 8);
          printf("K %3u x %8zu sites (%5.1f GB) reads %d prefetch %d work %2d: %8.1f us  %5.2f TB/s\n", s.K, s.sites, total / 1e9,
                 reads, prefetch, work, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
          fflush(stdout);
        }
    CK(hipFree(d));
  }
  return 0;
}
