// xcd_balance_bench.hip -- do the eight XCDs write at the same rate?  The whole-list kernel's write pattern
// (K streams, 2 KB per tile and stream) with (a) the plain tile mapping and (b) one contiguous eighth of the tiles
// per XCD; every workgroup records when it finished (wall_clock64) and which XCD it ran on (HW_REG_XCC_ID).
//   hipcc --offload-arch=gfx950 -O3 tools/xcd_balance_bench.hip -o tools/xcd_balance_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

template <bool XCD_LOCAL>
__global__ __launch_bounds__(256, 3) void k_pattern(v2d * base, size_t stride_g, unsigned int K, size_t all_tiles, int work,
                                                    unsigned long long * t_end, unsigned int * xcc, unsigned long long * t_start)
{
  const unsigned int lane = threadIdx.x & 63u;
  const unsigned int xcd = blockIdx.x & 7u;
  size_t lo = 0, tiles = all_tiles, wave, nwaves;
  if (XCD_LOCAL)
  {
    const size_t per = (all_tiles + 7) / 8;
    lo = xcd * per;
    tiles = lo + per < all_tiles ? per : (all_tiles > lo ? all_tiles - lo : 0);
    wave = (size_t)(blockIdx.x >> 3) * 4 + (threadIdx.x >> 6);
    nwaves = (size_t)(gridDim.x >> 3) * 4;
  }
  else
  {
    wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    nwaves = (size_t)gridDim.x * 4;
  }
  if (threadIdx.x == 0) t_start[blockIdx.x] = wall_clock64();
  for (size_t t = wave; t < tiles; t += nwaves)
  {
    v2d acc = {(double)t, (double)lane};
    for (unsigned int k = 0; k < K; ++k)
    {
      for (int w = 0; w < work; ++w) acc = acc * 1.0000001 + 0.5;
      v2d * out = base + (size_t)k * stride_g + (lo + t) * 128;
      __builtin_nontemporal_store(acc, out + lane);
      __builtin_nontemporal_store(acc, out + 64 + lane);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    t_end[blockIdx.x] = wall_clock64();
    unsigned int id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[blockIdx.x] = id & 0xf;
  }
}

// one pass: as many workgroups as it takes, each wave ONE tile position (all K streams): does the
// dispatcher hand a finished XCD new workgroups, or does every XCD get blockIdx % 8 whatever happens?
__global__ __launch_bounds__(256, 3) void k_one_pass(v2d * base, size_t stride_g, unsigned int K, size_t all_tiles,
                                                     unsigned long long * per_xcd_last, unsigned int * per_xcd_count)
{
  const unsigned int lane = threadIdx.x & 63u;
  const size_t t = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t < all_tiles)
  {
    v2d acc = {(double)t, (double)lane};
    for (unsigned int k = 0; k < K; ++k)
    {
      v2d * out = base + (size_t)k * stride_g + t * 128;
      __builtin_nontemporal_store(acc, out + lane);
      __builtin_nontemporal_store(acc, out + 64 + lane);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    unsigned int id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    id &= 0xf;
    atomicMax(per_xcd_last + id, wall_clock64());
    atomicAdd(per_xcd_count + id, 1u);
  }
}

int main()
{
  const unsigned int grid = 768;
  unsigned long long * d_end, * d_start;
  unsigned int * d_xcc;
  CK(hipMalloc((void **)&d_end, grid * 8)); CK(hipMalloc((void **)&d_start, grid * 8)); CK(hipMalloc((void **)&d_xcc, grid * 4));
  struct { unsigned int K; size_t sites; } shapes[] = {{62, 1000000}, {126, 1000000}};
  for (auto & s : shapes)
  {
    const size_t stride_g = (s.sites + 64) * 8, total = stride_g * s.K * 16;
    v2d * d;
    CK(hipMalloc((void **)&d, total + (4 << 20)));
    CK(hipMemset(d, 0, total));
    for (int local = 0; local < 2; ++local)
      for (int work : {0, 30})
      {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep)
        {
          CK(hipEventRecord(e0));
          if (local) k_pattern<true><<<grid, 256>>>(d, stride_g, s.K, s.sites / 16, work, d_end, d_xcc, d_start);
          else k_pattern<false><<<grid, 256>>>(d, stride_g, s.K, s.sites / 16, work, d_end, d_xcc, d_start);
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          if (rep >= 2 && ms < best) best = ms;
        }
        std::vector<unsigned long long> te(grid), ts(grid); std::vector<unsigned int> x(grid);
        CK(hipMemcpy(te.data(), d_end, grid * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ts.data(), d_start, grid * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(x.data(), d_xcc, grid * 4, hipMemcpyDeviceToHost));
        const unsigned long long t0 = *std::min_element(ts.begin(), ts.end());
        double last[16] = {0}, mean[16] = {0}; int n[16] = {0}; int agree = 0;
        for (unsigned int b = 0; b < grid; ++b)
        {
          const double us = (double)(te[b] - t0) / 100.0; // wall_clock64: 100 MHz
          last[x[b]] = std::max(last[x[b]], us); mean[x[b]] += us; n[x[b]]++;
          agree += (x[b] == (b & 7u));
        }
        const double bytes = (double)s.K * (s.sites / 16) * 2048;
        printf("K %3u, %s mapping, work %2d: %8.1f us  %5.2f TB/s   blockIdx%%8 == XCC_ID for %d of %u workgroups\n", s.K,
               local ? "XCD-local" : "plain    ", work, best * 1e3, bytes / (best * 1e-3) / 1e12, agree, grid);
        printf("   per XCD: last workgroup done at (us):");
        for (int i = 0; i < 8; ++i) printf(" %7.1f", last[i]);
        printf("\n   per XCD: mean finish (us)           :");
        for (int i = 0; i < 8; ++i) printf(" %7.1f", n[i] ? mean[i] / n[i] : 0.0);
        printf("\n");
      }
    {
      unsigned long long * d_last; unsigned int * d_cnt;
      CK(hipMalloc((void **)&d_last, 16 * 8)); CK(hipMalloc((void **)&d_cnt, 16 * 4));
      const size_t tiles = s.sites / 16;
      const unsigned int g1 = (unsigned int)((tiles + 3) / 4);
      for (int rep = 0; rep < 3; ++rep)
      {
        CK(hipMemset(d_last, 0, 16 * 8)); CK(hipMemset(d_cnt, 0, 16 * 4));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        k_one_pass<<<g1, 256>>>(d, stride_g, s.K, tiles, d_last, d_cnt);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long last[16]; unsigned int cnt[16];
        CK(hipMemcpy(last, d_last, 16 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(cnt, d_cnt, 16 * 4, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int i = 0; i < 8; ++i) { if (last[i] < t0) t0 = last[i]; if (last[i] > t1) t1 = last[i]; }
        const double bytes = (double)s.K * tiles * 2048;
        printf("K %3u, ONE PASS (%u workgroups): %8.1f us  %5.2f TB/s; workgroups per XCD:", s.K, g1, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
        for (int i = 0; i < 8; ++i) printf(" %u", cnt[i]);
        printf("; last finish relative to the earliest XCD (us):");
        for (int i = 0; i < 8; ++i) printf(" %.0f", (double)(last[i] - t0) / 100.0);
        printf("\n");
      }
    }
    CK(hipFree(d));
  }
  return 0;
}
