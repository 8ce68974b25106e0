// developer tool: rocPRIM Onesweep radix sort of (key, site) pairs under several kernel configurations, at the
// sizes repeats.hip sorts (one pair per site).   hipcc --offload-arch=gfx950 -O3 tools/sort_config_bench.hip -o tools/sort_config_bench.bin
#include <cstdio>
#include <cstring>
#include <vector>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <typename K, typename Config>
static int run(const char * name, unsigned int N, int bits)
{
  K * k[2]; unsigned int * v[2];
  for (int i = 0; i < 2; ++i) { CK(hipMalloc(&k[i], N * sizeof(K))); CK(hipMalloc(&v[i], N * 4)); }
  std::vector<K> h(N);
  unsigned long long s = 88172645463325252ull;
  for (unsigned int i = 0; i < N; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (K)(s & ((bits == 64) ? ~0ull : ((1ull << bits) - 1))); }
  size_t bytes = 0;
  { rocprim::double_buffer<K> kb(k[0], k[1]); rocprim::double_buffer<unsigned int> vb(v[0], v[1]);
    CK((rocprim::radix_sort_pairs<Config>(nullptr, bytes, kb, vb, N, 0u, (unsigned int)bits, 0, false))); }
  void * temp; CK(hipMalloc(&temp, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 6; ++rep)
  {
    CK(hipMemcpy(k[0], h.data(), N * sizeof(K), hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    rocprim::double_buffer<K> kb(k[0], k[1]); rocprim::double_buffer<unsigned int> vb(v[0], v[1]);
    CK(hipEventRecord(e0, 0));
    CK((rocprim::radix_sort_pairs<Config>(temp, bytes, kb, vb, N, 0u, (unsigned int)bits, 0, false)));
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  printf("%-34s %u-byte keys, %2d bits, %8u pairs: %7.1f us  (%.1f us per 8-bit digit)\n", name, (unsigned)sizeof(K), bits, N, best * 1e3, best * 1e3 / ((bits + 7) / 8));
  for (int i = 0; i < 2; ++i) { (void)hipFree(k[i]); (void)hipFree(v[i]); }
  (void)hipFree(temp);
  return 0;
}

using rocprim::default_config; using rocprim::kernel_config; using rocprim::radix_sort_config; using rocprim::radix_sort_onesweep_config;
template <unsigned B, unsigned I, unsigned R = 8>
using cfg = radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<B, I>, kernel_config<B, I>, R, rocprim::block_radix_rank_algorithm::match>, 0>;

template <typename K>
static int all(unsigned int N, int bits)
{
  int rc = 0;
  rc |= run<K, radix_sort_config<>>("library default (merge sort <= 2^20)", N, bits);
  rc |= run<K, radix_sort_config<default_config, default_config, default_config, 0>>("onesweep, default kernels", N, bits);
  rc |= run<K, cfg<256, 8>>("onesweep 256 x 8", N, bits);
  rc |= run<K, cfg<256, 16>>("onesweep 256 x 16", N, bits);
  rc |= run<K, cfg<512, 8>>("onesweep 512 x 8", N, bits);
  rc |= run<K, cfg<512, 16>>("onesweep 512 x 16", N, bits);
  rc |= run<K, cfg<1024, 4>>("onesweep 1024 x 4", N, bits);
  rc |= run<K, cfg<1024, 8>>("onesweep 1024 x 8", N, bits);
  rc |= run<K, cfg<256, 4>>("onesweep 256 x 4", N, bits);
  return rc;
}

int main()
{
  int rc = 0;
  rc |= all<unsigned int>(1000000u, 32);
  rc |= all<unsigned int>(1000000u, 8);
  rc |= all<unsigned long long>(1000000u, 38);
  rc |= all<unsigned int>(250000u, 32);
  rc |= all<unsigned int>(50000u, 24);
  return rc;
}
