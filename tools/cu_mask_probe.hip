// cu_mask_probe.hip -- which CUs does a stream made with hipExtStreamCreateWithCUMask use, and do two kernels on
// complementary masks run side by side?  (1) a kernel records (XCC_ID, SE/CU from HW_ID) per workgroup for a few
// masks; (2) a write-stream kernel on `x` CUs per XCD next to a latency-bound kernel on the rest: alone vs together.
//   hipcc --offload-arch=gfx950 -O3 tools/cu_mask_probe.hip -o tools/cu_mask_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_where(unsigned int * out)
{
  if (threadIdx.x == 0)
  {
    unsigned int xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    out[blockIdx.x] = (xcc & 15u) << 16 | ((hw >> 8) & 15u) | ((hw >> 13) & 7u) << 4; // xcc | se << 4 | cu
    // keep the CU busy for a while so that workgroups spread
    unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < 200000) {}
  }
}

typedef double v2d __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_write(v2d * p, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
  {
    const v2d v = {(double)i, 1.0};
    __builtin_nontemporal_store(v, p + i);
  }
}

// a latency-bound stand-in: dependent loads from a big table
__global__ __launch_bounds__(256) void k_chase(const unsigned int * tab, unsigned int mask, unsigned int steps, unsigned int * out)
{
  unsigned int j = (blockIdx.x * 256 + threadIdx.x) & mask;
  for (unsigned int s = 0; s < steps; ++s) j = tab[j] & mask;
  if (j == 0xffffffffu) out[0] = j;
}

static float ms_between(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main()
{
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("CUs %d\n", ncu);
  unsigned int * d_out; CK(hipMalloc((void **)&d_out, 4096 * 4));
  std::vector<unsigned int> h(4096);
  // (1) where do the bits of the mask point?
  for (int variant = 0; variant < 4; ++variant)
  {
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const char * what = "";
    if (variant == 0) { for (int i = 0; i < 32; ++i) mask[0] |= 1u << i; what = "bits 0..31"; }
    if (variant == 1) { for (int i = 0; i < 256; i += 8) mask[i / 32] |= 1u << (i % 32); what = "every 8th bit"; }
    if (variant == 2) { for (int i = 0; i < 8; ++i) mask[0] |= 1u << i; what = "bits 0..7"; }
    if (variant == 3) { for (int i = 128; i < 256; ++i) mask[i / 32] |= 1u << (i % 32); what = "bits 128..255"; }
    hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, 8, mask));
    CK(hipMemsetAsync(d_out, 0xff, 4096 * 4, s));
    k_where<<<1024, 64, 0, s>>>(d_out);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), d_out, 1024 * 4, hipMemcpyDeviceToHost));
    std::set<unsigned int> places; int per_xcc[16] = {0};
    for (int i = 0; i < 1024; ++i) places.insert(h[i]);
    for (unsigned int p : places) per_xcc[(p >> 16) & 15]++;
    printf("mask %-16s -> %zu distinct (xcc, se, cu) places; per XCC:", what, places.size());
    for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
    printf("\n");
    CK(hipStreamDestroy(s));
  }
  // (2) side by side
  const size_t nw = (size_t)1 << 28; // 4 GB of double2
  v2d * d_w; CK(hipMalloc((void **)&d_w, nw * 16));
  const unsigned int tmask = (1u << 26) - 1;
  unsigned int * d_tab; CK(hipMalloc((void **)&d_tab, ((size_t)tmask + 1) * 4));
  { std::vector<unsigned int> t((size_t)tmask + 1); unsigned int x = 12345; for (auto & v : t) { x = x * 1664525u + 1013904223u; v = x; } CK(hipMemcpy(d_tab, t.data(), t.size() * 4, hipMemcpyHostToDevice)); }
  hipEvent_t ev[4]; for (int i = 0; i < 4; ++i) CK(hipEventCreate(&ev[i]));
  for (int xw : {0, 32, 64, 96})
  {
    // the write kernel on the first xw mask bits, the chase kernel on the rest
    uint32_t mw[8] = {0}, mc[8] = {0};
    for (int i = 0; i < 256; ++i)
    {
      const bool w = i < xw; // (bit i is CU i / 8 of XCC i % 8: a contiguous range is spread evenly over the XCDs)
      (w ? mw : mc)[i / 32] |= 1u << (i % 32);
    }
    hipStream_t sw, sc;
    if (xw) CK(hipExtStreamCreateWithCUMask(&sw, 8, mw)); else CK(hipStreamCreate(&sw));
    if (xw) CK(hipExtStreamCreateWithCUMask(&sc, 8, mc)); else CK(hipStreamCreate(&sc));
    const unsigned int gw = xw ? xw * 8 : 256 * 8, gc = (256 - xw) * 2;
    // alone
    CK(hipEventRecord(ev[0], sw)); k_write<<<gw, 256, 0, sw>>>(d_w, nw); CK(hipEventRecord(ev[1], sw)); CK(hipStreamSynchronize(sw));
    CK(hipEventRecord(ev[2], sc)); k_chase<<<gc, 256, 0, sc>>>(d_tab, tmask, 3000, d_out); CK(hipEventRecord(ev[3], sc)); CK(hipStreamSynchronize(sc));
    const float w_alone = ms_between(ev[0], ev[1]), c_alone = ms_between(ev[2], ev[3]);
    // together
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(ev[0], sw)); CK(hipEventRecord(ev[2], sc));
    k_chase<<<gc, 256, 0, sc>>>(d_tab, tmask, 3000, d_out);
    k_write<<<gw, 256, 0, sw>>>(d_w, nw);
    CK(hipEventRecord(ev[1], sw)); CK(hipEventRecord(ev[3], sc));
    CK(hipDeviceSynchronize());
    printf("write kernel on %3d CUs (4.3 GB): alone %.3f ms (%.2f TB/s); chase kernel on the rest alone %.3f ms; together: write %.3f, chase %.3f ms\n",
           xw ? xw : 256, w_alone, nw * 16 / (w_alone * 1e-3) / 1e12, c_alone, ms_between(ev[0], ev[1]), ms_between(ev[2], ev[3]));
    CK(hipStreamDestroy(sw)); CK(hipStreamDestroy(sc));
  }
  return 0;
}
