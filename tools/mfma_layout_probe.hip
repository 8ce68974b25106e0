// developer tool: discover the operand/result lane layout of v_mfma_f64_4x4x4_4b_f64
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(int * table)
{
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb)
    {
      double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
      double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      unsigned long long m = __ballot(d != 0.0);
      if (lane == 0) table[la * 64 + lb] = m ? (__ffsll((long long)m) - 1) + 100 * (__popcll(m) - 1) : -1;
    }
}
int main()
{
  int * t, h[4096];
  hipMalloc(&t, sizeof(h));
  probe<<<1, 64>>>(t);
  hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
  for (int la = 0; la < 64; ++la) { for (int lb = 0; lb < 64; ++lb) printf("%d ", h[la * 64 + lb]); printf("\n"); }
  return 0;
}
