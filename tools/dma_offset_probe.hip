#include <hip/hip_runtime.h>
__global__ void k(const char * src, char * dst, unsigned int off)
{
  extern __shared__ char lds[];
  const unsigned int lane16 = threadIdx.x * 16u;
  const unsigned long long base = (unsigned long long)src + off;
  const unsigned int lds_b = 1024;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072"
               :: "s"(lds_b), "v"(lane16), "s"(base) : "memory", "m0");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  dst[threadIdx.x] = lds[1024 + threadIdx.x] + lds[2048 + threadIdx.x] + lds[4096 + threadIdx.x];
}
int main()
{
  char * s, * d; hipMalloc(&s, 8192); hipMalloc(&d, 64);
  char h[8192]; for (int i = 0; i < 8192; ++i) h[i] = (char)(i / 1024 + 1);
  hipMemcpy(s, h, 8192, hipMemcpyHostToDevice);
  k<<<1, 64, 16384>>>(s, d, 0);
  char r[64]; hipMemcpy(r, d, 64, hipMemcpyDeviceToHost);
  printf("sum of pieces as seen in LDS at +0 / +1024 / +3072: %d (expected 1 + 2 + 4 = 7 if the offset also moves the LDS address)\n", r[0]);
  return 0;
}
