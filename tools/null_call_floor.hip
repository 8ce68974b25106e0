// null_call_floor.hip -- what does the HIP stack charge for ONE result-returning call that does nothing?
// A one-workgroup kernel stores a sequence number to host-mapped memory (system scope); the host launches it and
// (a) spins on that word, (b) calls hipStreamSynchronize, (c) two dependent kernels then spins (a reducing kernel +
// a final-sum kernel).  The library's result-returning calls (lnL, derivatives) cannot be faster than (a).
//   hipcc --offload-arch=gfx950 -O3 tools/null_call_floor.hip -o tools/null_call_floor.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_publish(unsigned long long * host_word, unsigned long long seq)
{
  if (threadIdx.x == 0) __hip_atomic_store(host_word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_nothing(unsigned int * p) { if (threadIdx.x == 1000) *p = 1; }

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  volatile unsigned long long * word; CK(hipHostMalloc((void **)&word, 64, hipHostMallocMapped));
  unsigned long long * dword; CK(hipHostGetDevicePointer((void **)&dword, (void *)word, 0));
  unsigned int * dummy; CK(hipMalloc((void **)&dummy, 64));
  *word = 0;
  unsigned long long seq = 0;
  const int reps = 5000;
  for (int mode = 0; mode < 3; ++mode)
  {
    for (int warm = 0; warm < 2; ++warm)
    {
      const double t0 = now_us();
      for (int i = 0; i < reps; ++i)
      {
        ++seq;
        if (mode == 2) k_nothing<<<64, 256, 0, s>>>(dummy);
        k_publish<<<1, 64, 0, s>>>(dword, seq);
        if (mode == 1) CK(hipStreamSynchronize(s));
        else while (*word != seq) {}
      }
      const double t = (now_us() - t0) / reps;
      if (warm) printf("%-64s %6.2f us per call\n", mode == 0 ? "one kernel, host spins on a host-mapped word" : mode == 1 ? "one kernel, hipStreamSynchronize" : "two dependent kernels, host spins", t);
    }
  }
  return 0;
}
