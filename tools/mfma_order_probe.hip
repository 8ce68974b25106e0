// developer tool: in which order, and with how many roundings, does v_mfma_f64_4x4x4_4b_f64
// add the four products of a contraction to its accumulator?
//
// Why it matters: the reference's 20-state kernels (core_partials_avx2.c:671-750) keep four
// accumulators strided by j mod 4, each a chain of fused multiply-adds over j = m, m+4, ...,
// and add them as (a0+a1)+(a2+a3).  If one MFMA is a chain of four FMAs in k order onto C, a
// contraction chunk made of the states {m, m+4, m+8, m+12} IS the first four steps of chain m,
// and the 20-state matrix-core kernels can be bit-exact (partials_aa_mfma.hip).
//
// Layout (tools/mfma_layout_probe.hip): A lane l = A_blk[i = l&3][k = l>>4],
// B lane l = B_blk[k = l>>4][j = l&3], D lane l = D_blk[i = l>>4][j = l&3], blk = (l>>2)&3.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

__global__ void run(const double * a, const double * b, const double * c, double * d, int n)
{
  const int lane = threadIdx.x;
  for (int t = blockIdx.x; t < n; t += gridDim.x)
    d[t * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[t * 64 + lane], b[t * 64 + lane], c[t * 64 + lane], 0, 0, 0);
}

static unsigned long long rng = 0x9E3779B97F4A7C15ull;
static unsigned long long next64()
{
  unsigned long long z = (rng += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static double rnd(int spread, int base)
{
  // positive and negative, exponents spread over `spread` binades around 2^base
  const double m = 1.0 + (double)(next64() >> 11) / 9007199254740992.0;
  const int e = base + (int)(next64() % (unsigned)(2 * spread + 1)) - spread;
  return ((next64() & 1) ? -m : m) * ldexp(1.0, e);
}

int main()
{
  const int n = 20000;
  double * ha = (double *)malloc(n * 64 * 8), * hb = (double *)malloc(n * 64 * 8), * hc = (double *)malloc(n * 64 * 8),
         * hd = (double *)malloc(n * 64 * 8);
  const char * names[3] = {"normal range, exponents +-6", "tiny: products near 2^-1040 (denormal results)", "A with one nonzero per row (k = 0)"};
  for (int mode = 0; mode < 3; ++mode)
  {
    for (int t = 0; t < n * 64; ++t)
    {
      ha[t] = rnd(6, mode == 1 ? -520 : 0);
      hb[t] = rnd(6, mode == 1 ? -520 : 0);
      hc[t] = rnd(6, mode == 1 ? -1040 : 0);
      if (mode == 2 && (t & 63) >= 16) ha[t] = 0.0; // k = lane >> 4 != 0
    }
    double * da, * db, * dc, * dd;
    hipMalloc(&da, n * 64 * 8); hipMalloc(&db, n * 64 * 8); hipMalloc(&dc, n * 64 * 8); hipMalloc(&dd, n * 64 * 8);
    hipMemcpy(da, ha, n * 64 * 8, hipMemcpyHostToDevice);
    hipMemcpy(db, hb, n * 64 * 8, hipMemcpyHostToDevice);
    hipMemcpy(dc, hc, n * 64 * 8, hipMemcpyHostToDevice);
    run<<<256, 64>>>(da, db, dc, dd, n);
    hipMemcpy(hd, dd, n * 64 * 8, hipMemcpyDeviceToHost);
    // candidates: 24 orders of a chain of FMAs onto C; products summed first (in k order, fused) then + C;
    // exact sum rounded once (long double is not enough in general: counted only as a hint)
    long match_perm[24] = {0}, match_sumfirst = 0, match_ld = 0, total = 0;
    int perms[24][4], np = 0;
    for (int p0 = 0; p0 < 4; ++p0) for (int p1 = 0; p1 < 4; ++p1) for (int p2 = 0; p2 < 4; ++p2) for (int p3 = 0; p3 < 4; ++p3)
    {
      if (p0 == p1 || p0 == p2 || p0 == p3 || p1 == p2 || p1 == p3 || p2 == p3) continue;
      perms[np][0] = p0; perms[np][1] = p1; perms[np][2] = p2; perms[np][3] = p3; ++np;
    }
    for (int t = 0; t < n; ++t)
      for (int blk = 0; blk < 4; ++blk)
        for (int i = 0; i < 4; ++i)
          for (int j = 0; j < 4; ++j)
          {
            double A[4], B[4];
            for (int k = 0; k < 4; ++k)
            {
              A[k] = ha[t * 64 + (k << 4 | blk << 2 | i)];
              B[k] = hb[t * 64 + (k << 4 | blk << 2 | j)];
            }
            const int dl = i << 4 | blk << 2 | j;
            const double C = hc[t * 64 + dl], D = hd[t * 64 + dl];
            ++total;
            for (int p = 0; p < 24; ++p)
            {
              double r = C;
              for (int s = 0; s < 4; ++s) r = fma(A[perms[p][s]], B[perms[p][s]], r);
              if (memcmp(&r, &D, 8) == 0) ++match_perm[p];
            }
            double s = A[0] * B[0];
            for (int k = 1; k < 4; ++k) s = fma(A[k], B[k], s);
            s += C;
            if (memcmp(&s, &D, 8) == 0) ++match_sumfirst;
            long double e = (long double)C;
            for (int k = 0; k < 4; ++k) e += (long double)A[k] * (long double)B[k];
            const double er = (double)e;
            if (memcmp(&er, &D, 8) == 0) ++match_ld;
          }
    printf("== %s: %ld outputs\n", names[mode], total);
    for (int p = 0; p < 24; ++p)
      if (match_perm[p] * 10 > total * 9 || p == 0)
        printf("  FMA chain onto C in k order %d%d%d%d: %ld match (%.4f %%)\n", perms[p][0], perms[p][1], perms[p][2], perms[p][3],
               match_perm[p], 100.0 * match_perm[p] / total);
    printf("  products summed first, then + C: %.4f %%;  long-double sum rounded once: %.4f %%\n",
           100.0 * match_sumfirst / total, 100.0 * match_ld / total);
    hipFree(da); hipFree(db); hipFree(dc); hipFree(dd);
  }
  return 0;
}
