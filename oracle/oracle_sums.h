/* oracle_sums.h -- the three summation orders (see oracle.h).  Test infrastructure. */
#ifndef ORACLE_SUMS_H_
#define ORACLE_SUMS_H_
#include <math.h>

static inline double orc_pair4(double a, double b, double c, double d)
{
  return (a + b) + (c + d);
}

/* row . vec; fused selects fmadd (AVX2 kernels) vs mul,add (AVX kernels) for
 * the 20-state strided form */
static inline double orc_dot(const double * m, const double * v, unsigned int S, int fused)
{
  unsigned int j;
  if (S == 4) return orc_pair4(m[0] * v[0], m[1] * v[1], m[2] * v[2], m[3] * v[3]);
  if (S == 20)
  {
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    for (j = 0; j < S; j += 4)
    {
      unsigned int l;
      for (l = 0; l < 4; ++l)
        a[l] = fused ? fma(m[j + l], v[j + l], a[l]) : a[l] + m[j + l] * v[j + l];
    }
    return orc_pair4(a[0], a[1], a[2], a[3]);
  }
  {
    double a = 0.0;
    for (j = 0; j < S; ++j) a += m[j] * v[j];
    return a;
  }
}

/* sum of the row entries whose state bit is set */
static inline double orc_masksum(const double * m, unsigned int mask, unsigned int S)
{
  unsigned int j;
  double a = 0.0;
  if (S == 4)
    return orc_pair4((mask & 1u) ? m[0] : 0.0, (mask & 2u) ? m[1] : 0.0,
                     (mask & 4u) ? m[2] : 0.0, (mask & 8u) ? m[3] : 0.0);
  for (j = 0; j < S; ++j)
    if ((mask >> j) & 1u) a += m[j];
  return a;
}

static inline unsigned int orc_tipmask(unsigned int states, const unsigned int * tipmap,
                                       unsigned char code)
{
  return states == 4 ? (unsigned int)code : tipmap[code];
}
#endif
