// check_expm1.cpp -- TEST INFRASTRUCTURE.  Compiles the DEVICE expm1 restatement
// (libpll_amd/csrc/hip/numerics.hpp) for the host and compares it bit for bit
// with the C library's expm1 the reference calls (core_pmatrix.c:189-199).
// Built and run by tests/test_host.py; usage: check_expm1 <count>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

// host stand-ins for the few HIP spellings numerics.hpp uses
#define __device__
#define __forceinline__ inline
#define __restrict__
#define PLLHIP_NUMERICS_HOST_BUILD 1
static inline long long __double_as_longlong(double x) { long long v; memcpy(&v, &x, 8); return v; }
static inline double __longlong_as_double(long long v) { double x; memcpy(&x, &v, 8); return x; }
#include "numerics.hpp"

int main(int argc, char ** argv)
{
  const long n = argc > 1 ? atol(argv[1]) : 1000000;
  long bad = 0;
  srand48(12345);
  for (long i = 0; i < n; ++i)
  {
    const double u = drand48();
    double x;
    switch (i % 6)
    {
      case 0: x = -u * 50; break;            // typical lambda*r*t
      case 1: x = -exp(-40 * u); break;      // tiny arguments (the expm1 trick)
      case 2: x = -u * 800; break;           // saturating to -1
      case 3: x = (u - 0.5) * 4; break;      // around the reduction thresholds
      case 4: x = u * 720; break;            // positive, up to overflow
      default: x = -ldexp(u, -(int)(i % 1100)); break; // down to subnormals
    }
    const double a = pll_expm1(x), b = expm1(x);
    if (memcmp(&a, &b, 8) && !(a != a && b != b)) ++bad;
  }
  const double specials[] = {0.0, -0.0, INFINITY, -INFINITY, 709.782712893384, 709.79, -745.2};
  for (unsigned i = 0; i < sizeof(specials) / sizeof(double); ++i)
  {
    const double a = pll_expm1(specials[i]), b = expm1(specials[i]);
    if (memcmp(&a, &b, 8)) ++bad;
  }
  printf("checked %ld, mismatches %ld\n", n, bad);
  return bad ? 1 : 0;
}
