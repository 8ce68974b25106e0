/* gamma.c -- discrete-Gamma rate categories (Yang 1994).
 *
 * Replaces pll_compute_gamma_cats (gamma.c:220 of the reference).  Scalar host
 * maths, called once per alpha change: it stays on the CPU.  The published
 * algorithms are used with the reference's constants and expression order, so
 * the category rates -- inputs to every P-matrix -- come out identical:
 *   ln Gamma          Pike & Hill (1966), CACM Algorithm 291
 *   incomplete Gamma  Bhattacharjee (1970), Applied Statistics AS 32
 *   normal quantile   Odeh & Evans (1974), AS 70
 *   chi^2 quantile    Best & Roberts (1975), AS 91
 */
#include <stdio.h>

#include "internal.h"

#define GAMMA_ALPHA_MIN 0.02

static double ln_gamma(double alpha)
{
  double x = alpha, f = 0.0, z;
  if (x < 7.0)
  {
    f = 1.0;
    z = alpha - 1.0;
    while ((z = z + 1.0) < 7.0) f *= z;
    x = z;
    f = -log(f);
  }
  z = 1 / (x * x);
  return f + (x - 0.5) * log(x) - x + .918938533204673 +
         (((-.000595238095238 * z + .000793650793651) * z - .002777777777778) * z +
          .083333333333333) / x;
}

/* regularised lower incomplete gamma P(alpha, x); -1 on bad arguments */
static double incomplete_gamma(double x, double alpha, double ln_gamma_alpha)
{
  const double accurate = 1e-8, overflow = 1e30;
  const double p = alpha, g = ln_gamma_alpha;
  double factor, gin, rn, term;
  int i;

  if (x == 0) return 0;
  if (x < 0 || p <= 0) return -1;
  factor = exp(p * log(x) - x - g);

  if (!(x > 1 && x >= p))
  {
    /* series expansion */
    gin = 1;
    term = 1;
    rn = p;
    do
    {
      rn++;
      term *= x / rn;
      gin += term;
    } while (term > accurate);
    gin *= factor / p;
    return gin;
  }
  else
  {
    /* continued fraction */
    double a = 1 - p, b = a + x + 1, an, dif, pn[6];
    term = 0;
    pn[0] = 1;
    pn[1] = x;
    pn[2] = x + 1;
    pn[3] = x * b;
    gin = pn[2] / pn[3];
    for (;;)
    {
      a++;
      b += 2;
      term++;
      an = a * term;
      for (i = 0; i < 2; i++) pn[i + 4] = b * pn[i + 2] - an * pn[i];
      if (pn[5] != 0)
      {
        rn = pn[4] / pn[5];
        dif = fabs(gin - rn);
        if (!(dif > accurate) && dif <= accurate * rn) return 1 - factor * gin;
        gin = rn;
      }
      for (i = 0; i < 4; i++) pn[i] = pn[i + 2];
      if (!(fabs(pn[4]) < overflow))
        for (i = 0; i < 4; i++) pn[i] /= overflow;
    }
  }
}

static double point_normal(double prob)
{
  const double a0 = -.322232431088, a1 = -1, a2 = -.342242088547, a3 = -.0204231210245;
  const double a4 = -.453642210148e-4, b0 = .0993484626060, b1 = .588581570495;
  const double b2 = .531103462366, b3 = .103537752850, b4 = .0038560700634;
  double y, z, p = prob, p1;
  p1 = (p < 0.5 ? p : 1 - p);
  if (p1 < 1e-20) return -9999;
  y = sqrt(log(1 / (p1 * p1)));
  z = y + ((((y * a4 + a3) * y + a2) * y + a1) * y + a0) /
              ((((y * b4 + b3) * y + b2) * y + b1) * y + b0);
  return (p < 0.5 ? -z : z);
}

static double point_chi2(double prob, double v)
{
  const double e = .5e-6, aa = .6931471805, p = prob;
  double g, xx, c, ch, a, q, p1, p2, t, x, b, s1, s2, s3, s4, s5, s6;

  if (p < .000002 || p > .999998 || v <= 0) return -1;
  g = ln_gamma(v / 2);
  xx = v / 2;
  c = xx - 1;

  if (!(v >= -1.24 * log(p)))
  {
    ch = pow((p * xx * exp(g + xx * aa)), 1 / xx);
    if (ch - e < 0) return ch;
  }
  else if (v > .32)
  {
    x = point_normal(p);
    p1 = 0.222222 / v;
    ch = v * pow((x * sqrt(p1) + 1 - p1), 3.0);
    if (ch > 2.2 * v + 6) ch = -2 * (log(1 - p) - c * log(.5 * ch) + g);
  }
  else
  {
    ch = 0.4;
    a = log(1 - p);
    do
    {
      q = ch;
      p1 = 1 + ch * (4.67 + ch);
      p2 = ch * (6.73 + ch * (6.66 + ch));
      t = -0.5 + (4.67 + 2 * ch) / p1 - (6.73 + ch * (13.32 + 3 * ch)) / p2;
      ch -= (1 - exp(a + g + .5 * ch + c * aa) * p2 / p1) / t;
    } while (!(fabs(q / ch - 1) - .01 <= 0));
  }

  do
  {
    q = ch;
    p1 = .5 * ch;
    if ((t = incomplete_gamma(p1, xx, g)) < 0.0) return -1;
    p2 = p - t;
    t = p2 * exp(xx * aa + g + p1 - c * log(ch));
    b = t / ch;
    a = 0.5 * t - b * c;
    s1 = (210 + a * (140 + a * (105 + a * (84 + a * (70 + 60 * a))))) / 420;
    s2 = (420 + a * (735 + a * (966 + a * (1141 + 1278 * a)))) / 2520;
    s3 = (210 + a * (462 + a * (707 + 932 * a))) / 2520;
    s4 = (252 + a * (672 + 1182 * a) + c * (294 + a * (889 + 1740 * a))) / 5040;
    s5 = (84 + 264 * a + c * (175 + 606 * a)) / 2520;
    s6 = (120 + c * (346 + 127 * c)) / 5040;
    ch += t * (1 + 0.5 * t * s1 - b * c * (s1 - b * (s2 - b * (s3 - b * (s4 - b * (s5 - b * s6))))));
  } while (fabs(q / ch - 1) > e);
  return ch;
}

static double point_gamma(double prob, double alpha, double beta)
{
  return point_chi2(prob, 2.0 * (alpha)) / (2.0 * (beta));
}

int pll_compute_gamma_cats(double alpha, unsigned int categories, double * rates, int mode)
{
  unsigned int i;
  const double factor = alpha / alpha * categories;
  const double beta = alpha;

  if (alpha < GAMMA_ALPHA_MIN || categories < 1)
  {
    pll_amd_set_error(PLL_ERROR_PARAM_INVALID, "Invalid alpha value (%f)", alpha);
    return PLL_FAILURE;
  }
  if (categories == 1)
  {
    rates[0] = 1.0;
    return PLL_SUCCESS;
  }
  if (mode == PLL_GAMMA_RATES_MEDIAN)
  {
    const double middle = 1.0 / (2.0 * categories);
    double t = 0.0;
    for (i = 0; i < categories; i++)
      rates[i] = point_gamma((double)(i * 2 + 1) * middle, alpha, beta);
    for (i = 0; i < categories; i++) t += rates[i];
    for (i = 0; i < categories; i++) rates[i] *= factor / t;
    return PLL_SUCCESS;
  }
  if (mode == PLL_GAMMA_RATES_MEAN)
  {
    const double lnga1 = ln_gamma(alpha + 1);
    double * cut = (double *)malloc(categories * sizeof(double));
    if (!cut)
    {
      pll_amd_set_error(PLL_ERROR_MEM_ALLOC, "Unable to allocate enough memory.");
      return PLL_FAILURE;
    }
    for (i = 0; i < categories - 1; i++)
      cut[i] = point_gamma((i + 1.0) / categories, alpha, beta);
    for (i = 0; i < categories - 1; i++)
      cut[i] = incomplete_gamma(cut[i] * beta, alpha + 1, lnga1);
    rates[0] = cut[0] * factor;
    rates[categories - 1] = (1 - cut[categories - 2]) * factor;
    for (i = 1; i < categories - 1; i++) rates[i] = (cut[i] - cut[i - 1]) * factor;
    free(cut);
    return PLL_SUCCESS;
  }
  pll_amd_set_error(PLL_ERROR_PARAM_INVALID, "Invalid GAMMA discretization mode (%d)", mode);
  return PLL_FAILURE;
}
