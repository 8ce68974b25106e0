/* newton_floor.c -- the inner loop of branch-length optimisation from C (no Python in the loop; reference shape:
 * examples/newton): pll_update_sumtable once, then pll_compute_likelihood_derivatives call after call, each
 * synchronised by the two values it returns; and pll_compute_edge_loglikelihood the same way.  bench.py's `newton`
 * leg times the same calls through ctypes, which adds its own microseconds per call.
 *   gcc -O2 tools/newton_floor.c -Iinclude -Llibpll_amd -lpll_amd -Wl,-rpath,$PWD/libpll_amd -lm -o tools/newton_floor.bin
 *   tools/newton_floor.bin <states 4|20> <sites>
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "pll.h"

static double now_us(void)
{
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}

int main(int argc, char ** argv)
{
  const unsigned int S = argc > 1 ? (unsigned int)atoi(argv[1]) : 4;
  const unsigned int sites = argc > 2 ? (unsigned int)atoi(argv[2]) : 500000;
  const unsigned int T = 4, R = 4;
  const int reps = 2000;
  pll_partition_t * p = pll_partition_create(T, 2, S, sites, 1, 6, R, 2, PLL_ATTRIB_PATTERN_TIP);
  if (!p) { printf("create failed: %s\n", pll_errmsg); return 1; }
  const double freqs[4] = {0.28, 0.22, 0.24, 0.26}, gtr[6] = {1.2, 3.1, 0.9, 1.1, 3.4, 1.0};
  double rates[4];
  unsigned int pi[4] = {0, 0, 0, 0};
  pll_set_frequencies(p, 0, S == 4 ? freqs : pll_aa_freqs_lg);
  pll_set_subst_params(p, 0, S == 4 ? gtr : pll_aa_rates_lg);
  pll_compute_gamma_cats(0.7, R, rates, PLL_GAMMA_RATES_MEAN);
  pll_set_category_rates(p, rates);
  char * seq = (char *)malloc((size_t)sites + 1);
  srand(7);
  for (unsigned int t = 0; t < T; ++t)
  {
    for (unsigned int i = 0; i < sites; ++i) seq[i] = S == 4 ? "ACGT"[rand() & 3] : "ARNDCQEGHILKMFPSTWYV"[rand() % 20];
    seq[sites] = 0;
    if (!pll_set_tip_states(p, t, S == 4 ? pll_map_nt : pll_map_aa, seq)) { printf("tip states: %s\n", pll_errmsg); return 1; }
  }
  pll_operation_t ops[2] = {{4, 0, 0, 0, PLL_SCALE_BUFFER_NONE, 1, 1, PLL_SCALE_BUFFER_NONE},
                            {5, 1, 2, 2, PLL_SCALE_BUFFER_NONE, 3, 3, PLL_SCALE_BUFFER_NONE}};
  unsigned int mi[6] = {0, 1, 2, 3, 4, 5};
  double bl[6] = {0.05, 0.07, 0.11, 0.13, 0.17, 0.17};
  pll_update_prob_matrices(p, pi, mi, bl, 6);
  pll_update_partials(p, ops, 2);
  double lnl = pll_compute_edge_loglikelihood(p, 4, 0, 5, 1, 4, pi, NULL);
  double * st = (double *)pll_aligned_alloc((size_t)sites * R * S * sizeof(double), PLL_ALIGNMENT_AVX);
  if (!st) { printf("no sumtable\n"); return 1; }
  pll_update_sumtable(p, 4, 5, 0, 1, pi, st);
  double d1 = 0, d2 = 0;
  for (int i = 0; i < 200; ++i) pll_compute_likelihood_derivatives(p, 0, 1, 0.1, pi, st, &d1, &d2);
  double t0 = now_us();
  for (int i = 0; i < reps; ++i) pll_compute_likelihood_derivatives(p, 0, 1, 0.05 + 0.01 * (i % 7), pi, st, &d1, &d2);
  const double t_der = (now_us() - t0) / reps;
  t0 = now_us();
  for (int i = 0; i < reps; ++i) lnl = pll_compute_edge_loglikelihood(p, 4, 0, 5, 1, 4, pi, NULL);
  const double t_lnl = (now_us() - t0) / reps;
  t0 = now_us();
  for (int i = 0; i < 200; ++i) pll_update_sumtable(p, 4, 5, 0, 1, pi, st);
  pll_compute_likelihood_derivatives(p, 0, 1, 0.1, pi, st, &d1, &d2);
  const double t_sum = (now_us() - t0) / 200;
  const double der_bytes = (8.0 * S * R + 4) * sites, lnl_bytes = (2.0 * 8 * S * R + 2 * 4 + 4) * sites;
  printf("%u states, %u sites: pll_compute_likelihood_derivatives %.1f us per call (%.2f TB/s of its %.0f MB), "
         "pll_compute_edge_loglikelihood %.1f us (%.2f TB/s of %.0f MB), pll_update_sumtable %.1f us   [d1 %.6f d2 %.6f lnL %.6f]\n",
         S, sites, t_der, der_bytes / t_der / 1e6, der_bytes / 1e6, t_lnl, lnl_bytes / t_lnl / 1e6, lnl_bytes / 1e6, t_sum, d1, d2, lnl);
  pll_aligned_free(st);
  pll_partition_destroy(p);
  free(seq);
  return 0;
}
