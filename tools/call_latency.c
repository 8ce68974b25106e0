/* call_latency.c -- fixed cost of the API entry points, measured from C (no Python in
 * the loop): a balanced 64-taxon tree over a tiny 4-state alignment.
 *   gcc -O2 tools/call_latency.c -Iinclude -Llibpll_amd -lpll_amd -Wl,-rpath,$PWD/libpll_amd -lm -o tools/call_latency.bin
 */
#include <stdio.h>
#include <stdlib.h>
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
  const unsigned int T = 64, sites = argc > 1 ? (unsigned int)atoi(argv[1]) : 1000, R = 4;
  const int reps = 2000;
  pll_partition_t * p = pll_partition_create(T, T - 2, 4, sites, 1, 2 * T - 2, R, T - 2, PLL_ATTRIB_PATTERN_TIP);
  if (!p) { printf("create failed: %s\n", pll_errmsg); return 1; }
  const double freqs[4] = {0.28, 0.22, 0.24, 0.26}, gtr[6] = {1.2, 3.1, 0.9, 1.1, 3.4, 1.0};
  double rates[4];
  unsigned int pi[4] = {0, 0, 0, 0};
  pll_set_frequencies(p, 0, freqs);
  pll_set_subst_params(p, 0, gtr);
  pll_compute_gamma_cats(0.7, R, rates, PLL_GAMMA_RATES_MEAN);
  pll_set_category_rates(p, rates);
  char * seq = (char *)malloc(sites + 1);
  srand(7);
  for (unsigned int t = 0; t < T; ++t)
  {
    for (unsigned int i = 0; i < sites; ++i) seq[i] = "ACGT"[rand() & 3];
    seq[sites] = 0;
    pll_set_tip_states(p, t, pll_map_nt, seq);
  }
  /* balanced tree: level by level, node ids T, T+1, ... */
  pll_operation_t ops[62];
  unsigned int level[64], n = T, nops = 0, next = T;
  for (unsigned int i = 0; i < T; ++i) level[i] = i;
  while (n > 2)
  {
    for (unsigned int i = 0; i < n; i += 2)
    {
      pll_operation_t * o = &ops[nops++];
      o->parent_clv_index = next;
      o->parent_scaler_index = (int)(next - T);
      o->child1_clv_index = level[i];
      o->child2_clv_index = level[i + 1];
      o->child1_matrix_index = level[i];
      o->child2_matrix_index = level[i + 1];
      o->child1_scaler_index = level[i] >= T ? (int)(level[i] - T) : PLL_SCALE_BUFFER_NONE;
      o->child2_scaler_index = level[i + 1] >= T ? (int)(level[i + 1] - T) : PLL_SCALE_BUFFER_NONE;
      level[i / 2] = next++;
    }
    n /= 2;
  }
  unsigned int mi[126];
  double bl[126];
  for (unsigned int i = 0; i < 2 * T - 2; ++i) { mi[i] = i; bl[i] = 0.05 + 0.001 * i; }
  pll_update_prob_matrices(p, pi, mi, bl, 2 * T - 2);
  const unsigned int u = level[0], v = level[1];

  double lnl = 0, t0;
  for (int i = 0; i < 50; ++i) { pll_update_partials(p, ops, nops); lnl = pll_compute_edge_loglikelihood(p, u, (int)(u - T), v, (int)(v - T), u, pi, NULL); }
  t0 = now_us();
  for (int i = 0; i < reps; ++i) pll_update_partials(p, ops, nops);
  const double issue = (now_us() - t0) / reps;
  pll_amd_wait(p);
  t0 = now_us();
  for (int i = 0; i < reps; ++i) { pll_update_partials(p, ops, nops); pll_amd_wait(p); }
  const double up_wait = (now_us() - t0) / reps;
  t0 = now_us();
  for (int i = 0; i < reps; ++i) lnl = pll_compute_edge_loglikelihood(p, u, (int)(u - T), v, (int)(v - T), u, pi, NULL);
  const double lnl_us = (now_us() - t0) / reps;
  t0 = now_us();
  for (int i = 0; i < reps; ++i) { pll_update_partials(p, ops, nops); lnl = pll_compute_edge_loglikelihood(p, u, (int)(u - T), v, (int)(v - T), u, pi, NULL); }
  const double eval_us = (now_us() - t0) / reps;
  t0 = now_us();
  for (int i = 0; i < reps; ++i) { pll_update_prob_matrices(p, pi, mi, bl, 2 * T - 2); pll_update_partials(p, ops, nops); lnl = pll_compute_edge_loglikelihood(p, u, (int)(u - T), v, (int)(v - T), u, pi, NULL); }
  const double full_us = (now_us() - t0) / reps;
  printf("sites %u: update_partials(62 ops) issue only %.1f us (back to back, stream saturated) | + wait %.1f us | "
         "edge lnL %.1f us | partials + lnL %.1f us | P-matrices + partials + lnL %.1f us | lnL %.6f\n",
         sites, issue, up_wait, lnl_us, eval_us, full_us, lnl);
  pll_partition_destroy(p);
  free(seq);
  return 0;
}
