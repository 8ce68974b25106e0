/* step_floor.c -- the inner loop of SPR scoring / branch-length optimisation from C (no Python in the loop;
 * reference shape: test/src/partial-traversal.c): one branch length changes, the three ops on the path to the
 * root are redone, the edge log-likelihood is read -- pll_update_prob_matrices(1 matrix) + pll_update_partials(3
 * ops) + pll_compute_edge_loglikelihood, synchronised by the value the last call returns.  What this step costs at
 * a few thousand sites is launches and the result wait, not bytes.
 *   gcc -O2 tools/step_floor.c -Iinclude -Llibpll_amd -lpll_amd -Wl,-rpath,$PWD/libpll_amd -lm -o tools/step_floor.bin
 *   tools/step_floor.bin <states 4|20> <sites> [ops on the path, default 3]
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
  const unsigned int sites = argc > 2 ? (unsigned int)atoi(argv[2]) : 2000;
  unsigned int path = argc > 3 ? (unsigned int)atoi(argv[3]) : 3;
  const unsigned int T = 64, R = 4;
  const int reps = 3000;
  pll_partition_t * p = pll_partition_create(T, T - 2, S, sites, 1, 2 * T - 2, R, T - 2, PLL_ATTRIB_PATTERN_TIP);
  if (!p) { printf("create failed: %s\n", pll_errmsg); return 1; }
  const double freqs[4] = {0.28, 0.22, 0.24, 0.26}, gtr[6] = {1.2, 3.1, 0.9, 1.1, 3.4, 1.0};
  double rates[4];
  unsigned int pi[4] = {0, 0, 0, 0};
  pll_set_frequencies(p, 0, S == 4 ? freqs : pll_aa_freqs_lg);
  pll_set_subst_params(p, 0, S == 4 ? gtr : pll_aa_rates_lg);
  pll_compute_gamma_cats(0.7, R, rates, PLL_GAMMA_RATES_MEAN);
  pll_set_category_rates(p, rates);
  char * seq = (char *)malloc(sites + 1);
  srand(7);
  for (unsigned int t = 0; t < T; ++t)
  {
    for (unsigned int i = 0; i < sites; ++i) seq[i] = S == 4 ? "ACGT"[rand() & 3] : "ARNDCQEGHILKMFPSTWYV"[rand() % 20];
    seq[sites] = 0;
    if (!pll_set_tip_states(p, t, S == 4 ? pll_map_nt : pll_map_aa, seq)) { printf("tip states: %s\n", pll_errmsg); return 1; }
  }
  /* balanced tree: level by level, node ids T, T+1, ...; parent_of[] for the path to the root */
  pll_operation_t ops[62];
  unsigned int level[64], n = T, nops = 0, next = T, op_of[126];
  int parent_of[126];
  for (unsigned int i = 0; i < 126; ++i) parent_of[i] = -1;
  for (unsigned int i = 0; i < T; ++i) level[i] = i;
  while (n > 2)
  {
    for (unsigned int i = 0; i < n; i += 2)
    {
      pll_operation_t * o = &ops[nops];
      op_of[next] = nops++;
      o->parent_clv_index = next;
      o->parent_scaler_index = (int)(next - T);
      o->child1_clv_index = level[i];
      o->child2_clv_index = level[i + 1];
      o->child1_matrix_index = level[i];
      o->child2_matrix_index = level[i + 1];
      o->child1_scaler_index = level[i] >= T ? (int)(level[i] - T) : PLL_SCALE_BUFFER_NONE;
      o->child2_scaler_index = level[i + 1] >= T ? (int)(level[i + 1] - T) : PLL_SCALE_BUFFER_NONE;
      parent_of[level[i]] = parent_of[level[i + 1]] = (int)next;
      level[i / 2] = next++;
    }
    n /= 2;
  }
  unsigned int mi[126];
  double bl[126];
  for (unsigned int i = 0; i < 2 * T - 2; ++i) { mi[i] = i; bl[i] = 0.05 + 0.001 * i; }
  pll_update_prob_matrices(p, pi, mi, bl, 2 * T - 2);
  const unsigned int u = level[0], v = level[1];
  pll_update_partials(p, ops, nops);
  double lnl = pll_compute_edge_loglikelihood(p, u, (int)(u - T), v, (int)(v - T), u, pi, NULL);

  /* the path: from an inner node `path` levels below the root edge upwards */
  if (path < 1) path = 1;
  if (path > 5) path = 5;
  unsigned int node = T; /* first inner node (a cherry's parent): 5 ops to the top of its side */
  for (unsigned int skip = 5; skip > path; --skip) node = (unsigned int)parent_of[node];
  pll_operation_t sub[8];
  unsigned int nsub = 0, changed = ops[op_of[node]].child1_matrix_index;
  for (int x = (int)node; x >= 0; x = parent_of[x]) sub[nsub++] = ops[op_of[x]];
  printf("%u states, %u sites, %u ops on the path (matrix %u changes)\n", S, sites, nsub, changed);

  double t_pm = 0, t_up = 0, t_lnl = 0, t0, t1, t2, t3;
  for (int i = -200; i < reps; ++i)
  {
    const double len = 0.05 + 1e-4 * (i & 1023);
    t0 = now_us();
    pll_update_prob_matrices(p, pi, &changed, &len, 1);
    t1 = now_us();
    pll_update_partials(p, sub, nsub);
    t2 = now_us();
    lnl = pll_compute_edge_loglikelihood(p, u, (int)(u - T), v, (int)(v - T), u, pi, NULL);
    t3 = now_us();
    if (i >= 0) { t_pm += t1 - t0; t_up += t2 - t1; t_lnl += t3 - t2; }
  }
  printf("  step %.1f us = pll_update_prob_matrices %.1f + pll_update_partials %.1f + pll_compute_edge_loglikelihood %.1f (returns the value)   lnL %.6f\n",
         (t_pm + t_up + t_lnl) / reps, t_pm / reps, t_up / reps, t_lnl / reps, lnl);
  /* the same without the matrix update: what the two remaining calls cost */
  t0 = now_us();
  for (int i = 0; i < reps; ++i) { pll_update_partials(p, sub, nsub); lnl = pll_compute_edge_loglikelihood(p, u, (int)(u - T), v, (int)(v - T), u, pi, NULL); }
  printf("  partials + lnL only %.1f us;", (now_us() - t0) / reps);
  t0 = now_us();
  for (int i = 0; i < reps; ++i) lnl = pll_compute_edge_loglikelihood(p, u, (int)(u - T), v, (int)(v - T), u, pi, NULL);
  printf(" lnL only %.1f us\n", (now_us() - t0) / reps);
  pll_partition_destroy(p);
  free(seq);
  return 0;
}
