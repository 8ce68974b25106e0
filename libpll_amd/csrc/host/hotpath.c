/* hotpath.c -- the partition-level likelihood API: thin argument checks, then
 * straight into the HIP shim.
 *
 * Replaces pll_update_partials (partials.c:177), pll_compute_edge_loglikelihood
 * (likelihood.c:478), pll_compute_root_loglikelihood (likelihood.c:121),
 * pll_update_sumtable (derivatives.c:164) and
 * pll_compute_likelihood_derivatives (derivatives.c:243).  The tip-tip /
 * tip-inner / inner-inner case split the reference does here is made by the
 * shim from the same rule (clv index < tips under PLL_ATTRIB_PATTERN_TIP).
 */
#include <stdio.h>

#include "internal.h"

/* pll_operation_t and pllhip_op_t are the same eight 32-bit fields */
typedef char op_layout_check[(sizeof(pll_operation_t) == sizeof(pllhip_op_t)) ? 1 : -1];

void pll_update_partials(pll_partition_t * p, const pll_operation_t * ops, unsigned int count)
{
  pll_amd_partition_t * q = pll_amd_priv(p);
  unsigned int i;
  int rc;
  if (!pll_amd_flush_model(p)) return;
  if (q->rep)
  {
    /* site repeats: identification and execution interleave (repeats.c) */
    if (!pll_amd_repeats_update(p, ops, count)) return;
  }
  else if ((rc = pllhip_update_partials(q->ctx, (const pllhip_op_t *)ops, count)))
  {
    /* void function: errors are reported through pll_errno only, as in the reference */
    pll_amd_fail_hip(rc, "CLV update");
    return;
  }
  if (!pll_amd_mirror_mode && q->auto_mirror && !q->rep && pllhip_shard_count(q->ctx) == 1)
  {
    /* the mirrors a small partition keeps current by itself: every CLV and scale buffer the list wrote, in ONE
       launch that writes pinned host memory and one wait (the per-buffer copies below cost 25 us each: a three-op
       step 180 us instead of 19; batched: profiles/r6_auto_mirror_step_floor.txt) */
    pllhip_mirror_job_t jobs[64];
    unsigned int n = 0;
    const unsigned int nodes = p->tips + p->clv_buffers;
    const size_t clv_b = (size_t)q->sites_alloc * p->rate_cats * p->states * sizeof(double);
    const size_t sc_b = (size_t)q->sites_alloc * ((p->attributes & PLL_ATTRIB_RATE_SCALERS) ? p->rate_cats : 1) * sizeof(unsigned int);
    if (!q->clv_pinned) q->clv_pinned = (unsigned char *)calloc(nodes ? nodes : 1, 1);
    if (!q->scaler_pinned) q->scaler_pinned = (unsigned char *)calloc(p->scale_buffers ? p->scale_buffers : 1, 1);
    for (i = 0; q->clv_pinned && q->scaler_pinned && i <= count; ++i)
    {
      if (i == count || n + 2 > 64)
      {
        if (n && (rc = pllhip_mirror_batch(q->ctx, jobs, n))) { pll_amd_fail_hip(rc, "mirror copy"); return; }
        n = 0;
        if (i == count) break;
      }
      {
        const unsigned int node = ops[i].parent_clv_index;
        const int sc = ops[i].parent_scaler_index;
        if (node < nodes)
        {
          if (p->clv[node] && !q->clv_pinned[node]) { pll_amd_sync_clv(p, node); }   /* (allocated by an explicit sync earlier) */
          else
          {
            if (!p->clv[node] && (p->clv[node] = (double *)pllhip_host_alloc(clv_b))) q->clv_pinned[node] = 1;
            if (p->clv[node]) { jobs[n].kind = 0; jobs[n].index = node; jobs[n].h = p->clv[node]; ++n; }
          }
        }
        if (sc != PLL_SCALE_BUFFER_NONE && (unsigned int)sc < p->scale_buffers)
        {
          if (p->scale_buffer[sc] && !q->scaler_pinned[sc]) { pll_amd_sync_scaler(p, (unsigned int)sc); }
          else
          {
            if (!p->scale_buffer[sc] && (p->scale_buffer[sc] = (unsigned int *)pllhip_host_alloc(sc_b))) q->scaler_pinned[sc] = 1;
            if (p->scale_buffer[sc]) { jobs[n].kind = 1; jobs[n].index = (unsigned int)sc; jobs[n].h = p->scale_buffer[sc]; ++n; }
          }
        }
      }
    }
  }
  else if (PLL_AMD_MIRRORS(p))
    for (i = 0; i < count; ++i)
    {
      pll_amd_sync_clv(p, ops[i].parent_clv_index);
      if (ops[i].parent_scaler_index != PLL_SCALE_BUFFER_NONE)
        pll_amd_sync_scaler(p, (unsigned int)ops[i].parent_scaler_index);
    }
}

double pll_compute_edge_loglikelihood(pll_partition_t * p, unsigned int parent_clv_index,
                                      int parent_scaler_index, unsigned int child_clv_index,
                                      int child_scaler_index, unsigned int matrix_index,
                                      const unsigned int * freqs_indices, double * persite_lnl)
{
  double lnl = -INFINITY;
  int rc;
  if (!pll_amd_flush_model(p)) return -INFINITY;
  if (!pll_amd_repeats_scaler_ok(p, parent_clv_index, parent_scaler_index) ||
      !pll_amd_repeats_scaler_ok(p, child_clv_index, child_scaler_index))
    return -INFINITY;
  rc = pllhip_edge_loglikelihood(pll_amd_priv(p)->ctx, parent_clv_index, parent_scaler_index,
                                 child_clv_index, child_scaler_index, matrix_index,
                                 freqs_indices, persite_lnl, &lnl);
  if (rc)
  {
    pll_amd_fail_hip(rc, "edge log-likelihood");
    return -INFINITY;
  }
  return lnl;
}

double pll_compute_root_loglikelihood(pll_partition_t * p, unsigned int clv_index,
                                      int scaler_index, const unsigned int * freqs_indices,
                                      double * persite_lnl)
{
  double lnl = -INFINITY;
  int rc;
  if (!pll_amd_flush_model(p)) return -INFINITY;
  if (!pll_amd_repeats_scaler_ok(p, clv_index, scaler_index)) return -INFINITY;
  rc = pllhip_root_loglikelihood(pll_amd_priv(p)->ctx, clv_index, scaler_index, freqs_indices,
                                 persite_lnl, &lnl);
  if (rc)
  {
    pll_amd_fail_hip(rc, "root log-likelihood");
    return -INFINITY;
  }
  return lnl;
}

/* The caller's sumtable pointer is only a KEY here: the table itself lives in a device
 * slot.  Slots are handed out as tables appear (one per live host buffer, e.g. one per
 * branch), up to the context's budget; beyond it the least recently used slot is recycled
 * and its key remembered: the host buffer of such a table was never written (outside
 * mirror mode), so using it again must fail loudly, not upload garbage. */
static int slot_of(pll_amd_partition_t * q, const double * key)
{
  unsigned int s;
  for (s = 0; s < q->sumtable_used; ++s)
    if (q->sumtable_key[s] == key)
    {
      q->sumtable_stamp[s] = ++q->sumtable_clock;
      return (int)s;
    }
  return -1;
}

static int was_evicted(const pll_amd_partition_t * q, const double * key)
{
  unsigned int i;
  for (i = 0; i < q->sumtable_evicted_n; ++i)
    if (key && q->sumtable_evicted[i] == key) return 1;
  return 0;
}

static void evicted_remove(pll_amd_partition_t * q, const double * key)
{
  unsigned int i;
  for (i = 0; i < q->sumtable_evicted_n; ++i)
    if (q->sumtable_evicted[i] == key) q->sumtable_evicted[i--] = q->sumtable_evicted[--q->sumtable_evicted_n];
}

static void evicted_add(pll_amd_partition_t * q, const double * key)
{
  if (was_evicted(q, key)) return;
  if (q->sumtable_evicted_n == q->sumtable_evicted_cap)
  {
    const unsigned int cap = q->sumtable_evicted_cap ? 2 * q->sumtable_evicted_cap : 64;
    const double ** grown = (const double **)realloc((void *)q->sumtable_evicted, cap * sizeof(*grown));
    if (!grown) return; /* (out of memory: the guard loses a key, nothing else) */
    q->sumtable_evicted = grown;
    q->sumtable_evicted_cap = cap;
  }
  q->sumtable_evicted[q->sumtable_evicted_n++] = key;
}

/* a slot for a table that has none yet */
static int slot_assign(pll_amd_partition_t * q, const double * key)
{
  unsigned int s, i;
  int freed = -1;
  if (!q->sumtable_cap) q->sumtable_cap = pllhip_sumtable_budget(q->ctx);
  evicted_remove(q, key); /* it is alive again */
  /* a slot the client gave back (pll_amd_forget_sumtable) before a new one: its device buffer is
     there already (allocate / forget / free in a loop grew device memory up to the budget) */
  for (i = 0; i < q->sumtable_used && freed < 0; ++i)
    if (!q->sumtable_key[i]) freed = (int)i;
  if (freed >= 0)
    s = (unsigned int)freed;
  else if (q->sumtable_used < q->sumtable_cap)
    s = q->sumtable_used++;
  else
  {
    s = 0;
    for (i = 1; i < q->sumtable_used; ++i)
      if (q->sumtable_stamp[i] < q->sumtable_stamp[s]) s = i;
    if (q->sumtable_key[s]) evicted_add(q, q->sumtable_key[s]);
  }
  q->sumtable_key[s] = key;
  q->sumtable_stamp[s] = ++q->sumtable_clock;
  return (int)s;
}

/* the client is done with this host buffer (e.g. about to free it): forget the key, so a
   new buffer at the same address is not mistaken for it; the device buffer is kept for the
   next table */
int pll_amd_forget_sumtable(pll_partition_t * p, const double * sumtable)
{
  pll_amd_partition_t * q = pll_amd_priv(p);
  int slot = slot_of(q, sumtable);
  evicted_remove(q, sumtable);
  if (slot < 0) return PLL_FAILURE;
  q->sumtable_key[slot] = NULL;
  q->sumtable_stamp[slot] = 0; /* first to be reused */
  return PLL_SUCCESS;
}

int pll_update_sumtable(pll_partition_t * p, unsigned int parent_clv_index,
                        unsigned int child_clv_index, int parent_scaler_index,
                        int child_scaler_index, const unsigned int * params_indices,
                        double * sumtable)
{
  pll_amd_partition_t * q = pll_amd_priv(p);
  int slot, rc;
  unsigned int n;
  if (!sumtable)
  {
    pll_amd_set_error(PLL_ERROR_PARAM_INVALID, "sumtable is NULL");
    return PLL_FAILURE;
  }
  for (n = 0; n < p->rate_cats; ++n)
    if (!p->eigen_decomp_valid[params_indices[n]])
      if (!pll_update_eigen(p, params_indices[n])) return PLL_FAILURE;
  if (!pll_amd_flush_model(p)) return PLL_FAILURE;
  if (!pll_amd_repeats_scaler_ok(p, parent_clv_index, parent_scaler_index) ||
      !pll_amd_repeats_scaler_ok(p, child_clv_index, child_scaler_index))
    return PLL_FAILURE;
  slot = slot_of(q, sumtable);
  if (slot < 0) slot = slot_assign(q, sumtable);
  rc = pllhip_update_sumtable(q->ctx, parent_clv_index, parent_scaler_index, child_clv_index,
                              child_scaler_index, params_indices, (unsigned int)slot);
  if (rc)
  {
    q->sumtable_key[slot] = NULL;
    return pll_amd_fail_hip(rc, "sumtable update");
  }
  if (PLL_AMD_MIRRORS(p)) return pll_amd_sync_sumtable(p, sumtable);
  /* The table lives on the device; the caller's buffer is its key and is NOT filled.  A reference client that reads
     it anyway (derivatives.c hands it to pll_core_likelihood_derivatives; a client may sum it itself) must not find
     plausible numbers there: the first site's entries become signalling NaNs (VERDICT r3 Weak 8). */
  {
    const unsigned long long poison = PLL_AMD_SUMTABLE_POISON;
    const unsigned int span = p->states * p->rate_cats;
    unsigned int i;
    for (i = 0; i < span; ++i) memcpy(&sumtable[i], &poison, sizeof(poison));
  }
  return PLL_SUCCESS;
}

int pll_amd_sync_sumtable(pll_partition_t * p, double * sumtable)
{
  pll_amd_partition_t * q = pll_amd_priv(p);
  int slot = slot_of(q, sumtable), rc;
  if (slot < 0)
  {
    if (was_evicted(q, sumtable))
      pll_amd_set_error(PLL_ERROR_HIP_SUMTABLE_EVICTED,
                        "the device copy of this sumtable was recycled (more than %u live sumtables; "
                        "PLL_AMD_SUMTABLE_SLOTS raises the limit)", q->sumtable_cap);
    else
      pll_amd_set_error(PLL_ERROR_PARAM_INVALID, "no device sumtable is associated with this buffer");
    return PLL_FAILURE;
  }
  if ((rc = pllhip_get_sumtable(q->ctx, (unsigned int)slot, sumtable)))
    return pll_amd_fail_hip(rc, "sumtable download");
  return PLL_SUCCESS;
}

int pll_compute_likelihood_derivatives(pll_partition_t * p, int parent_scaler_index,
                                       int child_scaler_index, double branch_length,
                                       const unsigned int * params_indices,
                                       const double * sumtable, double * d_f, double * dd_f)
{
  pll_amd_partition_t * q = pll_amd_priv(p);
  const unsigned int S = p->states, R = p->rate_cats;
  unsigned int i, j;
  int slot, rc;
  double * diag;
  /* (the scaler indices only matter to the asc-bias terms, core_derivatives.c:683-686) */
  if (!pll_amd_flush_model(p)) return PLL_FAILURE;
  slot = slot_of(q, sumtable);
  if (slot < 0)
  {
    if (was_evicted(q, sumtable))
    {
      /* produced on the device, then recycled: the host buffer was never written */
      pll_amd_set_error(PLL_ERROR_HIP_SUMTABLE_EVICTED,
                        "the device copy of this sumtable was recycled (more than %u live sumtables; call "
                        "pll_update_sumtable again, or raise PLL_AMD_SUMTABLE_SLOTS)", q->sumtable_cap);
      return PLL_FAILURE;
    }
    /* a table the caller filled itself: take the host contents */
    slot = slot_assign(q, sumtable);
    if ((rc = pllhip_put_sumtable(q->ctx, (unsigned int)slot, sumtable)))
    {
      q->sumtable_key[slot] = NULL;
      return pll_amd_fail_hip(rc, "sumtable upload");
    }
  }

  /* e^{lambda r t}, its first and second t-derivative per (category, state):
     core_derivatives.c:560-575, same expression order, libm exp */
  diag = (double *)malloc((size_t)R * S * 4 * sizeof(double));
  if (!diag)
  {
    pll_amd_set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate memory for diagptable");
    return PLL_FAILURE;
  }
  for (i = 0; i < R; ++i)
  {
    const double * ev = p->eigenvals[params_indices[i]];
    const double ki = p->rates[i] / (1.0 - p->prop_invar[params_indices[i]]);
    double * dp = diag + (size_t)i * S * 4;
    for (j = 0; j < S; ++j, dp += 4)
    {
      dp[0] = exp(ev[j] * ki * branch_length);
      dp[1] = ev[j] * ki * dp[0];
      dp[2] = ev[j] * ki * ev[j] * ki * dp[0];
      dp[3] = 0;
    }
  }
  rc = pllhip_likelihood_derivatives(q->ctx, (unsigned int)slot, parent_scaler_index,
                                     child_scaler_index, params_indices, diag, d_f, dd_f);
  free(diag);
  if (rc) return pll_amd_fail_hip(rc, "likelihood derivatives");
  return PLL_SUCCESS;
}
