/* repeats.c -- site repeats: alignment sites that cannot be told apart below a node
 * share one CLV entry there.
 *
 * NOT part of the reference snapshot this library replaces (libpll 0.3.2 has no site
 * repeats; BASELINE.json's config 5 names them, later libpll versions implement them
 * as PLL_ATTRIB_SITE_REPEATS).  Own design, opt-in through that attribute bit, 4-state
 * data with PLL_ATTRIB_PATTERN_TIP only.  The contract is that every result equals the
 * one obtained without the attribute: tests/test_gpu_repeats.py checks CLVs (expanded),
 * scale buffers, per-site lnL and derivatives bit for bit against the plain path.
 *
 * Two sites belong to the same class at a node iff their classes at both children
 * agree (at a tip: iff the tip shows the same character).  The classes of a parent are
 * therefore the distinct pairs (class at child 1, class at child 2).  A node is stored
 * by class when it has at most half as many classes as sites; otherwise it is stored
 * per site as usual, and so is every node above it.
 *
 * This file is the bookkeeping: classes depend on the topology and on the tip sequences
 * only, so each CLV slot remembers from which (child slot, generation of that child's
 * classes) pair its classes were built and asks the device to identify them again
 * (hip/repeats.hip: sort of the pairs, O(sites) per op) only when that signature
 * changes -- branch lengths and model parameters never do.  It also expands CLVs and
 * scale buffers stored by class when a host mirror is asked for.
 * pll_amd_identify_repeats() below is the same identification in plain C; the tests use
 * it to check the device's classes.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "internal.h"

/* Distinct pairs (ida[n], idb[n]) numbered by first appearance.
 *   ida/idb   class (or character) of each site at the two children, values < na / < nb
 *   max       give up beyond this many classes
 * On success returns the class count and fills site_id[sites], lrow[count],
 * rrow[count]; returns 0 when there are more than `max` classes (outputs undefined). */
unsigned int pll_amd_identify_repeats(const unsigned int * ida, unsigned int na,
                                      const unsigned int * idb, unsigned int nb,
                                      unsigned int sites, unsigned int max,
                                      unsigned int * site_id, unsigned int * lrow,
                                      unsigned int * rrow)
{
  unsigned int count = 0, n;
  const uint64_t cells = (uint64_t)na * nb;
  if (!sites || !na || !nb) return 0;
  if (max > sites) max = sites;

  if (cells <= ((uint64_t)1 << 22))
  {
    /* direct table over all pairs */
    int32_t * table = (int32_t *)malloc((size_t)cells * sizeof(int32_t));
    if (!table) return 0;
    memset(table, 0xff, (size_t)cells * sizeof(int32_t));
    for (n = 0; n < sites; ++n)
    {
      const size_t key = (size_t)ida[n] * nb + idb[n];
      int32_t c = table[key];
      if (c < 0)
      {
        if (count == max)
        {
          free(table);
          return 0;
        }
        c = (int32_t)count;
        table[key] = c;
        lrow[count] = ida[n];
        rrow[count] = idb[n];
        ++count;
      }
      site_id[n] = (unsigned int)c;
    }
    free(table);
    return count;
  }

  /* open addressing on the 64-bit pair; at most `max` <= sites entries */
  {
    size_t cap = 1;
    while (cap < 2 * (size_t)max + 2) cap <<= 1;
    uint64_t * keys = (uint64_t *)malloc(cap * sizeof(uint64_t));
    unsigned int * vals = (unsigned int *)malloc(cap * sizeof(unsigned int));
    if (!keys || !vals)
    {
      free(keys);
      free(vals);
      return 0;
    }
    memset(keys, 0xff, cap * sizeof(uint64_t)); /* all-ones = empty (no pair has both halves ~0) */
    for (n = 0; n < sites; ++n)
    {
      const uint64_t key = ((uint64_t)ida[n] << 32) | idb[n];
      size_t h = (size_t)((key * 0x9e3779b97f4a7c15ull) >> 20) & (cap - 1);
      while (keys[h] != key && keys[h] != ~(uint64_t)0) h = (h + 1) & (cap - 1);
      if (keys[h] != key)
      {
        if (count == max)
        {
          free(keys);
          free(vals);
          return 0;
        }
        keys[h] = key;
        vals[h] = count;
        lrow[count] = ida[n];
        rrow[count] = idb[n];
        ++count;
      }
      site_id[n] = vals[h];
    }
    free(keys);
    free(vals);
    return count;
  }
}

void pll_amd_repeats_free(pll_amd_partition_t * q)
{
  unsigned int i;
  if (!q->rep) return;
  for (i = 0; i < q->pub.tips + q->pub.clv_buffers; ++i) free(q->rep[i].site_id);
  free(q->rep);
  free(q->scaler_owner);
  q->rep = NULL;
  q->scaler_owner = NULL;
}

int pll_amd_repeats_alloc(pll_amd_partition_t * q)
{
  const unsigned int nodes = q->pub.tips + q->pub.clv_buffers;
  unsigned int i;
  q->rep = (pll_amd_node_repeats_t *)calloc(nodes, sizeof(pll_amd_node_repeats_t));
  q->scaler_owner = (int *)malloc((q->pub.scale_buffers ? q->pub.scale_buffers : 1) * sizeof(int));
  if (!q->rep || !q->scaler_owner) return 0;
  for (i = 0; i < q->pub.scale_buffers; ++i) q->scaler_owner[i] = -1;
  for (i = 0; i < q->pub.tips; ++i) q->rep[i].gen = 1;
  return 1;
}

/* a tip's characters changed: every class built on it is stale */
void pll_amd_repeats_tip_changed(pll_amd_partition_t * q, unsigned int tip)
{
  if (q->rep) q->rep[tip].gen++;
}

/* pll_update_partials under site repeats: bring the classes of every parent up to date
 * (list order: children first) and run the ops.  The row maps belong to CLV SLOTS, and a
 * list may write a slot more than once (re-rooting an unrooted tree does): ops already
 * accepted that read or write a slot must run with its old maps, so the list is handed
 * to the device in pieces -- a piece ends where a slot it touches gets new classes.
 * Returns PLL_SUCCESS / PLL_FAILURE. */
int pll_amd_repeats_update(pll_partition_t * p, const pll_operation_t * ops, unsigned int count)
{
  pll_amd_partition_t * q = pll_amd_priv(p);
  const unsigned int tips = p->tips, nodes = p->tips + p->clv_buffers;
  unsigned int i, start = 0;
  int rc;

  q->rep_epoch++;
  for (i = 0; i < count; ++i)
  {
    const pll_operation_t * op = &ops[i];
    const unsigned int c1 = op->child1_clv_index, c2 = op->child2_clv_index;
    pll_amd_node_repeats_t * par;
    unsigned int sig[4], classes = 0;
    if (op->parent_clv_index >= nodes || op->parent_clv_index < tips || c1 >= nodes || c2 >= nodes)
      continue; /* the device call reports the bad index */
    par = &q->rep[op->parent_clv_index];
    /* A scale buffer is stored the way the CLV it was written with is: a child's counts
       can only be taken from the buffer that belongs to that child. */
    {
      const int cs[2] = {op->child1_scaler_index, op->child2_scaler_index};
      const unsigned int cc[2] = {c1, c2};
      int s;
      for (s = 0; s < 2; ++s)
        if (cs[s] >= 0 && (unsigned int)cs[s] < p->scale_buffers && q->scaler_owner[cs[s]] != (int)cc[s])
        {
          pll_amd_set_error(PLL_ERROR_PARAM_INVALID,
                            "Site repeats: scale buffer %d was not written together with CLV %u.", cs[s], cc[s]);
          return PLL_FAILURE;
        }
    }
    if (op->parent_scaler_index >= 0 && (unsigned int)op->parent_scaler_index < p->scale_buffers)
      q->scaler_owner[op->parent_scaler_index] = (int)op->parent_clv_index;
    sig[0] = c1;
    sig[1] = q->rep[c1].gen;
    sig[2] = c2;
    sig[3] = q->rep[c2].gen;
    if (!par->sig_valid || memcmp(sig, par->sig, sizeof(sig)))
    {
      if (par->touched == q->rep_epoch && i > start)
      {
        /* earlier ops of this piece use the slot as it is now: run them first */
        if ((rc = pllhip_update_partials(q->ctx, (const pllhip_op_t *)(ops + start), i - start)))
          return pll_amd_fail_hip(rc, "CLV update");
        start = i;
        q->rep_epoch++;
      }
      if ((rc = pllhip_identify_repeats(q->ctx, op->parent_clv_index, c1, c2, p->sites / 2, &classes)))
        return pll_amd_fail_hip(rc, "site-repeat identification");
      par->classes = classes;
      par->site_id_valid = 0;
      par->gen++;
      memcpy(par->sig, sig, sizeof(sig));
      par->sig_valid = 1;
    }
    par->touched = q->rep_epoch;
    q->rep[c1].touched = q->rep_epoch;
    q->rep[c2].touched = q->rep_epoch;
  }
  if (count > start && (rc = pllhip_update_partials(q->ctx, (const pllhip_op_t *)(ops + start), count - start)))
    return pll_amd_fail_hip(rc, "CLV update");
  return PLL_SUCCESS;
}

const unsigned int * pll_amd_repeats_site_id(pll_partition_t * p, unsigned int clv_index)
{
  pll_amd_partition_t * q = pll_amd_priv(p);
  pll_amd_node_repeats_t * r;
  int rc;
  if (!q->rep || clv_index >= p->tips + p->clv_buffers || !q->rep[clv_index].classes) return NULL;
  r = &q->rep[clv_index];
  if (r->site_id_valid) return r->site_id;
  if (!r->site_id) r->site_id = (unsigned int *)malloc((size_t)p->sites * sizeof(unsigned int));
  if (!r->site_id)
  {
    pll_amd_set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate site-repeat map.");
    return NULL;
  }
  if ((rc = pllhip_get_site_id(q->ctx, clv_index, r->site_id)))
  {
    pll_amd_fail_hip(rc, "download of site-repeat map");
    return NULL;
  }
  r->site_id_valid = 1;
  return r->site_id;
}

/* Expand `rows` stored by class into one row per site, in place: buf holds `classes`
 * rows of `per` elements of `elem` bytes at its start and has room for `sites` rows. */
int pll_amd_repeats_expand(void * buf, const unsigned int * site_id, unsigned int classes,
                           unsigned int sites, size_t row_bytes)
{
  char * rows = (char *)malloc((size_t)classes * row_bytes);
  size_t s;
  if (!rows)
  {
    pll_amd_set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate site-repeat expansion buffer.");
    return PLL_FAILURE;
  }
  memcpy(rows, buf, (size_t)classes * row_bytes);
  for (s = 0; s < sites; ++s)
    memcpy((char *)buf + s * row_bytes, rows + (size_t)site_id[s] * row_bytes, row_bytes);
  free(rows);
  return PLL_SUCCESS;
}

/* lnL / sumtable / derivative calls: is `scaler` the buffer written with `clv`? */
int pll_amd_repeats_scaler_ok(pll_partition_t * p, unsigned int clv, int scaler)
{
  pll_amd_partition_t * q = pll_amd_priv(p);
  if (!q->rep || scaler < 0 || (unsigned int)scaler >= p->scale_buffers) return 1;
  if (q->scaler_owner[scaler] == (int)clv) return 1;
  pll_amd_set_error(PLL_ERROR_PARAM_INVALID,
                    "Site repeats: scale buffer %d was not written together with CLV %u.", scaler, clv);
  return 0;
}

unsigned int pll_amd_repeats_classes(const pll_partition_t * p, unsigned int clv_index)
{
  const pll_amd_partition_t * q = pll_amd_priv(p);
  if (!q->rep || clv_index >= p->tips + p->clv_buffers) return 0;
  /* (a partition over several devices identifies per shard: the sum of the shards' rows) */
  if (pllhip_shard_count(q->ctx) > 1) return pllhip_repeats_rows(q->ctx, clv_index);
  return q->rep[clv_index].classes;
}
