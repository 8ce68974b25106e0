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
 * therefore the distinct pairs (class at child 1, class at child 2), numbered in order
 * of first appearance.  Per inner node the host keeps
 *      site_id[site]  -> class         (NULL when the node is not compressed)
 * and hands the device, per class, the row of each child it is computed from
 * (lrow / rrow: a class index of that child, a site index if the child is not
 * compressed, a tip character if it is a tip).  A node is compressed when it has at
 * most half as many classes as sites; otherwise it is stored per site as usual, and so
 * is every node above it.
 *
 * Identification is host work, O(sites) per op, done when an op is seen for the first
 * time with a given pair of children and reused afterwards (branch lengths and model
 * parameters do not change classes): a signature per CLV slot of (child slot,
 * generation of the child's classes) x 2 decides.
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

/* Bring the classes of every parent in `ops` up to date (list order: children first)
 * and tell the device about the ones that changed.  Returns PLL_SUCCESS / PLL_FAILURE. */
int pll_amd_repeats_update(pll_partition_t * p, const pll_operation_t * ops, unsigned int count)
{
  pll_amd_partition_t * q = pll_amd_priv(p);
  const unsigned int sites = p->sites, tips = p->tips, nodes = p->tips + p->clv_buffers;
  unsigned int i, n;
  unsigned int * ida = NULL, * idb = NULL, * lrow = NULL, * rrow = NULL, * sid = NULL;
  int ok = PLL_SUCCESS;

  for (i = 0; i < count && ok; ++i)
  {
    const pll_operation_t * op = &ops[i];
    const unsigned int c[2] = {op->child1_clv_index, op->child2_clv_index};
    pll_amd_node_repeats_t * par;
    unsigned int sig[4], na[2], classes = 0;
    int compress = 1, s;
    if (op->parent_clv_index >= nodes || op->parent_clv_index < tips || c[0] >= nodes || c[1] >= nodes)
      continue; /* the device call reports the bad index */
    par = &q->rep[op->parent_clv_index];
    if (op->parent_scaler_index >= 0 && (unsigned int)op->parent_scaler_index < p->scale_buffers)
      q->scaler_owner[op->parent_scaler_index] = (int)op->parent_clv_index;
    sig[0] = c[0];
    sig[1] = q->rep[c[0]].gen;
    sig[2] = c[1];
    sig[3] = q->rep[c[1]].gen;
    if (par->sig_valid && !memcmp(sig, par->sig, sizeof(sig))) continue; /* classes still right */

    /* children: a tip shows characters (16 codes), a compressed inner node its classes;
       an inner node stored per site cannot be the basis of a compression */
    for (s = 0; s < 2; ++s)
    {
      if (c[s] < tips) na[s] = 16;
      else if (q->rep[c[s]].site_id) na[s] = q->rep[c[s]].classes;
      else compress = 0;
    }
    if (compress)
    {
      if (!ida)
      {
        ida = (unsigned int *)malloc((size_t)sites * sizeof(unsigned int));
        idb = (unsigned int *)malloc((size_t)sites * sizeof(unsigned int));
        lrow = (unsigned int *)malloc((size_t)sites * sizeof(unsigned int));
        rrow = (unsigned int *)malloc((size_t)sites * sizeof(unsigned int));
        sid = (unsigned int *)malloc((size_t)sites * sizeof(unsigned int));
        if (!ida || !idb || !lrow || !rrow || !sid)
        {
          pll_amd_set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate site-repeat work arrays.");
          ok = PLL_FAILURE;
          break;
        }
      }
      for (s = 0; s < 2; ++s)
      {
        unsigned int * dst = s ? idb : ida;
        if (c[s] < tips)
        {
          const unsigned char * codes = p->tipchars[c[s]];
          for (n = 0; n < sites; ++n) dst[n] = codes[n] & 15u;
        }
        else
          memcpy(dst, q->rep[c[s]].site_id, (size_t)sites * sizeof(unsigned int));
      }
      classes = pll_amd_identify_repeats(ida, na[0], idb, na[1], sites, sites / 2, sid, lrow, rrow);
    }

    par->gen++;
    memcpy(par->sig, sig, sizeof(sig));
    par->sig_valid = 1;
    if (classes)
    {
      int rc;
      if (!par->site_id) par->site_id = (unsigned int *)malloc((size_t)sites * sizeof(unsigned int));
      if (!par->site_id)
      {
        pll_amd_set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate site-repeat classes.");
        ok = PLL_FAILURE;
        break;
      }
      memcpy(par->site_id, sid, (size_t)sites * sizeof(unsigned int));
      par->classes = classes;
      if ((rc = pllhip_put_repeats(q->ctx, op->parent_clv_index, classes, sid, lrow, rrow)))
        ok = pll_amd_fail_hip(rc, "upload of site-repeat classes");
    }
    else
    {
      int rc;
      free(par->site_id);
      par->site_id = NULL;
      par->classes = 0;
      if ((rc = pllhip_put_repeats(q->ctx, op->parent_clv_index, 0, NULL, NULL, NULL)))
        ok = pll_amd_fail_hip(rc, "reset of site-repeat classes");
    }
  }
  free(ida);
  free(idb);
  free(lrow);
  free(rrow);
  free(sid);
  return ok;
}

unsigned int pll_amd_repeats_classes(const pll_partition_t * p, unsigned int clv_index)
{
  const pll_amd_partition_t * q = pll_amd_priv(p);
  if (!q->rep || clv_index >= p->tips + p->clv_buffers || !q->rep[clv_index].site_id) return 0;
  return q->rep[clv_index].classes;
}
