/* compress.c -- site-pattern compression: identical alignment
 * columns are merged and counted, which is what feeds pll_set_pattern_weights
 * and shrinks `sites` -- and with it every kernel's work -- by 2-10x on real data.
 *
 * Replaces pll_compress_site_patterns (compress.c:138).  Contract kept: the
 * sequences are rewritten IN PLACE with the unique columns in ascending
 * lexicographic order of their state codes (compared as C `char`s), terminated
 * by NUL; characters that map to the same state come back as the LAST such
 * character of the map ('a' for A/a); the returned weight array has the new
 * length, *length is updated.
 *
 * The reference sorts column strings with a randomised multikey quicksort
 * (compress.c:33); any correct sort yields the same output because the order is
 * total and equal columns are merged.  Here: an index sort by 3-way radix
 * quicksort (Bentley & Sedgewick 1997) on column-major bytes, no recursion on the
 * equal partition beyond `count` levels, explicit stack.
 */
#include <stdio.h>

#include "internal.h"

typedef struct
{
  size_t lo, hi; /* [lo, hi) of the index array */
  int depth;
} span_t;

static inline int key(const signed char * cols, size_t stride, unsigned int idx, int depth)
{
  return cols[(size_t)idx * stride + depth];
}

static int sort_columns(unsigned int * idx, size_t n, const signed char * cols, size_t stride,
                        int count)
{
  size_t cap = 64, top = 0;
  span_t * st = (span_t *)malloc(cap * sizeof(span_t));
  if (!st) return 0;
  st[top++] = (span_t){0, n, 0};
  while (top)
  {
    span_t s = st[--top];
    size_t lt, gt, i;
    int v;
    unsigned int t;
    if (s.hi - s.lo <= 1 || s.depth >= count) continue;
    /* median-of-three pivot on the current byte */
    {
      const int a = key(cols, stride, idx[s.lo], s.depth);
      const int b = key(cols, stride, idx[s.lo + (s.hi - s.lo) / 2], s.depth);
      const int c = key(cols, stride, idx[s.hi - 1], s.depth);
      v = a < b ? (b < c ? b : (a < c ? c : a)) : (a < c ? a : (b < c ? c : b));
    }
    lt = s.lo;
    gt = s.hi;
    i = s.lo;
    while (i < gt)
    {
      const int k = key(cols, stride, idx[i], s.depth);
      if (k < v) { t = idx[lt]; idx[lt] = idx[i]; idx[i] = t; ++lt; ++i; }
      else if (k > v) { --gt; t = idx[gt]; idx[gt] = idx[i]; idx[i] = t; }
      else ++i;
    }
    if (top + 3 > cap)
    {
      span_t * g;
      cap *= 2;
      g = (span_t *)realloc(st, cap * sizeof(span_t));
      if (!g) { free(st); return 0; }
      st = g;
    }
    st[top++] = (span_t){s.lo, lt, s.depth};
    st[top++] = (span_t){lt, gt, s.depth + 1};
    st[top++] = (span_t){gt, s.hi, s.depth};
  }
  free(st);
  return 1;
}

unsigned int * pll_compress_site_patterns(char ** sequence, const unsigned int * map, int count,
                                          int * length)
{
  unsigned char enc[PLL_ASCII_SIZE], dec[PLL_ASCII_SIZE];
  unsigned int maxval = 0, * idx = NULL, * weight = NULL;
  signed char * cols = NULL;
  const size_t n = (size_t)*length;
  size_t i, unique = 0;
  int j;

  if (!count || !map || map[0]) return NULL;

  /* state codes that fit a byte are used as they are; otherwise distinct states
     are renumbered 1, 2, ... in ASCII order (compress.c:83-107,156-166) */
  for (i = 0; i < PLL_ASCII_SIZE; ++i)
    if (map[i] > maxval) maxval = map[i];
  memset(enc, 0, sizeof(enc));
  if (maxval >= PLL_ASCII_SIZE)
  {
    unsigned char next = 1;
    for (i = 0; i < PLL_ASCII_SIZE; ++i)
    {
      size_t k;
      if (!map[i] || enc[i]) continue;
      for (k = i; k < PLL_ASCII_SIZE; ++k)
        if (map[k] == map[i]) enc[k] = next;
      ++next;
    }
  }
  else
    for (i = 0; i < PLL_ASCII_SIZE; ++i) enc[i] = (unsigned char)map[i];
  memset(dec, 0, sizeof(dec));
  for (i = 0; i < PLL_ASCII_SIZE; ++i)
    if (map[i]) dec[enc[i]] = (unsigned char)i;

  cols = (signed char *)malloc(n * (size_t)count);
  idx = (unsigned int *)malloc(n * sizeof(unsigned int));
  weight = (unsigned int *)malloc((n ? n : 1) * sizeof(unsigned int));
  if (!cols || !idx || !weight)
  {
    free(cols);
    free(idx);
    free(weight);
    pll_amd_set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate space for matrix columns.");
    return NULL;
  }
  /* column-major copy of the encoded alignment */
  for (j = 0; j < count; ++j)
    for (i = 0; i < n; ++i)
      cols[i * (size_t)count + j] = (signed char)enc[(unsigned char)sequence[j][i]];
  for (i = 0; i < n; ++i) idx[i] = (unsigned int)i;

  if (!sort_columns(idx, n, cols, (size_t)count, count))
  {
    free(cols);
    free(idx);
    free(weight);
    pll_amd_set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate space for sorting.");
    return NULL;
  }

  /* merge equal neighbours, write the unique columns back decoded */
  for (i = 0; i < n; ++i)
  {
    const signed char * c = cols + (size_t)idx[i] * count;
    if (i && !memcmp(c, cols + (size_t)idx[i - 1] * count, (size_t)count))
    {
      weight[unique - 1]++;
      continue;
    }
    for (j = 0; j < count; ++j) sequence[j][unique] = (char)dec[(unsigned char)c[j]];
    weight[unique++] = 1;
  }
  for (j = 0; j < count; ++j) sequence[j][unique] = 0;
  free(cols);
  free(idx);
  {
    unsigned int * fit = (unsigned int *)realloc(weight, (unique ? unique : 1) * sizeof(unsigned int));
    if (fit) weight = fit;
  }
  *length = (int)unique;
  return weight;
}
