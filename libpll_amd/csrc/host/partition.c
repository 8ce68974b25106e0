/* partition.c -- the partition container: creation, destruction, tip data,
 * pattern weights, host mirrors.
 *
 * Replaces the container half of the reference's pll.c (pll_partition_create
 * pll.c:399, pll_partition_destroy :820, pll_set_tip_states :966,
 * pll_set_tip_clv :1001, pll_set_pattern_weights :1047, create_charmap :272,
 * update_charmap :136).  Small model arrays live on the host exactly as in the
 * reference (clients read them); CLVs, scalers, tip codes and P-matrices live
 * in HBM behind partition->ctx, with host mirrors filled on request.
 */
#include <stdarg.h>
#include <stdio.h>

#include "internal.h"

__thread int pll_errno;
__thread char pll_errmsg[200] = {0};

int pll_amd_mirror_mode = 0;
/* Device selection: the calling thread's own value if that thread has set one (distinct threads may create
 * partitions on distinct devices concurrently), else the process-wide default -- whatever
 * pll_amd_set_device / pll_amd_set_devices was last given by ANY thread: a client that selects its device once on
 * the main thread and creates partitions from worker threads gets that device there (round 3 made the setting
 * thread-local only, and such workers silently fell back to the environment) --, else the environment. */
static __thread int g_device = -1;
static int g_device_default = -1; /* (read and written with __atomic builtins) */

void pll_amd_set_error(int code, const char * fmt, ...)
{
  va_list ap;
  pll_errno = code;
  va_start(ap, fmt);
  vsnprintf(pll_errmsg, sizeof(pll_errmsg), fmt, ap);
  va_end(ap);
}

int pll_amd_fail_hip(int rc, const char * what)
{
  pll_amd_set_error(PLL_ERROR_HIP_RUNTIME, "%s: %s (rc=%d)", what, pllhip_last_error(), rc);
  return PLL_FAILURE;
}

void * pll_aligned_alloc(size_t size, size_t alignment)
{
  void * mem = NULL;
  if (alignment < sizeof(void *)) alignment = sizeof(void *);
  if (posix_memalign(&mem, alignment, size ? size : alignment)) return NULL;
  return mem;
}

void pll_aligned_free(void * ptr) { free(ptr); }

int pll_amd_device_count(void)
{
  int n = 0;
  if (pllhip_device_count(&n)) return 0;
  return n;
}

/* The process-wide defaults (device, device list) belong to the thread that selected a device FIRST -- the client that
 * picks its device once on the main thread and creates partitions from workers -- : only that thread's later calls move
 * them.  A worker that selects a device of its own changes its own choice and nobody else's (as
 * last-writer-wins, partitions created concurrently by threads that had set nothing landed on whichever device a
 * sibling had just chosen). */
/* (round 6, ADVICE r5: the owner used to be named by the address of a thread-local variable -- an address the next
 * thread may be given once the owner has exited, which then silently became the owner, while nobody else could ever
 * change the defaults again.  Now: the kernel's thread id, unique among the live threads of the process, and an explicit
 * way to give the ownership up -- pll_amd_set_device(-1) from the owner; the next thread that selects becomes the owner.
 * An owner that exits without doing so keeps the defaults it set; they can then only be overridden per thread.) */
#include <sys/syscall.h>
#include <unistd.h>
static long g_default_owner = 0; /* 0: nobody yet */
static long this_thread(void)
{
  static __thread long tid = 0;
  if (!tid) tid = (long)syscall(SYS_gettid);
  return tid;
}
static int owns_defaults(void)
{
  long none = 0;
  const long me = this_thread();
  if (__atomic_compare_exchange_n(&g_default_owner, &none, me, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) return 1;
  return none == me;
}

int pll_amd_set_device(int device)
{
  if (device < 0)
  {
    /* back to "not set by this thread"; the owner of the process-wide defaults also gives the ownership up (the
     * defaults themselves stay as they are until the next owner selects) */
    long me = this_thread();
    g_device = -1;
    (void)__atomic_compare_exchange_n(&g_default_owner, &me, 0, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE);
    return PLL_SUCCESS;
  }
  g_device = device;
  if (owns_defaults()) __atomic_store_n(&g_device_default, device, __ATOMIC_RELAXED);
  return PLL_SUCCESS;
}

void pll_amd_set_mirror_mode(int on) { pll_amd_mirror_mode = on ? 1 : 0; }

unsigned int pll_amd_shard_count(const pll_partition_t * p) { return pllhip_shard_count(pll_amd_priv(p)->ctx); }

/* Devices the NEXT pll_partition_create shards its sites over (pll_amd_set_devices, else env
 * PLL_AMD_DEVICES = "0-7" / "0,1,2" / "all"; an ordinal may repeat).  One entry or none: the
 * partition lives on one device as before. */
#define PLL_AMD_MAX_DEVICES 64
static __thread int g_devices[PLL_AMD_MAX_DEVICES];
static __thread int g_ndevices = -1; /* -1: not set by this thread: the process-wide list, then the environment */
static int g_devices_default[PLL_AMD_MAX_DEVICES];
static int g_ndevices_default = -1;
static volatile int g_devices_lock = 0; /* (a spin lock: the list is 64 ints, copied under it) */

static void devices_lock(void) { while (__atomic_exchange_n(&g_devices_lock, 1, __ATOMIC_ACQUIRE)) { } }
static void devices_unlock(void) { __atomic_store_n(&g_devices_lock, 0, __ATOMIC_RELEASE); }

int pll_amd_set_devices(const int * devices, unsigned int count)
{
  unsigned int i;
  if (count > PLL_AMD_MAX_DEVICES || (count && !devices))
  {
    pll_amd_set_error(PLL_ERROR_PARAM_INVALID, "pll_amd_set_devices: at most %d devices", PLL_AMD_MAX_DEVICES);
    return PLL_FAILURE;
  }
  for (i = 0; i < count; ++i) g_devices[i] = devices[i];
  g_ndevices = count ? (int)count : -1; /* an empty list: back to the process default / the environment's */
  if (owns_defaults())
  {
    devices_lock();
    for (i = 0; i < count; ++i) g_devices_default[i] = devices[i];
    g_ndevices_default = count ? (int)count : -1;
    devices_unlock();
  }
  return PLL_SUCCESS;
}

/* fills list[], returns how many (0: no sharding requested), -1 on a malformed PLL_AMD_DEVICES */
static int device_list(int * list)
{
  const char * e;
  int n = 0;
  if (g_ndevices >= 0)
  {
    memcpy(list, g_devices, (size_t)g_ndevices * sizeof(int));
    return g_ndevices;
  }
  devices_lock();
  n = g_ndevices_default;
  if (n >= 0) memcpy(list, g_devices_default, (size_t)n * sizeof(int));
  devices_unlock();
  if (n >= 0) return n;
  n = 0;
  e = getenv("PLL_AMD_DEVICES");
  if (!e || !*e) return 0;
  if (!strcmp(e, "all"))
  {
    const int have = pll_amd_device_count();
    for (n = 0; n < have && n < PLL_AMD_MAX_DEVICES; ++n) list[n] = n;
    return n;
  }
  while (*e)
  {
    char * end;
    long a = strtol(e, &end, 10), b;
    if (end == e || a < 0) return -1;
    b = a;
    if (*end == '-')
    {
      e = end + 1;
      b = strtol(e, &end, 10);
      if (end == e || b < a) return -1;
    }
    for (; a <= b; ++a)
    {
      if (n >= PLL_AMD_MAX_DEVICES) return -1;
      list[n++] = (int)a;
    }
    if (*end == ',') ++end;
    else if (*end) return -1;
    e = end;
  }
  return n;
}

static int default_device(void)
{
  if (g_device >= 0) return g_device;
  {
    const int d = __atomic_load_n(&g_device_default, __ATOMIC_RELAXED);
    if (d >= 0) return d;
  }
  const char * e = getenv("PLL_AMD_DEVICE");
  if (!e || !*e) e = getenv("LOCAL_RANK");
  if (e && *e)
  {
    int d = atoi(e);
    int n = pll_amd_device_count();
    if (n > 0) return d % n;
  }
  return 0;
}

int pll_amd_core_device(void) { return default_device(); }
int pll_amd_get_device(void) { return default_device(); }

static void free_ptr_array(void ** a, unsigned int n)
{
  unsigned int i;
  if (!a) return;
  for (i = 0; i < n; ++i) free(a[i]);
  free(a);
}

void pll_partition_destroy(pll_partition_t * p)
{
  pll_amd_partition_t * q = pll_amd_priv(p);
  unsigned int nodes;
  if (!p) return;
  nodes = p->tips + p->clv_buffers;
  if (q->ctx) pllhip_ctx_destroy(q->ctx);
  free((void *)q->sumtable_evicted);
  {
    /* mirrors that are the device layer's pinned memory go back there */
    unsigned int i;
    for (i = 0; q->clv_pinned && p->clv && i < nodes; ++i)
      if (q->clv_pinned[i]) { pllhip_host_free(p->clv[i]); p->clv[i] = NULL; }
    for (i = 0; q->scaler_pinned && p->scale_buffer && i < p->scale_buffers; ++i)
      if (q->scaler_pinned[i]) { pllhip_host_free(p->scale_buffer[i]); p->scale_buffer[i] = NULL; }
    free(q->clv_pinned);
    free(q->scaler_pinned);
  }
  free_ptr_array((void **)p->clv, nodes);
  if (p->pmatrix)
  {
    free(p->pmatrix[0]);
    free(p->pmatrix);
  }
  free(p->rates);
  free(p->rate_weights);
  free_ptr_array((void **)p->subst_params, p->rate_matrices);
  free_ptr_array((void **)p->scale_buffer, p->scale_buffers);
  free_ptr_array((void **)p->frequencies, p->rate_matrices);
  free(p->prop_invar);
  free(p->invariant);
  free(p->pattern_weights);
  free(p->eigen_decomp_valid);
  free_ptr_array((void **)p->eigenvecs, p->rate_matrices);
  free_ptr_array((void **)p->inv_eigenvecs, p->rate_matrices);
  free_ptr_array((void **)p->eigenvals, p->rate_matrices);
  free_ptr_array((void **)p->tipchars, p->tips);
  free(p->charmap);
  free(p->tipmap);
  free(p->ttlookup);
  free(q->model_dirty);
  pll_amd_repeats_free(q);
  q->magic = 0;
  free(q);
}

static double ** alloc_rows(unsigned int rows, size_t cols)
{
  unsigned int i;
  double ** a = (double **)calloc(rows ? rows : 1, sizeof(double *));
  if (!a) return NULL;
  for (i = 0; i < rows; ++i)
    if (!(a[i] = (double *)pll_aligned_alloc(cols * sizeof(double), PLL_ALIGNMENT_HIP)))
    {
      free_ptr_array((void **)a, rows);
      return NULL;
    }
    else
      memset(a[i], 0, cols * sizeof(double));
  return a;
}

pll_partition_t * pll_partition_create(unsigned int tips, unsigned int clv_buffers,
                                       unsigned int states, unsigned int sites,
                                       unsigned int rate_matrices, unsigned int prob_matrices,
                                       unsigned int rate_cats, unsigned int scale_buffers,
                                       unsigned int attributes)
{
  unsigned int i;
  pll_amd_partition_t * q;
  pll_partition_t * p;
  pllhip_shape_t sh;
  int rc, ndev = 0;

  /* at most one ISA flag, as in the reference (pll.c:413-418); the flag itself
     selects nothing here */
  if (__builtin_popcount(attributes & PLL_ATTRIB_ARCH_MASK) > 1)
  {
    pll_amd_set_error(PLL_ERROR_PARAM_INVALID, "Multiple architecture flags specified.");
    return NULL;
  }
  /* Ascertainment bias: `states` extra sites behind the alignment (pll.c:492-495).
     With PATTERN_TIP and other than 4 states the reference fills the extra tip
     characters with ASCII values where charmap codes belong (pll.c:886-901), which
     makes every extra site impossible; that combination is refused here rather
     than reproduced. */
  if ((attributes & (PLL_ATTRIB_AB_MASK | PLL_ATTRIB_AB_FLAG)) &&
      (attributes & PLL_ATTRIB_PATTERN_TIP) && states != 4)
  {
    pll_amd_set_error(PLL_ERROR_HIP_UNSUPPORTED,
                      "Ascertainment bias correction with PLL_ATTRIB_PATTERN_TIP needs 4 states; "
                      "use tip CLVs for %u-state data.", states);
    return NULL;
  }
  /* site repeats (own extension, host/repeats.c): 4 states, pattern tips, no asc-bias */
  if ((attributes & PLL_ATTRIB_SITE_REPEATS) &&
      ((states != 4 && states != 20) || !(attributes & PLL_ATTRIB_PATTERN_TIP) ||
       (attributes & (PLL_ATTRIB_AB_MASK | PLL_ATTRIB_AB_FLAG)) ||
       /* (20 states with other category counts than 1, 2, 4 -- their ops run in chunks of the categories, which do
        * not follow row maps -- take the attribute since round 5 and store every CLV per site: the client's code
        * path is the same, the results are the plain partition's) */
       !(states == 20 || rate_cats == 1 || rate_cats == 2 || rate_cats == 4 || rate_cats == 8)))
  {
    pll_amd_set_error(PLL_ERROR_HIP_UNSUPPORTED,
                      "PLL_ATTRIB_SITE_REPEATS needs 4 states (1/2/4/8 rate categories) or 20 states, "
                      "PLL_ATTRIB_PATTERN_TIP and no ascertainment-bias attribute.");
    return NULL;
  }
  if (!states || !sites || !rate_cats || !rate_matrices || (tips + clv_buffers) == 0)
  {
    pll_amd_set_error(PLL_ERROR_PARAM_INVALID, "Empty partition dimensions.");
    return NULL;
  }

  /* no device, no partition: this library has no CPU compute path */
  if (pllhip_device_count(&ndev) || ndev <= 0)
  {
    pll_amd_set_error(PLL_ERROR_HIP_NODEVICE, "No HIP device available: %s",
                      pllhip_last_error());
    return NULL;
  }

  q = (pll_amd_partition_t *)calloc(1, sizeof(pll_amd_partition_t));
  if (!q)
  {
    pll_amd_set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate memory for partition.");
    return NULL;
  }
  p = &q->pub;
  q->magic = PLL_AMD_MAGIC;

  p->tips = tips;
  p->clv_buffers = clv_buffers;
  p->states = states;
  p->sites = sites;
  p->pattern_weight_sum = sites;
  p->rate_matrices = rate_matrices;
  p->prob_matrices = prob_matrices;
  p->rate_cats = rate_cats;
  p->scale_buffers = scale_buffers;
  p->attributes = attributes;
  p->alignment = PLL_ALIGNMENT_HIP;
  p->states_padded = states;
  p->asc_bias_alloc = (attributes & (PLL_ATTRIB_AB_MASK | PLL_ATTRIB_AB_FLAG)) ? 1 : 0;
  q->sites_alloc = p->asc_bias_alloc ? sites + states : sites;

  /* host-side arrays the reference exposes; CLV / scaler mirrors start NULL */
  p->eigen_decomp_valid = (int *)calloc(rate_matrices, sizeof(int));
  p->clv = (double **)calloc(tips + clv_buffers, sizeof(double *));
  p->scale_buffer = (unsigned int **)calloc(scale_buffers ? scale_buffers : 1,
                                            sizeof(unsigned int *));
  p->pmatrix = (double **)calloc(prob_matrices ? prob_matrices : 1, sizeof(double *));
  p->eigenvecs = alloc_rows(rate_matrices, (size_t)states * states);
  p->inv_eigenvecs = alloc_rows(rate_matrices, (size_t)states * states);
  p->eigenvals = alloc_rows(rate_matrices, states);
  p->subst_params = alloc_rows(rate_matrices, (size_t)states * (states - 1) / 2 + 1);
  p->frequencies = alloc_rows(rate_matrices, states);
  p->rates = (double *)calloc(rate_cats, sizeof(double));
  p->rate_weights = (double *)calloc(rate_cats, sizeof(double));
  p->prop_invar = (double *)calloc(rate_matrices, sizeof(double));
  p->pattern_weights = (unsigned int *)calloc(q->sites_alloc, sizeof(unsigned int));
  q->model_dirty = (int *)calloc(rate_matrices, sizeof(int));
  if (!p->eigen_decomp_valid || !p->clv || !p->scale_buffer || !p->pmatrix || !p->eigenvecs ||
      !p->inv_eigenvecs || !p->eigenvals || !p->subst_params || !p->frequencies || !p->rates ||
      !p->rate_weights || !p->prop_invar || !p->pattern_weights || !q->model_dirty)
    goto oom;
  if (prob_matrices)
  {
    /* one contiguous block like the reference (pll.c:559-579) */
    const size_t per = (size_t)states * states * rate_cats;
    p->pmatrix[0] = (double *)pll_aligned_alloc(prob_matrices * per * sizeof(double),
                                                PLL_ALIGNMENT_HIP);
    if (!p->pmatrix[0]) goto oom;
    memset(p->pmatrix[0], 0, prob_matrices * per * sizeof(double));
    for (i = 1; i < prob_matrices; ++i) p->pmatrix[i] = p->pmatrix[i - 1] + per;
  }
  /* defaults (pll.c:742-748, 773-786) */
  for (i = 0; i < rate_cats; ++i) p->rate_weights[i] = 1.0 / rate_cats;
  for (i = 0; i < sites; ++i) p->pattern_weights[i] = 1;
  for (i = 0; i < rate_matrices; ++i) q->model_dirty[i] = 1;
  q->rates_dirty = 1;

  {
    /* host mirrors kept current for small partitions (internal.h: auto_mirror) */
    const char * e = getenv("PLL_AMD_AUTO_MIRROR_MB");
    const double limit_mb = e ? atof(e) : 64.0;
    const double clv_mb = (double)((attributes & PLL_ATTRIB_PATTERN_TIP) ? clv_buffers : tips + clv_buffers) *
                          (double)q->sites_alloc * rate_cats * states * sizeof(double) / (1024.0 * 1024.0);
    q->auto_mirror = limit_mb > 0.0 && clv_mb < limit_mb;
  }
  memset(&sh, 0, sizeof(sh));
  sh.device = default_device();
  sh.states = states;
  sh.rate_cats = rate_cats;
  sh.sites = q->sites_alloc;
  sh.tips = tips;
  sh.clv_buffers = clv_buffers;
  sh.rate_matrices = rate_matrices;
  sh.prob_matrices = prob_matrices;
  sh.scale_buffers = scale_buffers;
  sh.pattern_tip = (attributes & PLL_ATTRIB_PATTERN_TIP) ? 1 : 0;
  sh.rate_scalers = (attributes & PLL_ATTRIB_RATE_SCALERS) ? 1 : 0;
  sh.asc_states = p->asc_bias_alloc ? states : 0;
  {
    int devices[PLL_AMD_MAX_DEVICES];
    const int ndev = device_list(devices);
    if (ndev < 0)
    {
      pll_partition_destroy(p);
      pll_amd_set_error(PLL_ERROR_PARAM_INVALID, "PLL_AMD_DEVICES: expected a list like 0-7 or 0,2,3");
      return NULL;
    }
    if (ndev > 1) rc = pllhip_ctx_create_sharded(&sh, devices, (unsigned int)ndev, &q->ctx);
    else
    {
      if (ndev == 1) sh.device = devices[0];
      rc = pllhip_ctx_create(&sh, &q->ctx);
    }
  }
  if (rc)
  {
    pll_amd_set_error(rc == -1 ? PLL_ERROR_HIP_UNSUPPORTED : PLL_ERROR_HIP_RUNTIME,
                      "Cannot create device context: %s", pllhip_last_error());
    q->ctx = NULL;
    pll_partition_destroy(p);
    return NULL;
  }
  if ((attributes & PLL_ATTRIB_SITE_REPEATS) && !pll_amd_repeats_alloc(q))
  {
    pll_partition_destroy(p);
    pll_amd_set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate site-repeat bookkeeping.");
    return NULL;
  }
  if (p->asc_bias_alloc)
  {
    /* the extra sites start with weight 0 (pll.c:785-786) */
    if ((rc = pllhip_put_pattern_weights(q->ctx, p->pattern_weights)) ||
        (rc = pllhip_set_asc(q->ctx, (int)(attributes & PLL_ATTRIB_AB_MASK), p->pattern_weight_sum)))
    {
      pll_amd_fail_hip(rc, "ascertainment bias setup");
      pll_partition_destroy(p);
      return NULL;
    }
  }
  return p;

oom:
  pll_partition_destroy(p);
  pll_amd_set_error(PLL_ERROR_MEM_ALLOC, "Unable to allocate enough memory.");
  return NULL;
}

/* ---- tip encoding ------------------------------------------------------- */

/* Assign 1-byte codes to the distinct state masks of `map`, in ASCII order of
 * first appearance, extending an existing tipmap (create_charmap pll.c:305-325,
 * update_charmap pll.c:136-260).  4-state data is not remapped: the code is the
 * ambiguity mask itself and maxstates = largest mask + 1. */
static int merge_charmap(pll_partition_t * p, const unsigned int * map)
{
  unsigned int i, j, known = 0, fresh = 0;
  unsigned int seen[PLL_ASCII_SIZE];
  unsigned int nseen = 0;

  if (!p->charmap)
  {
    p->charmap = (unsigned char *)calloc(PLL_ASCII_SIZE, 1);
    p->tipmap = (unsigned int *)calloc(PLL_ASCII_SIZE, sizeof(unsigned int));
    p->tipchars = (unsigned char **)calloc(p->tips, sizeof(unsigned char *));
    if (!p->charmap || !p->tipmap || !p->tipchars)
    {
      pll_amd_set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate charmap for tip-tip precomputation.");
      return PLL_FAILURE;
    }
    for (i = 0; i < p->tips; ++i)
      if (!(p->tipchars[i] = (unsigned char *)calloc(pll_amd_priv(p)->sites_alloc, 1)))
      {
        pll_amd_set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate space for storing tip characters.");
        return PLL_FAILURE;
      }
    p->maxstates = 0;
  }
  else
    while (known < PLL_ASCII_SIZE && p->tipmap[known]) ++known;

  /* count masks not yet in the tipmap */
  for (i = 0; i < PLL_ASCII_SIZE; ++i)
  {
    if (!map[i]) continue;
    for (j = 0; j < known; ++j)
      if (p->tipmap[j] == map[i]) break;
    if (j < known) continue;
    for (j = 0; j < nseen; ++j)
      if (seen[j] == map[i]) break;
    if (j == nseen) { seen[nseen++] = map[i]; ++fresh; }
  }
  if (known + fresh >= PLL_ASCII_SIZE)
  {
    pll_amd_set_error(PLL_ERROR_PARAM_INVALID,
                      "Cannot specify 256 or more states with PLL_ATTRIB_PATTERN_TIP.");
    return PLL_FAILURE;
  }

  memset(p->charmap, 0, PLL_ASCII_SIZE);
  for (i = 0; i < PLL_ASCII_SIZE; ++i)
  {
    unsigned int code;
    if (!map[i]) continue;
    for (code = 0; code < known; ++code)
      if (p->tipmap[code] == map[i]) break;
    if (code == known) p->tipmap[known++] = map[i];
    p->charmap[i] = (unsigned char)code;
  }

  if (fresh || !p->maxstates)
  {
    if (p->states == 4)
    {
      unsigned int top = 0;
      for (i = 0; i < known; ++i)
        if (p->tipmap[i] > top) top = p->tipmap[i];
      p->maxstates = top + 1;
    }
    else
      p->maxstates = known;
    pll_amd_priv(p)->tipmap_dirty = 1;
  }
  return PLL_SUCCESS;
}

static int illegal_state(char c)
{
  pll_amd_set_error(PLL_ERROR_TIPDATA_ILLEGALSTATE, "Illegal state code in tip \"%c\"", c);
  return PLL_FAILURE;
}

int pll_set_tip_states(pll_partition_t * p, unsigned int tip_index, const unsigned int * map,
                       const char * sequence)
{
  pll_amd_partition_t * q = pll_amd_priv(p);
  unsigned int i, j;
  int rc;

  if (tip_index >= p->tips)
  {
    pll_amd_set_error(PLL_ERROR_PARAM_INVALID, "Tip index %u out of range", tip_index);
    return PLL_FAILURE;
  }

  if (p->attributes & PLL_ATTRIB_PATTERN_TIP)
  {
    unsigned char * codes;
    if (!merge_charmap(p, map)) return PLL_FAILURE;
    codes = p->tipchars[tip_index];
    for (i = 0; i < p->sites; ++i)
    {
      const unsigned int m = map[(unsigned char)sequence[i]];
      if (!m) return illegal_state(sequence[i]);
      /* 4 states: the mask is the code (pll.c:825-845); else the charmap code (pll.c:862-883) */
      codes[i] = (p->states == 4) ? (unsigned char)m : p->charmap[(unsigned char)sequence[i]];
    }
    /* ascertainment sites: site sites+k shows state k (pll.c:847-855; 4 states only) */
    if (p->asc_bias_alloc)
      for (i = 0; i < p->states; ++i) codes[p->sites + i] = (unsigned char)(1u << i);
    if ((rc = pllhip_put_tipchars(q->ctx, tip_index, codes)))
      return pll_amd_fail_hip(rc, "upload of tip characters");
    pll_amd_repeats_tip_changed(q, tip_index);
    return PLL_SUCCESS;
  }

  /* tip as CLV: one 0/1 vector per site, replicated over the categories on the
     device (set_tipclv pll.c:905-939) */
  {
    const unsigned int S = p->states;
    double * v = (double *)calloc((size_t)q->sites_alloc * S, sizeof(double));
    if (!v)
    {
      pll_amd_set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate tip vector staging.");
      return PLL_FAILURE;
    }
    for (i = 0; i < p->sites; ++i)
    {
      unsigned int m = map[(unsigned char)sequence[i]];
      if (!m)
      {
        free(v);
        return illegal_state(sequence[i]);
      }
      for (j = 0; j < S; ++j, m >>= 1) v[(size_t)i * S + j] = (double)(m & 1u);
    }
    /* ascertainment sites: unit vectors (pll.c:943-961) */
    if (p->asc_bias_alloc)
      for (i = 0; i < S; ++i) v[((size_t)p->sites + i) * S + i] = 1.0;
    rc = pllhip_put_tip_clv_persite(q->ctx, tip_index, v, S);
    free(v);
    if (rc) return pll_amd_fail_hip(rc, "upload of tip CLV");
    if (PLL_AMD_MIRRORS(p) && !pll_amd_sync_clv(p, tip_index)) return PLL_FAILURE;
  }
  return PLL_SUCCESS;
}

int pll_set_tip_clv(pll_partition_t * p, unsigned int tip_index, const double * clv, int padding)
{
  int rc;
  (void)padding; /* states_padded == states here, both layouts coincide */
  if (p->attributes & PLL_ATTRIB_PATTERN_TIP)
  {
    pll_amd_set_error(PLL_ERROR_TIPDATA_ILLEGALFUNCTION,
                      "Cannot use pll_set_tip_clv with PLL_ATTRIB_PATTERN_TIP.");
    return PLL_FAILURE;
  }
  if (tip_index >= p->tips + p->clv_buffers)
  {
    pll_amd_set_error(PLL_ERROR_PARAM_INVALID, "CLV index %u out of range", tip_index);
    return PLL_FAILURE;
  }
  if (p->asc_bias_alloc)
  {
    /* the caller's vector covers the alignment; the ascertainment sites get unit
       vectors (pll.c:1028-1043) */
    const unsigned int S = p->states;
    unsigned int i;
    double * v = (double *)calloc((size_t)pll_amd_priv(p)->sites_alloc * S, sizeof(double));
    if (!v)
    {
      pll_amd_set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate tip vector staging.");
      return PLL_FAILURE;
    }
    memcpy(v, clv, (size_t)p->sites * S * sizeof(double));
    for (i = 0; i < S; ++i) v[((size_t)p->sites + i) * S + i] = 1.0;
    rc = pllhip_put_tip_clv_persite(pll_amd_priv(p)->ctx, tip_index, v, S);
    free(v);
  }
  else
    rc = pllhip_put_tip_clv_persite(pll_amd_priv(p)->ctx, tip_index, clv, p->states);
  if (rc) return pll_amd_fail_hip(rc, "upload of tip CLV");
  if (PLL_AMD_MIRRORS(p) && !pll_amd_sync_clv(p, tip_index)) return PLL_FAILURE;
  return PLL_SUCCESS;
}

void pll_set_pattern_weights(pll_partition_t * p, const unsigned int * w)
{
  unsigned int i;
  int rc;
  memcpy(p->pattern_weights, w, (size_t)p->sites * sizeof(unsigned int));
  p->pattern_weight_sum = 0;
  for (i = 0; i < p->sites; ++i) p->pattern_weight_sum += w[i];
  rc = pllhip_put_pattern_weights(pll_amd_priv(p)->ctx, p->pattern_weights);
  if (!rc && p->asc_bias_alloc)
    rc = pllhip_set_asc(pll_amd_priv(p)->ctx, (int)(p->attributes & PLL_ATTRIB_AB_MASK),
                        p->pattern_weight_sum);
  if (rc) pll_amd_fail_hip(rc, "upload of pattern weights");
}

/* pll.c:1061-1107 */
int pll_set_asc_bias_type(pll_partition_t * p, int asc_bias_type)
{
  unsigned int i;
  int rc, pinv = 0;
  const int bits = asc_bias_type & PLL_ATTRIB_AB_MASK;
  if (!p->asc_bias_alloc)
  {
    pll_amd_set_error(PLL_ERROR_AB_NOSUPPORT,
                      "Partition was not created with ascertainment bias support");
    return PLL_FAILURE;
  }
  for (i = 0; i < p->rate_matrices; ++i) pinv |= (p->prop_invar[i] > 0);
  if (asc_bias_type != 0 && pinv)
  {
    pll_amd_set_error(PLL_ERROR_INVAR_INCOMPAT,
                      "Invariant sites are not compatible with asc bias correction");
    return PLL_FAILURE;
  }
  if (bits != asc_bias_type)
  {
    pll_amd_set_error(PLL_ERROR_AB_INVALIDMETHOD, "Illegal ascertainment bias algorithm \"%d\"",
                      asc_bias_type);
    return PLL_FAILURE;
  }
  p->attributes = (p->attributes & ~(unsigned int)PLL_ATTRIB_AB_MASK) | (unsigned int)bits;
  if ((rc = pllhip_set_asc(pll_amd_priv(p)->ctx, bits, p->pattern_weight_sum)))
    return pll_amd_fail_hip(rc, "ascertainment bias type");
  return PLL_SUCCESS;
}

/* pll.c:1109-1116: how often each state's invariant pattern would have been seen */
void pll_set_asc_state_weights(pll_partition_t * p, const unsigned int * state_weights)
{
  int rc;
  if (!p->asc_bias_alloc)
  {
    pll_amd_set_error(PLL_ERROR_AB_NOSUPPORT,
                      "Partition was not created with ascertainment bias support");
    return;
  }
  memcpy(p->pattern_weights + p->sites, state_weights, (size_t)p->states * sizeof(unsigned int));
  rc = pllhip_put_pattern_weights(pll_amd_priv(p)->ctx, p->pattern_weights);
  if (rc) pll_amd_fail_hip(rc, "upload of state weights");
}

/* ---- host mirrors -------------------------------------------------------- */

int pll_amd_sync_clv(pll_partition_t * p, unsigned int idx)
{
  pll_amd_partition_t * q = pll_amd_priv(p);
  const size_t n = (size_t)q->sites_alloc * p->rate_cats * p->states;
  int rc;
  if (idx >= p->tips + p->clv_buffers ||
      ((p->attributes & PLL_ATTRIB_PATTERN_TIP) && idx < p->tips))
  {
    pll_amd_set_error(PLL_ERROR_PARAM_INVALID, "CLV index %u has no CLV", idx);
    return PLL_FAILURE;
  }
  if (!p->clv[idx] && !(p->clv[idx] = (double *)pll_aligned_alloc(n * sizeof(double),
                                                                 PLL_ALIGNMENT_HIP)))
  {
    pll_amd_set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate CLV mirror.");
    return PLL_FAILURE;
  }
  if ((rc = pllhip_get_clv(q->ctx, idx, p->clv[idx]))) return pll_amd_fail_hip(rc, "CLV download");
  if (q->rep && q->rep[idx].classes)
  {
    /* stored by class on the device: the mirror shows one row per site */
    const unsigned int * sid = pll_amd_repeats_site_id(p, idx);
    if (!sid || !pll_amd_repeats_expand(p->clv[idx], sid, q->rep[idx].classes, p->sites,
                                        (size_t)p->rate_cats * p->states * sizeof(double)))
      return PLL_FAILURE;
  }
  return PLL_SUCCESS;
}

int pll_amd_sync_scaler(pll_partition_t * p, unsigned int idx)
{
  pll_amd_partition_t * q = pll_amd_priv(p);
  const size_t n = (size_t)q->sites_alloc *
                   ((p->attributes & PLL_ATTRIB_RATE_SCALERS) ? p->rate_cats : 1);
  int rc;
  if (idx >= p->scale_buffers)
  {
    pll_amd_set_error(PLL_ERROR_PARAM_INVALID, "Scaler index %u out of range", idx);
    return PLL_FAILURE;
  }
  if (!p->scale_buffer[idx] &&
      !(p->scale_buffer[idx] = (unsigned int *)calloc(n, sizeof(unsigned int))))
  {
    pll_amd_set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate scaler mirror.");
    return PLL_FAILURE;
  }
  if ((rc = pllhip_get_scaler(q->ctx, idx, p->scale_buffer[idx])))
    return pll_amd_fail_hip(rc, "scaler download");
  if (q->rep && q->scaler_owner[idx] >= 0 && q->rep[q->scaler_owner[idx]].classes)
  {
    /* written together with a CLV that is stored by class: same expansion */
    const unsigned int owner = (unsigned int)q->scaler_owner[idx];
    const size_t per = (p->attributes & PLL_ATTRIB_RATE_SCALERS) ? p->rate_cats : 1;
    const unsigned int * sid = pll_amd_repeats_site_id(p, owner);
    if (!sid || !pll_amd_repeats_expand(p->scale_buffer[idx], sid, q->rep[owner].classes, p->sites,
                                        per * sizeof(unsigned int)))
      return PLL_FAILURE;
  }
  return PLL_SUCCESS;
}

int pll_amd_sync_pmatrix(pll_partition_t * p, unsigned int idx)
{
  int rc;
  if (idx >= p->prob_matrices)
  {
    pll_amd_set_error(PLL_ERROR_PARAM_INVALID, "P-matrix index %u out of range", idx);
    return PLL_FAILURE;
  }
  if ((rc = pllhip_get_pmatrix(pll_amd_priv(p)->ctx, idx, p->pmatrix[idx])))
    return pll_amd_fail_hip(rc, "P-matrix download");
  return PLL_SUCCESS;
}

int pll_amd_wait(pll_partition_t * p)
{
  int rc = pllhip_wait(pll_amd_priv(p)->ctx);
  if (rc) return pll_amd_fail_hip(rc, "stream synchronize");
  return PLL_SUCCESS;
}

const char * pll_amd_rccl_path(void) { return pllhip_rccl_path(); }

int pll_amd_comm_unique_id(void * id)
{
  int rc = pllhip_comm_unique_id(id);
  if (rc) return pll_amd_fail_hip(rc, "RCCL unique id");
  return PLL_SUCCESS;
}

int pll_amd_comm_init(pll_partition_t * p, int rank, int nranks, const void * id)
{
  int rc = pllhip_comm_init(pll_amd_priv(p)->ctx, rank, nranks, id);
  if (rc) return pll_amd_fail_hip(rc, "RCCL communicator");
  return PLL_SUCCESS;
}

unsigned long long pll_amd_comm_reduces(pll_partition_t * p) { return pllhip_comm_reduces(pll_amd_priv(p)->ctx); }

int pll_amd_timer_start(pll_partition_t * p)
{
  int rc = pllhip_timer_start(pll_amd_priv(p)->ctx);
  if (rc) return pll_amd_fail_hip(rc, "timer start");
  return PLL_SUCCESS;
}

int pll_amd_timer_stop_ms(pll_partition_t * p, float * ms)
{
  int rc = pllhip_timer_stop_ms(pll_amd_priv(p)->ctx, ms);
  if (rc) return pll_amd_fail_hip(rc, "timer stop");
  return PLL_SUCCESS;
}

unsigned int pll_amd_timer_shard_ms(pll_partition_t * p, float * ms, unsigned int cap)
{
  return pllhip_timer_shard_ms(pll_amd_priv(p)->ctx, ms, cap);
}

int pll_amd_profile_enable(pll_partition_t * p, int on)
{
  int rc = pllhip_profile_enable(pll_amd_priv(p)->ctx, on);
  if (rc) return pll_amd_fail_hip(rc, "profile enable");
  return PLL_SUCCESS;
}

int pll_amd_scaling_certificate(pll_partition_t * p, unsigned long long * stats4)
{
  int rc = pllhip_cert_stats(pll_amd_priv(p)->ctx, stats4);
  if (rc) return pll_amd_fail_hip(rc, "scaling certificate");
  return PLL_SUCCESS;
}

int pll_amd_write_ceiling(pll_partition_t * p, const pll_operation_t * operations, unsigned int count, unsigned int reps,
                          float * ms_per_pass, double * bytes_per_pass)
{
  int rc = pllhip_write_ceiling(pll_amd_priv(p)->ctx, (const pllhip_op_t *)operations, count, reps, ms_per_pass, bytes_per_pass);
  if (rc) return pll_amd_fail_hip(rc, "write ceiling");
  return PLL_SUCCESS;
}

int pll_amd_placement_info(pll_partition_t * p, double * gbs, unsigned int cap, int * kept)
{
  return pllhip_placement_info(pll_amd_priv(p)->ctx, gbs, cap, kept);
}

int pll_amd_arena_fill_bandwidth(pll_partition_t * p, double * gbs)
{
  int rc = pllhip_arena_fill_bandwidth(pll_amd_priv(p)->ctx, gbs);
  if (rc) return pll_amd_fail_hip(rc, "arena fill bandwidth");
  return PLL_SUCCESS;
}

int pll_amd_list_kinds(pll_partition_t * p, unsigned int * kinds8)
{
  int rc = pllhip_aa_list_kinds(pll_amd_priv(p)->ctx, kinds8);
  if (rc) return pll_amd_fail_hip(rc, "list kinds");
  return PLL_SUCCESS;
}

int pll_amd_profile_read(pll_partition_t * p, unsigned int * launches, double * total_ms)
{
  int rc = pllhip_profile_read(pll_amd_priv(p)->ctx, launches, total_ms);
  if (rc) return pll_amd_fail_hip(rc, "profile read");
  return PLL_SUCCESS;
}
