/*
 * pll_amd.h -- public C interface of libpll_amd, the MI355X-native
 * Felsenstein-pruning library.
 *
 * The library is a drop-in for ONE path of libpll 0.3.2: partition container
 * -> P-matrices -> conditional-likelihood-vector (CLV) updates -> edge/root
 * log-likelihood -> sumtable / branch-length derivatives.  Every function
 * below keeps the name, argument order, argument meaning and error behaviour
 * of the reference declaration it replaces (cited as pll.h:<line>, relative to
 * the reference's src/ directory), so a client written against the
 * reference's header links against libpll_amd.so unchanged.
 *
 * What is different underneath:
 *   - all likelihood arithmetic runs in hand-written HIP kernels on gfx950;
 *     CLVs, scale buffers, tip characters and P-matrices live in HBM;
 *   - there is NO CPU compute path.  If no HIP device can be opened,
 *     pll_partition_create fails with PLL_ERROR_HIP_* (it never falls back);
 *   - partition->clv[i], ->scale_buffer[i] and ->pmatrix[i] are host MIRRORS,
 *     refreshed only on demand: pll_show_clv / pll_show_pmatrix refresh what
 *     they print; other readers call pll_amd_sync_* first (bottom of file).
 *   - the ISA bits of `attributes` (PLL_ATTRIB_ARCH_*) are accepted and
 *     ignored: states_padded == states and alignment == 16 always.
 */
#ifndef PLL_AMD_H_
#define PLL_AMD_H_

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLL_EXPORT __attribute__((visibility("default")))

/* ---- constants: values fixed by the reference ABI (pll.h:75-167) ---- */

#define PLL_FAILURE 0
#define PLL_SUCCESS 1
#define PLL_FALSE 0
#define PLL_TRUE 1

#define PLL_ASCII_SIZE 256

/* 2^256 and 2^-256: numerical scaling of CLVs (pll.h:89-93) */
#define PLL_SCALE_FACTOR 0x1p+256
#define PLL_SCALE_THRESHOLD 0x1p-256
#define PLL_SCALE_BUFFER_NONE (-1)
/* per-rate scaling: cap on the scaler difference undone in lnL (pll.h:97) */
#define PLL_SCALE_RATE_MAXDIFF 4

#define PLL_MISC_EPSILON 1e-8

#define PLL_ALIGNMENT_CPU 8
#define PLL_ALIGNMENT_SSE 16
#define PLL_ALIGNMENT_AVX 32
#define PLL_ALIGNMENT_HIP 16

/* attribute word (pll.h:106-122); bits 0-3 are accepted but select nothing */
#define PLL_ATTRIB_ARCH_CPU 0
#define PLL_ATTRIB_ARCH_SSE (1 << 0)
#define PLL_ATTRIB_ARCH_AVX (1 << 1)
#define PLL_ATTRIB_ARCH_AVX2 (1 << 2)
#define PLL_ATTRIB_ARCH_AVX512 (1 << 3)
#define PLL_ATTRIB_ARCH_MASK 0xF
#define PLL_ATTRIB_PATTERN_TIP (1 << 4)
#define PLL_ATTRIB_AB_LEWIS (1 << 5)
#define PLL_ATTRIB_AB_FELSENSTEIN (2 << 5)
#define PLL_ATTRIB_AB_STAMATAKIS (3 << 5)
#define PLL_ATTRIB_AB_MASK (7 << 5)
#define PLL_ATTRIB_AB_FLAG (1 << 8)
#define PLL_ATTRIB_RATE_SCALERS (1 << 9)
/* Not in libpll 0.3.2 (later versions define this bit): sites that cannot be told
 * apart below a node share one CLV entry there.  4- or 20-state data with
 * PLL_ATTRIB_PATTERN_TIP; every result equals the one without the bit. */
#define PLL_ATTRIB_SITE_REPEATS (1 << 10)

/* error codes shared with the reference (pll.h:137-167) */
#define PLL_ERROR_FILE_OPEN 100
#define PLL_ERROR_FILE_SEEK 101
#define PLL_ERROR_FILE_EOF 102
#define PLL_ERROR_FASTA_ILLEGALCHAR 103
#define PLL_ERROR_FASTA_UNPRINTABLECHAR 104
#define PLL_ERROR_FASTA_INVALIDHEADER 105
#define PLL_ERROR_PHYLIP_SYNTAX 106
#define PLL_ERROR_PHYLIP_LONGSEQ 107
#define PLL_ERROR_PHYLIP_NONALIGNED 108
#define PLL_ERROR_PHYLIP_ILLEGALCHAR 109
#define PLL_ERROR_PHYLIP_UNPRINTABLECHAR 110
#define PLL_ERROR_MEM_ALLOC 112
#define PLL_ERROR_PARAM_INVALID 113
#define PLL_ERROR_TIPDATA_ILLEGALSTATE 114
#define PLL_ERROR_TIPDATA_ILLEGALFUNCTION 115
#define PLL_ERROR_INVAR_INCOMPAT 117
#define PLL_ERROR_INVAR_PROPORTION 118
#define PLL_ERROR_INVAR_PARAMINDEX 119
#define PLL_ERROR_INVAR_NONEFOUND 120
#define PLL_ERROR_AB_INVALIDMETHOD 121
#define PLL_ERROR_AB_NOSUPPORT 122
#define PLL_ERROR_EINVAL 130
/* new codes of this library, outside the reference's range */
#define PLL_ERROR_HIP_NODEVICE 200
#define PLL_ERROR_HIP_RUNTIME 201
#define PLL_ERROR_HIP_UNSUPPORTED 202
#define PLL_ERROR_HIP_SUMTABLE_EVICTED 203

#define PLL_GAMMA_RATES_MEAN 0
#define PLL_GAMMA_RATES_MEDIAN 1

#define PLL_TREE_TRAVERSE_POSTORDER 1
#define PLL_TREE_TRAVERSE_PREORDER 2

/* ---- data types: field order and types are the reference ABI ---- */

/* pll.h:202-244.  Clients read these fields directly, so the layout is frozen.
 * The device state hangs off a private tail allocated behind this struct. */
typedef struct pll_partition
{
  unsigned int tips;
  unsigned int clv_buffers;
  unsigned int states;
  unsigned int sites;
  unsigned int pattern_weight_sum;
  unsigned int rate_matrices;
  unsigned int prob_matrices;
  unsigned int rate_cats;
  unsigned int scale_buffers;
  unsigned int attributes;

  size_t alignment;
  unsigned int states_padded;

  double ** clv;                /* host mirrors, NULL until synced        */
  double ** pmatrix;            /* host mirrors of the device P-matrices  */
  double * rates;
  double * rate_weights;
  double ** subst_params;
  unsigned int ** scale_buffer; /* host mirrors, NULL until synced        */
  double ** frequencies;
  double * prop_invar;
  int * invariant;
  unsigned int * pattern_weights;

  int * eigen_decomp_valid;
  double ** eigenvecs;
  double ** inv_eigenvecs;
  double ** eigenvals;

  unsigned int maxstates;
  unsigned char ** tipchars;    /* host copies of the encoded tip sequences */
  unsigned char * charmap;
  double * ttlookup;            /* unused: tip-tip products are formed on the fly */
  unsigned int * tipmap;

  int asc_bias_alloc;
} pll_partition_t;

/* ---- alignment readers (pll.h:271-308; layouts frozen, fields are public) ---- */
#define PLL_LINEALLOC 2048

typedef struct pll_msa_s
{
  int count;
  int length;
  char ** sequence;
  char ** label;
} pll_msa_t;

typedef struct pll_fasta
{
  FILE * fp;
  char line[PLL_LINEALLOC];
  const unsigned int * chrstatus;
  long no;
  long filesize;
  long lineno;
  long stripped_count;
  long stripped[256];
} pll_fasta_t;

typedef struct pll_phylip_s
{
  FILE * fp;
  char * line;
  size_t line_size;
  size_t line_maxsize;
  char buffer[PLL_LINEALLOC];
  const unsigned int * chrstatus;
  long no;
  long filesize;
  long lineno;
  long stripped_count;
  long stripped[256];
} pll_phylip_t;

/* pll.h:183-200: host CPU feature record.  Nothing in this library dispatches on
 * it (the kernels run on the GPU); kept because clients probe and print it. */
typedef struct pll_hardware_s
{
  int init;
  int altivec_present;
  int mmx_present;
  int sse_present;
  int sse2_present;
  int sse3_present;
  int ssse3_present;
  int sse41_present;
  int sse42_present;
  int popcnt_present;
  int avx_present;
  int avx2_present;
} pll_hardware_t;

/* pll.h:249-259: one pruning step parent <- (child1, child2) */
typedef struct pll_operation
{
  unsigned int parent_clv_index;
  int parent_scaler_index;
  unsigned int child1_clv_index;
  unsigned int child1_matrix_index;
  int child1_scaler_index;
  unsigned int child2_clv_index;
  unsigned int child2_matrix_index;
  int child2_scaler_index;
} pll_operation_t;

/* tree nodes as clients build them (pll.h:312-350); layouts frozen.  An inner
 * node of an unrooted tree is a ring of three pll_unode_t linked by `next`. */
typedef struct pll_unode_s
{
  char * label;
  double length;
  unsigned int node_index;
  unsigned int clv_index;
  int scaler_index;
  unsigned int pmatrix_index;
  struct pll_unode_s * next;
  struct pll_unode_s * back;
  void * data;
} pll_unode_t;

typedef struct pll_rnode_s
{
  char * label;
  double length;
  unsigned int node_index;
  unsigned int clv_index;
  int scaler_index;
  unsigned int pmatrix_index;
  struct pll_rnode_s * left;
  struct pll_rnode_s * right;
  struct pll_rnode_s * parent;
  void * data;
} pll_rnode_t;

/* ---- global data (pll.h:470-522) ---- */

PLL_EXPORT extern __thread int pll_errno;
PLL_EXPORT extern __thread char pll_errmsg[200];

PLL_EXPORT extern const unsigned int pll_map_bin[256];
PLL_EXPORT extern const unsigned int pll_map_nt[256];
PLL_EXPORT extern const unsigned int pll_map_aa[256];
/* character classes for the sequence readers below: 0 = stripped and counted,
 * 1 = data, 2 = fatal, 3 = stripped silently (maps.c of the reference) */
PLL_EXPORT extern const unsigned int pll_map_fasta[256];
PLL_EXPORT extern const unsigned int pll_map_phylip[256];

PLL_EXPORT extern const double pll_aa_rates_dayhoff[190];
PLL_EXPORT extern const double pll_aa_freqs_dayhoff[20];
PLL_EXPORT extern const double pll_aa_rates_lg[190];
PLL_EXPORT extern const double pll_aa_freqs_lg[20];
PLL_EXPORT extern const double pll_aa_rates_dcmut[190];
PLL_EXPORT extern const double pll_aa_freqs_dcmut[20];
PLL_EXPORT extern const double pll_aa_rates_jtt[190];
PLL_EXPORT extern const double pll_aa_freqs_jtt[20];
PLL_EXPORT extern const double pll_aa_rates_mtrev[190];
PLL_EXPORT extern const double pll_aa_freqs_mtrev[20];
PLL_EXPORT extern const double pll_aa_rates_wag[190];
PLL_EXPORT extern const double pll_aa_freqs_wag[20];
PLL_EXPORT extern const double pll_aa_rates_rtrev[190];
PLL_EXPORT extern const double pll_aa_freqs_rtrev[20];
PLL_EXPORT extern const double pll_aa_rates_cprev[190];
PLL_EXPORT extern const double pll_aa_freqs_cprev[20];
PLL_EXPORT extern const double pll_aa_rates_vt[190];
PLL_EXPORT extern const double pll_aa_freqs_vt[20];
PLL_EXPORT extern const double pll_aa_rates_blosum62[190];
PLL_EXPORT extern const double pll_aa_freqs_blosum62[20];
PLL_EXPORT extern const double pll_aa_rates_mtmam[190];
PLL_EXPORT extern const double pll_aa_freqs_mtmam[20];
PLL_EXPORT extern const double pll_aa_rates_mtart[190];
PLL_EXPORT extern const double pll_aa_freqs_mtart[20];
PLL_EXPORT extern const double pll_aa_rates_mtzoa[190];
PLL_EXPORT extern const double pll_aa_freqs_mtzoa[20];
PLL_EXPORT extern const double pll_aa_rates_pmb[190];
PLL_EXPORT extern const double pll_aa_freqs_pmb[20];
PLL_EXPORT extern const double pll_aa_rates_hivb[190];
PLL_EXPORT extern const double pll_aa_freqs_hivb[20];
PLL_EXPORT extern const double pll_aa_rates_hivw[190];
PLL_EXPORT extern const double pll_aa_freqs_hivw[20];
PLL_EXPORT extern const double pll_aa_rates_jttdcmut[190];
PLL_EXPORT extern const double pll_aa_freqs_jttdcmut[20];
PLL_EXPORT extern const double pll_aa_rates_flu[190];
PLL_EXPORT extern const double pll_aa_freqs_flu[20];
PLL_EXPORT extern const double pll_aa_rates_stmtrev[190];
PLL_EXPORT extern const double pll_aa_freqs_stmtrev[20];
/* four-matrix mixtures (pll.h:499-522): one matrix + frequency set per rate category */
PLL_EXPORT extern const double pll_aa_rates_lg4m[4][190];
PLL_EXPORT extern const double pll_aa_rates_lg4x[4][190];
PLL_EXPORT extern const double pll_aa_freqs_lg4m[4][20];
PLL_EXPORT extern const double pll_aa_freqs_lg4x[4][20];

/* ---- partition container (replaces pll.h:530-555) ---- */

PLL_EXPORT pll_partition_t * pll_partition_create(unsigned int tips,
                                                  unsigned int clv_buffers,
                                                  unsigned int states,
                                                  unsigned int sites,
                                                  unsigned int rate_matrices,
                                                  unsigned int prob_matrices,
                                                  unsigned int rate_cats,
                                                  unsigned int scale_buffers,
                                                  unsigned int attributes);
PLL_EXPORT void pll_partition_destroy(pll_partition_t * partition);

PLL_EXPORT int pll_set_tip_states(pll_partition_t * partition,
                                  unsigned int tip_index,
                                  const unsigned int * map,
                                  const char * sequence);
PLL_EXPORT int pll_set_tip_clv(pll_partition_t * partition,
                               unsigned int tip_index,
                               const double * clv,
                               int padding);
PLL_EXPORT void pll_set_pattern_weights(pll_partition_t * partition,
                                        const unsigned int * pattern_weights);

/* ---- model parameters (replaces pll.h:569-597) ---- */

PLL_EXPORT void pll_set_subst_params(pll_partition_t * partition,
                                     unsigned int params_index,
                                     const double * params);
PLL_EXPORT void pll_set_frequencies(pll_partition_t * partition,
                                    unsigned int params_index,
                                    const double * frequencies);
PLL_EXPORT void pll_set_category_rates(pll_partition_t * partition,
                                       const double * rates);
PLL_EXPORT void pll_set_category_weights(pll_partition_t * partition,
                                         const double * rate_weights);
PLL_EXPORT int pll_update_eigen(pll_partition_t * partition,
                                unsigned int params_index);
PLL_EXPORT int pll_update_prob_matrices(pll_partition_t * partition,
                                        const unsigned int * params_index,
                                        const unsigned int * matrix_indices,
                                        const double * branch_lengths,
                                        unsigned int count);
PLL_EXPORT unsigned int pll_count_invariant_sites(pll_partition_t * partition,
                                                  unsigned int * state_inv_count);
PLL_EXPORT int pll_update_invariant_sites(pll_partition_t * partition);
PLL_EXPORT int pll_update_invariant_sites_proportion(pll_partition_t * partition,
                                                     unsigned int params_index,
                                                     double prop_invar);

/* fasta.c:40-324: one record per call; *head and *seq are malloc'ed for the caller */
PLL_EXPORT pll_fasta_t * pll_fasta_open(const char * filename, const unsigned int * map);
PLL_EXPORT int pll_fasta_getnext(pll_fasta_t * fd, char ** head, long * head_len, char ** seq,
                                 long * seq_len, long * seqno);
PLL_EXPORT void pll_fasta_close(pll_fasta_t * fd);
PLL_EXPORT long pll_fasta_getfilesize(const pll_fasta_t * fd);
PLL_EXPORT long pll_fasta_getfilepos(pll_fasta_t * fd);
PLL_EXPORT int pll_fasta_rewind(pll_fasta_t * fd);
/* phylip.c:282-730: whole alignments, sequential or interleaved layout */
PLL_EXPORT pll_phylip_t * pll_phylip_open(const char * filename, const unsigned int * map);
PLL_EXPORT int pll_phylip_rewind(pll_phylip_t * fd);
PLL_EXPORT void pll_phylip_close(pll_phylip_t * fd);
PLL_EXPORT pll_msa_t * pll_phylip_parse_interleaved(pll_phylip_t * fd);
PLL_EXPORT pll_msa_t * pll_phylip_parse_sequential(pll_phylip_t * fd);
PLL_EXPORT void pll_msa_destroy(pll_msa_t * msa);

/* hardware.c:159-189 */
PLL_EXPORT extern pll_hardware_t pll_hardware;
PLL_EXPORT int pll_hardware_probe(void);
PLL_EXPORT void pll_hardware_dump(void);
PLL_EXPORT void pll_hardware_ignore(void);
/* site repeats (PLL_ATTRIB_SITE_REPEATS): number of classes the CLV is stored in, 0 when
 * it is stored per site; and the identification step itself (distinct pairs of child
 * classes, numbered by first appearance; returns 0 beyond `max` classes) */
PLL_EXPORT unsigned int pll_amd_repeats_classes(const pll_partition_t * partition,
                                                unsigned int clv_index);
PLL_EXPORT unsigned int pll_amd_identify_repeats(const unsigned int * ida, unsigned int na,
                                                 const unsigned int * idb, unsigned int nb,
                                                 unsigned int sites, unsigned int max,
                                                 unsigned int * site_id, unsigned int * lrow,
                                                 unsigned int * rrow);

/* pll.c:1061-1116: ascertainment-bias correction of a partition created with
 * PLL_ATTRIB_AB_FLAG or one of PLL_ATTRIB_AB_LEWIS / _FELSENSTEIN / _STAMATAKIS.
 * (With PLL_ATTRIB_PATTERN_TIP only 4-state data is accepted, see DESIGN.md.) */
PLL_EXPORT int pll_set_asc_bias_type(pll_partition_t * partition, int asc_bias_type);
PLL_EXPORT void pll_set_asc_state_weights(pll_partition_t * partition,
                                          const unsigned int * state_weights);

PLL_EXPORT void * pll_aligned_alloc(size_t size, size_t alignment);
PLL_EXPORT void pll_aligned_free(void * ptr);

/* ---- the hot path (replaces pll.h:607-646) ---- */

PLL_EXPORT void pll_update_partials(pll_partition_t * partition,
                                    const pll_operation_t * operations,
                                    unsigned int count);

PLL_EXPORT double pll_compute_root_loglikelihood(pll_partition_t * partition,
                                                 unsigned int clv_index,
                                                 int scaler_index,
                                                 const unsigned int * freqs_indices,
                                                 double * persite_lnl);

PLL_EXPORT double pll_compute_edge_loglikelihood(pll_partition_t * partition,
                                                 unsigned int parent_clv_index,
                                                 int parent_scaler_index,
                                                 unsigned int child_clv_index,
                                                 int child_scaler_index,
                                                 unsigned int matrix_index,
                                                 const unsigned int * freqs_indices,
                                                 double * persite_lnl);

/* `sumtable` is a caller-owned host buffer as in the reference.  The library
 * keeps the authoritative copy on the device, keyed by this pointer, and
 * writes the host buffer only when pll_amd_set_mirror_mode(1) is active or
 * pll_amd_sync_sumtable is called.  One device table per live host buffer (one
 * per branch is fine), up to a byte budget (32 GiB worth, at least 4; env
 * PLL_AMD_SUMTABLE_SLOTS); beyond it the least recently used device table is
 * recycled, and a later pll_compute_likelihood_derivatives / pll_amd_sync_sumtable
 * on ITS buffer fails with PLL_ERROR_HIP_SUMTABLE_EVICTED (203) -- call
 * pll_update_sumtable again.  A buffer the library has never seen is taken as
 * filled by the caller and uploaded. */
PLL_EXPORT int pll_update_sumtable(pll_partition_t * partition,
                                   unsigned int parent_clv_index,
                                   unsigned int child_clv_index,
                                   int parent_scaler_index,
                                   int child_scaler_index,
                                   const unsigned int * params_indices,
                                   double * sumtable);

PLL_EXPORT int pll_compute_likelihood_derivatives(pll_partition_t * partition,
                                                  int parent_scaler_index,
                                                  int child_scaler_index,
                                                  double branch_length,
                                                  const unsigned int * params_indices,
                                                  const double * sumtable,
                                                  double * d_f,
                                                  double * dd_f);

/* ---- the steps either side of the path: tree -> op list, and
 * alignment -> unique site patterns + weights (replaces pll.h:725,731,788,794
 * and :1735) ---- */

PLL_EXPORT int pll_utree_traverse(pll_unode_t * root,
                                  int traversal,
                                  int (*cbtrav)(pll_unode_t *),
                                  pll_unode_t ** outbuffer,
                                  unsigned int * trav_size);
PLL_EXPORT void pll_utree_create_operations(pll_unode_t * const * trav_buffer,
                                            unsigned int trav_buffer_size,
                                            double * branches,
                                            unsigned int * pmatrix_indices,
                                            pll_operation_t * ops,
                                            unsigned int * matrix_count,
                                            unsigned int * ops_count);
PLL_EXPORT int pll_rtree_traverse(pll_rnode_t * root,
                                  int traversal,
                                  int (*cbtrav)(pll_rnode_t *),
                                  pll_rnode_t ** outbuffer,
                                  unsigned int * trav_size);
PLL_EXPORT void pll_rtree_create_operations(pll_rnode_t * const * trav_buffer,
                                            unsigned int trav_buffer_size,
                                            double * branches,
                                            unsigned int * pmatrix_indices,
                                            pll_operation_t * ops,
                                            unsigned int * matrix_count,
                                            unsigned int * ops_count);
PLL_EXPORT unsigned int * pll_compress_site_patterns(char ** sequence,
                                                     const unsigned int * map,
                                                     int count,
                                                     int * length);

/* ---- support (replaces pll.h:650-664) ---- */

PLL_EXPORT int pll_compute_gamma_cats(double alpha,
                                      unsigned int categories,
                                      double * output_rates,
                                      int rates_mode);
PLL_EXPORT void pll_show_pmatrix(const pll_partition_t * partition,
                                 unsigned int index,
                                 unsigned int float_precision);
PLL_EXPORT void pll_show_clv(const pll_partition_t * partition,
                             unsigned int clv_index,
                             int scaler_index,
                             unsigned int float_precision);

/* ---- the array-level entry points (reference: src/pll.h:827-1013, 1659) ----
 *
 * Same names, argument order and meaning as the reference's pll_core_* functions; every
 * operand is a host array.  Each call stages its operands into a per-thread scratch device
 * context, runs the kernels the partition-level calls run, and copies the results back
 * (PCIe-bound by construction: keep data in a partition where speed matters).  Arrays are
 * unpadded (states_padded == states).  The reference pads rows of states under its SIMD flags
 * (pll.c:437-451): a call that carries PLL_ATTRIB_ARCH_SSE / _AVX / _AVX2 with a state count that
 * flag would pad (5 or 7 states under AVX ...) fails with PLL_ERROR_PARAM_INVALID -- the caller's
 * arrays have another stride -- instead of being misread; where the padding is none (4, 8, 20
 * states) the flag is accepted and ignored.  Otherwise of `attrib` only PLL_ATTRIB_RATE_SCALERS
 * is looked at.  Errors: pll_errno / pll_errmsg (PLL_ERROR_HIP_*), -INFINITY from the double
 * functions, PLL_FAILURE from the int functions.
 * The calling thread keeps a few scratch contexts (one per recent shape); pll_amd_core_release
 * frees them -- call it before a thread that used pll_core_* exits. */
PLL_EXPORT void pll_core_create_lookup(unsigned int states, unsigned int rate_cats, double * lookup,
                                       const double * left_matrix, const double * right_matrix,
                                       const unsigned int * tipmap, unsigned int tipmap_size,
                                       unsigned int attrib);                       /* pll.h:829, core_partials.c:725 */
PLL_EXPORT void pll_core_update_partial_tt(unsigned int states, unsigned int sites, unsigned int rate_cats,
                                           double * parent_clv, unsigned int * parent_scaler,
                                           const unsigned char * left_tipchars,
                                           const unsigned char * right_tipchars, const unsigned int * tipmap,
                                           unsigned int tipmap_size, const double * lookup,
                                           unsigned int attrib);                   /* pll.h:838, core_partials.c:82 */
PLL_EXPORT void pll_core_update_partial_ti(unsigned int states, unsigned int sites, unsigned int rate_cats,
                                           double * parent_clv, unsigned int * parent_scaler,
                                           const unsigned char * left_tipchars, const double * right_clv,
                                           const double * left_matrix, const double * right_matrix,
                                           const unsigned int * right_scaler, const unsigned int * tipmap,
                                           unsigned int tipmap_size, unsigned int attrib); /* pll.h:850, core_partials.c:354 */
PLL_EXPORT void pll_core_update_partial_ii(unsigned int states, unsigned int sites, unsigned int rate_cats,
                                           double * parent_clv, unsigned int * parent_scaler,
                                           const double * left_clv, const double * right_clv,
                                           const double * left_matrix, const double * right_matrix,
                                           const unsigned int * left_scaler, const unsigned int * right_scaler,
                                           unsigned int attrib);                   /* pll.h:864, core_partials.c:510 */
PLL_EXPORT void pll_core_create_lookup_4x4(unsigned int rate_cats, double * lookup, const double * left_matrix,
                                           const double * right_matrix);           /* pll.h:877 */
PLL_EXPORT void pll_core_update_partial_tt_4x4(unsigned int sites, unsigned int rate_cats, double * parent_clv,
                                               unsigned int * parent_scaler, const unsigned char * left_tipchars,
                                               const unsigned char * right_tipchars, const double * lookup,
                                               unsigned int attrib);               /* pll.h:882 */
PLL_EXPORT void pll_core_update_partial_ti_4x4(unsigned int sites, unsigned int rate_cats, double * parent_clv,
                                               unsigned int * parent_scaler, const unsigned char * left_tipchars,
                                               const double * right_clv, const double * left_matrix,
                                               const double * right_matrix, const unsigned int * right_scaler,
                                               unsigned int attrib);               /* pll.h:891 */
PLL_EXPORT int pll_core_update_sumtable_ti_4x4(unsigned int sites, unsigned int rate_cats, const double * parent_clv,
                                               const unsigned char * left_tipchars,
                                               const unsigned int * parent_scaler, double * const * eigenvecs,
                                               double * const * inv_eigenvecs, double * const * freqs,
                                               const unsigned int * tipmap, double * sumtable,
                                               unsigned int attrib);               /* pll.h:904 */
PLL_EXPORT int pll_core_update_sumtable_ii(unsigned int states, unsigned int sites, unsigned int rate_cats,
                                           const double * parent_clv, const double * child_clv,
                                           const unsigned int * parent_scaler, const unsigned int * child_scaler,
                                           double * const * eigenvecs, double * const * inv_eigenvecs,
                                           double * const * freqs, double * sumtable,
                                           unsigned int attrib);                   /* pll.h:916, core_derivatives.c:125 */
PLL_EXPORT int pll_core_update_sumtable_ti(unsigned int states, unsigned int sites, unsigned int rate_cats,
                                           const double * parent_clv, const unsigned char * left_tipchars,
                                           const unsigned int * parent_scaler, double * const * eigenvecs,
                                           double * const * inv_eigenvecs, double * const * freqs,
                                           const unsigned int * tipmap, unsigned int tipmap_size,
                                           double * sumtable, unsigned int attrib); /* pll.h:929, core_derivatives.c:277 */
PLL_EXPORT int pll_core_likelihood_derivatives(unsigned int states, unsigned int sites, unsigned int rate_cats,
                                               const double * rate_weights, const unsigned int * parent_scaler,
                                               const unsigned int * child_scaler, const int * invariant,
                                               const unsigned int * pattern_weights, double branch_length,
                                               const double * prop_invar, double * const * freqs,
                                               const double * rates, double * const * eigenvals,
                                               const double * sumtable, double * d_f, double * dd_f,
                                               unsigned int attrib);               /* pll.h:943, core_derivatives.c:501 */
PLL_EXPORT double pll_core_edge_loglikelihood_ii(unsigned int states, unsigned int sites, unsigned int rate_cats,
                                                 const double * parent_clv, const unsigned int * parent_scaler,
                                                 const double * child_clv, const unsigned int * child_scaler,
                                                 const double * pmatrix, double * const * frequencies,
                                                 const double * rate_weights, const unsigned int * pattern_weights,
                                                 const double * invar_proportion, const int * invar_indices,
                                                 const unsigned int * freqs_indices, double * persite_lnl,
                                                 unsigned int attrib);             /* pll.h:963, core_likelihood.c:726 */
PLL_EXPORT double pll_core_edge_loglikelihood_ti(unsigned int states, unsigned int sites, unsigned int rate_cats,
                                                 const double * parent_clv, const unsigned int * parent_scaler,
                                                 const unsigned char * tipchars, const unsigned int * tipmap,
                                                 unsigned int tipmap_size, const double * pmatrix,
                                                 double * const * frequencies, const double * rate_weights,
                                                 const unsigned int * pattern_weights,
                                                 const double * invar_proportion, const int * invar_indices,
                                                 const unsigned int * freqs_indices, double * persite_lnl,
                                                 unsigned int attrib);             /* pll.h:980, core_likelihood.c:412 */
PLL_EXPORT double pll_core_edge_loglikelihood_ti_4x4(unsigned int sites, unsigned int rate_cats,
                                                     const double * parent_clv, const unsigned int * parent_scaler,
                                                     const unsigned char * tipchars, const double * pmatrix,
                                                     double * const * frequencies, const double * rate_weights,
                                                     const unsigned int * pattern_weights,
                                                     const double * invar_proportion, const int * invar_indices,
                                                     const unsigned int * freqs_indices, double * persite_lnl,
                                                     unsigned int attrib);         /* pll.h:998, core_likelihood.c:211 */
PLL_EXPORT double pll_core_root_loglikelihood(unsigned int states, unsigned int sites, unsigned int rate_cats,
                                              const double * clv, const unsigned int * scaler,
                                              double * const * frequencies, const double * rate_weights,
                                              const unsigned int * pattern_weights,
                                              const double * invar_proportion, const int * invar_indices,
                                              const unsigned int * freqs_indices, double * persite_lnl,
                                              unsigned int attrib);                /* pll.h:1013, core_likelihood.c:25 */
PLL_EXPORT int pll_core_update_pmatrix(double ** pmatrix, unsigned int states, unsigned int rate_cats,
                                       const double * rates, const double * branch_lengths,
                                       const unsigned int * matrix_indices, const unsigned int * params_indices,
                                       const double * prop_invar, double * const * eigenvals,
                                       double * const * eigenvecs, double * const * inv_eigenvecs,
                                       unsigned int count, unsigned int attrib);   /* pll.h:1659, core_pmatrix.c:24 */
PLL_EXPORT void pll_amd_core_release(void);

/* ---- additions of this library (no reference counterpart) ---- */

/* Device the NEXT pll_partition_create OF THE CALLING THREAD binds to.  Kept per thread, like pll_errno
 * (pll.c:24-25) -- distinct threads may create partitions on distinct devices concurrently, as the reference lets
 * threads create partitions concurrently -- WITH a process-wide default: a thread that has not set a device uses what
 * the thread that selected a device FIRST in this process set last (a client that selects its device once on the main
 * thread and creates partitions from workers gets that device there; a worker that selects a device of its own moves
 * nobody else's), then env PLL_AMD_DEVICE, else LOCAL_RANK, else 0.  Mixed use -- some threads select, others rely
 * on the default -- is therefore deterministic as long as the thread that owns the default selects before the others
 * create.  pll_amd_get_device() = what a partition created now by the calling thread would get.
 * pll_amd_set_device(-1): the calling thread has no device of its own any more (it follows the default again), and
 * if it was the thread that owns the process-wide default it gives that ownership up -- the next thread that selects
 * a device becomes the owner (the owner is named by its kernel thread id; an owner that exits without this call keeps
 * the default it set, which other threads can then only override for themselves).
 * pll_amd_set_devices() below follows the same rule. */
PLL_EXPORT int pll_amd_set_device(int device);
PLL_EXPORT int pll_amd_get_device(void);
PLL_EXPORT int pll_amd_device_count(void);
/* One partition over several GPUs of this process: the NEXT pll_partition_create splits its
 * sites into contiguous ranges over these devices (default: env PLL_AMD_DEVICES = "0-7",
 * "0,1,2" or "all"; count 0 clears the list; an ordinal may repeat).  Nothing else changes for
 * the client: pll_update_partials runs every range's op list on its device,
 * pll_compute_edge_loglikelihood / pll_compute_likelihood_derivatives return the sum over the
 * ranges (added on the host in range order: deterministic), the pll_amd_sync_* mirrors and
 * persite_lnl are gathered.  PLL_ATTRIB_SITE_REPEATS composes with it (every range identifies its own
 * classes); pll_amd_comm_init does not.  The hot calls (P-matrices, op lists, sumtables, the calls that return
 * a sum) are enqueued on the devices by one host thread per range (round 5; PLLHIP_SHARD_THREADS=0: by the
 * calling thread, range after range) and their results are polled in host-mapped memory.  The
 * list belongs to the calling thread; calls on such a partition leave the caller's current HIP
 * device as they found it. */
PLL_EXPORT int pll_amd_set_devices(const int * devices, unsigned int count);
/* shards of a partition (1 = one device) */
PLL_EXPORT unsigned int pll_amd_shard_count(const pll_partition_t * partition);

/* Mirror mode: 1 = after every mutating call copy the touched CLVs, scale
 * buffers, P-matrices and sumtables back to the host mirrors (lets unmodified
 * reference programs that peek at partition->clv[] work); 0 (default) = the
 * host mirrors are refreshed only by the pll_amd_sync_* calls. */
PLL_EXPORT void pll_amd_set_mirror_mode(int on);

PLL_EXPORT int pll_amd_sync_clv(pll_partition_t * partition, unsigned int clv_index);
PLL_EXPORT int pll_amd_sync_scaler(pll_partition_t * partition, unsigned int scaler_index);
PLL_EXPORT int pll_amd_sync_pmatrix(pll_partition_t * partition, unsigned int matrix_index);
PLL_EXPORT int pll_amd_sync_sumtable(pll_partition_t * partition, double * sumtable);
/* Outside mirror mode pll_update_sumtable does not fill the caller's buffer -- but it does MARK it: the entries of
 * the first site (states * rate_cats doubles) are set to this signalling NaN, so that a client which reads the
 * buffer without pll_amd_sync_sumtable finds NaNs that propagate (and trap under feenableexcept(FE_INVALID)), not
 * yesterday's table or zeros.  pll_amd_sync_sumtable and mirror mode overwrite them with the table. */
#define PLL_AMD_SUMTABLE_POISON 0x7FF453554D544142ull /* sNaN, payload "SUMTAB" */
/* Tell the library a sumtable buffer is about to be freed or refilled by hand: its key is
 * forgotten, so a new buffer at the same address is not mistaken for the old table. */
PLL_EXPORT int pll_amd_forget_sumtable(pll_partition_t * partition, const double * sumtable);
/* block until all work enqueued for this partition has finished */
PLL_EXPORT int pll_amd_wait(pll_partition_t * partition);

/* Site sharding across GPUs (one process per GPU).  After this call every
 * log-likelihood / derivative result of the partition is summed over the
 * `nranks` shards with one RCCL all-reduce on the partition's stream.
 * unique_id: the 128-byte ncclUniqueId obtained from pll_amd_comm_unique_id on
 * rank 0 and distributed by the caller (e.g. torch.distributed broadcast). */
PLL_EXPORT int pll_amd_comm_unique_id(void * unique_id_128bytes);
PLL_EXPORT int pll_amd_comm_init(pll_partition_t * partition, int rank, int nranks,
                                 const void * unique_id_128bytes);
/* Collectives the partition has entered so far (0 without pll_amd_comm_init).  Every rank of a job counts the same
 * number at the same point of the client's program -- also when one rank's scaling certificate (below) makes it run a
 * list again: that happens ahead of the evaluation, never between its collectives. */
PLL_EXPORT unsigned long long pll_amd_comm_reduces(pll_partition_t * partition);
/* Which RCCL the process runs on: the file the collective symbols were bound to and how -- the
 * copy the host program had already mapped (PyTorch ships its own librccl.so) is used if there is
 * one, so that the process never holds two instances; "" before the first pll_amd_comm_* call. */
PLL_EXPORT const char * pll_amd_rccl_path(void);

/* The host eigen solver behind pll_update_eigen, callable without a partition
 * (and without a GPU): eigen system of the reversible rate matrix given by the
 * n(n-1)/2 exchangeabilities and n frequencies.  evecs / inv_evecs are n x n
 * row-major. */
PLL_EXPORT int pll_amd_eigen_decompose(unsigned int n, const double * subst_params,
                                       const double * freqs, double * eigenvals,
                                       double * evecs, double * inv_evecs);

/* HIP-event stopwatch on the partition's stream (bench.py needs the kernel
 * time of THIS stream; torch.cuda.Event only sees torch's own stream). */
PLL_EXPORT int pll_amd_timer_start(pll_partition_t * partition);
PLL_EXPORT int pll_amd_timer_stop_ms(pll_partition_t * partition, float * ms);
/* A partition sharded over several devices (pll_amd_set_devices): what the last stop measured on each
 * device's own stream -- the stopwatch reports the slowest, this shows which one it was.  Returns the
 * number of shards (1 for an ordinary partition) and fills at most `cap` entries. */
PLL_EXPORT unsigned int pll_amd_timer_shard_ms(pll_partition_t * partition, float * ms, unsigned int cap);

/* Per-kernel-class launch timing (HIP events around every hot-kernel launch
 * while enabled).  Arrays have PLL_AMD_PROF_KINDS entries, indexed
 * 0 = CLV update inner-inner, 1 = tip-inner, 2 = tip-tip, 3 = log-likelihood,
 * 4 = sumtable, 5 = derivatives, 6 = P-matrices. */
#define PLL_AMD_PROF_KINDS 7
PLL_EXPORT int pll_amd_profile_enable(pll_partition_t * partition, int on);
PLL_EXPORT int pll_amd_profile_read(pll_partition_t * partition, unsigned int * launches,
                                    double * total_ms);

/* The scaling certificate of 20-state partitions (pllhip.h: pllhip_cert_stats).  The whole-list kernel
 * runs the mat-vec of tip-inner ops on the matrix cores; those CLVs -- and what is computed from them -- agree with
 * the reference's (src/core_partials_avx.c:1229-1284) to ~1e-15 per op instead of bit for bit, and every scaling
 * decision taken on such a value (src/core_partials_avx2.c:752-800) is checked: a largest entry within a window of
 * 2^-256 far wider than the accumulated difference makes the library run the op list again in the reference's
 * order before anything reads its results.  stats[0] = op lists that ran with the check, [1] = flags raised,
 * [2] = lists run again, [3] = uncertified decisions (a partial traversal in the reference's order over CLVs an
 * earlier call left approximate, within 6e-11 of the threshold).  While stats[3] == 0 every scaler count of the
 * partition is the reference's.  PLLHIP_AA_TI_MFMA=0 (environment, read when a partition is created): reference
 * order everywhere, every CLV bit for bit. */
PLL_EXPORT int pll_amd_scaling_certificate(pll_partition_t * partition, unsigned long long * stats4);

/* Measurement only (bench.py's roofline.box_ceiling; pllhip.h: pllhip_write_ceiling): nothing but the stores of an op
 * list with the whole-list kernels' own address pattern, `reps` times, timed with HIP events.  OVERWRITES the CLVs and
 * scale buffers the list's ops write: call pll_update_partials with the list again before reading anything.
 * pll_amd_list_kinds: what the 20-state whole-list kernel made of the last list it planned (pllhip_aa_list_kinds). */
PLL_EXPORT int pll_amd_write_ceiling(pll_partition_t * partition, const pll_operation_t * operations, unsigned int count,
                                     unsigned int reps, float * ms_per_pass, double * bytes_per_pass);
PLL_EXPORT int pll_amd_list_kinds(pll_partition_t * partition, unsigned int * kinds8);
/* Where the partition's CLVs lie (pllhip.h: pllhip_placement_info): a partition of 384 MB or more tries up to
 * PLLHIP_PLACEMENT_TRIES (environment, default 12) places in device memory when it is created and keeps the one it can
 * write fastest.  Returns the number of places tried (0: none -- a small partition, or PLLHIP_PLACEMENT_TRIES=1),
 * the rate in GB/s at which each took the list kernels' store pattern, and which one was kept. */
PLL_EXPORT int pll_amd_placement_info(pll_partition_t * partition, double * gbs, unsigned int cap, int * kept);
/* Measurement only (pllhip.h: pllhip_arena_fill_bandwidth): GB/s of one timed zeroing pass over the partition's CLV
 * arena -- OVERWRITES every CLV (tip CLVs included: upload them again). */
PLL_EXPORT int pll_amd_arena_fill_bandwidth(pll_partition_t * partition, double * gbs);

#ifdef __cplusplus
}
#endif
#endif /* PLL_AMD_H_ */
