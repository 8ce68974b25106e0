// aa_mfma.hpp -- internal: pieces shared by the 20-state matrix-core kernels
// (partials_aa_mfma.hip, likelihood_aa_mfma.hip).  Layouts and the measured
// reasons for the instruction choice are documented in partials_aa_mfma.hip.
#pragma once
#include "ctx.hpp"
#include "numerics.hpp"

typedef double v4d __attribute__((ext_vector_type(4)));
#define PLL_AS1 __attribute__((address_space(1)))
#define PLL_AS3 __attribute__((address_space(3)))

namespace
{
constexpr int S20 = 20;

template <int RC>
struct aa_geom
{
  static constexpr int ROW_G = RC * 10 + 1;            // 16-B granules per site row incl. the pad
  static constexpr int TILE_G = 16 * ROW_G;            // granules per tile image
  static constexpr int N_IT = (TILE_G + 63) / 64;      // wave-instructions to move a tile
  static constexpr int REGION_B = N_IT * 1024;         // bytes reserved per wave
  static constexpr int PTAB = 2 * RC * S20 * S20;      // doubles
  static constexpr size_t LDS_BYTES = (size_t)PTAB * 8 + 4 * (size_t)REGION_B;
};

// Per-lane byte offsets of the N_IT 16-byte granules a lane moves for a tile, relative
// to the tile's first byte in HBM: granule P = it*64 + lane of the padded image is
// (site P / ROW_G, column P % ROW_G); the pad column re-loads the row's last granule.
// Computed once per kernel: 11 VGPRs instead of 64-bit addresses rebuilt per tile.
// (rt / rf: the CLV's rate categories in all and the first one of the RC this kernel instance works on -- rt = RC,
// rf = 0 unless a launch takes a CHUNK of the categories of a partition with 3, 5, 6, 7, 8 ... of them,
// partials_aa_mfma.hip "SPLIT"; compile-time constants wherever the caller's are)
template <int RC>
__device__ __forceinline__ void tile_offsets(unsigned int lane, unsigned int (&off)[aa_geom<RC>::N_IT],
                                             unsigned int rt = RC, unsigned int rf = 0)
{
  using G = aa_geom<RC>;
#pragma unroll
  for (int it = 0; it < G::N_IT; ++it)
  {
    int P = it * 64 + (int)lane;
    if (P > G::TILE_G - 1) P = G::TILE_G - 1;
    const int site = P / G::ROW_G;
    int col = P - site * G::ROW_G;
    if (col > G::ROW_G - 2) col = G::ROW_G - 2;
    off[it] = ((unsigned int)site * (rt * 10u) + rf * 10u + (unsigned int)col) * 16u;
  }
}

// copy the 16-site tile starting at site0 (wave-uniform) of `clv` into the wave's LDS
// image.  Sites past the end of the CLV are read too (every per-site array carries
// PLLHIP_TAIL_SITES of slack); what is computed from them is never stored.
template <int RC, bool NT>
__device__ __forceinline__ void dma_tile(const double * __restrict__ clv, size_t site0,
                                         const unsigned int (&off)[aa_geom<RC>::N_IT], char * region,
                                         unsigned int rt = RC)
{
  using G = aa_geom<RC>;
  // the tile's base address is the same in every lane: say so, it then lives in SGPRs
  const unsigned long long b = (unsigned long long)(clv + site0 * (size_t)(rt * 20u));
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)b);
  const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(b >> 32));
  const char * base = reinterpret_cast<const char *>(((unsigned long long)hi << 32) | lo);
#pragma unroll
  for (int it = 0; it < G::N_IT; ++it)
    __builtin_amdgcn_global_load_lds((const PLL_AS1 void *)(base + off[it]),
                                     (PLL_AS3 void *)(region + it * 1024), 16, 0, NT ? 2 : 0);
}

// Site repeats: the same tile, but site s of it lives in row `rows` (lane s & 15 holds
// the row of site s) of a CLV stored by class.  A row is 160*RC contiguous bytes, so every
// lane still moves one 16-byte granule per instruction; only the address differs.
template <int RC, bool NT>
__device__ __forceinline__ void dma_tile_rows(const double * __restrict__ clv, unsigned int rows,
                                              const unsigned int (&off)[aa_geom<RC>::N_IT], char * region)
{
  using G = aa_geom<RC>;
  constexpr unsigned int ROW_BYTES = RC * 160;
  const char * base = reinterpret_cast<const char *>(clv);
#pragma unroll
  for (int it = 0; it < G::N_IT; ++it)
  {
    const unsigned int site = off[it] / ROW_BYTES, col = off[it] - site * ROW_BYTES;
    const size_t row = (unsigned int)__shfl((int)rows, (int)site, 64);
    __builtin_amdgcn_global_load_lds((const PLL_AS1 void *)(base + row * ROW_BYTES + col),
                                     (PLL_AS3 void *)(region + it * 1024), 16, 0, NT ? 2 : 0);
  }
}

// B operands of the whole tile: b[k][c] = state 4c+q of (site s, rate k)
template <int RC>
__device__ __forceinline__ void read_b_operands(const char * region, unsigned int s, unsigned int q,
                                                double (&b)[RC][5])
{
  constexpr int ROW_B = aa_geom<RC>::ROW_G * 16;
#pragma unroll
  for (int k = 0; k < RC; ++k)
#pragma unroll
    for (int c = 0; c < 5; ++c)
      b[k][c] = *reinterpret_cast<const double *>(region + s * ROW_B + k * 160 + (4 * c + q) * 8);
}

// o[g] = state 4g+q of  P . (column s of the tile) for one child and one rate
// (pk = that rate's 20x20 matrix in LDS, bk = the column's B operands)
__device__ __forceinline__ void rate_matvec(const double * pk, const double (&bk)[5],
                                            unsigned int lane, double (&o)[5])
{
  const unsigned int i = lane & 3u, q = lane >> 4;
#pragma unroll
  for (int g = 0; g < 5; ++g)
  {
    double ag[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) ag[c] = pk[(4 * g + i) * S20 + 4 * c + q];
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < 5; ++c)
      acc = __builtin_amdgcn_mfma_f64_4x4x4f64(ag[c], bk[c], acc, 0, 0, 0);
    o[g] = acc;
  }
}

// x[k][g] for all rates of one child
template <int RC>
__device__ __forceinline__ void tile_matvec(const double * ptab_child, const double (&b)[RC][5],
                                            unsigned int lane, double (&x)[RC][5])
{
#pragma unroll
  for (int k = 0; k < RC; ++k) rate_matvec(ptab_child + (size_t)k * S20 * S20, b[k], lane, x[k]);
}

// ---- the same contraction in the REFERENCE's summation order (core_partials_avx2.c:632-750):
// four accumulators strided by j mod 4, each a chain of fused multiply-adds over j = m, m+4,
// ..., m+16 starting from zero, added as (a0+a1)+(a2+a3).  v_mfma_f64_4x4x4 adds its four
// products to the accumulator as a chain of FMAs in k order, one rounding each (measured on
// 1.28 M random outputs incl. denormals: tools/mfma_order_probe.hip,
// profiles/r3_mfma_f64_accumulation_order.txt), so a contraction chunk made of the states
// {m, m+4, m+8, m+12} IS the first four steps of chain m; the fifth (state 16+m) is a second
// MFMA whose A operand is zero except in k-slot m (adding +0 products changes nothing).  Eight
// MFMAs per four output rows instead of five, three VALU adds per output -- and the result is
// the reference's bit for bit.
//
// B operands for that: lane (s, q) holds states 4q..4q+3 (b[k][0..3]) and 16+q (b[k][4])
template <int RC>
__device__ __forceinline__ void read_b_chain(const char * region, unsigned int s, unsigned int q,
                                             double (&b)[RC][5])
{
  constexpr int ROW_B = aa_geom<RC>::ROW_G * 16;
#pragma unroll
  for (int k = 0; k < RC; ++k)
  {
    const char * p = region + s * ROW_B + k * 160;
    const double2 v0 = *reinterpret_cast<const double2 *>(p + q * 32);
    const double2 v1 = *reinterpret_cast<const double2 *>(p + q * 32 + 16);
    b[k][0] = v0.x; b[k][1] = v0.y; b[k][2] = v1.x; b[k][3] = v1.y;
    b[k][4] = *reinterpret_cast<const double *>(p + 128 + q * 8);
  }
}

// o[g] = state 4g+q of  P . (column s of the tile), reference order.  The matrix is in LDS in its
// natural layout; a14 / a5 are the lane's two addresses into matrix 0 -- row (lane & 3), columns
// 4q..4q+3 / column 16+q -- made ONCE per kernel (chain_lane_bases): every operand is then one
// of the two registers plus a constant.  (Spelled with indices the compiler kept a base
// register per (rate, row group, child) alive across the tile loop and spilled 100 of them.)
__device__ __forceinline__ void chain_lane_bases(const double * ptab, unsigned int lane, const char *& a14,
                                                 const char *& a5)
{
  const unsigned int i = lane & 3u, q = lane >> 4;
  a14 = reinterpret_cast<const char *>(ptab) + i * 160u + q * 32u;
  a5 = reinterpret_cast<const char *>(ptab) + i * 160u + 128u + q * 8u;
}

template <int MAT_BYTE_OFFSET>
__device__ __forceinline__ void rate_matvec_chain(const char * a14, const char * a5, const double (&bk)[5],
                                                  unsigned int lane, double (&o)[5])
{
  const unsigned int q = lane >> 4;
#pragma unroll
  for (int g = 0; g < 5; ++g)
  {
    const double2 a01 = *reinterpret_cast<const double2 *>(a14 + MAT_BYTE_OFFSET + g * 640);
    const double2 a23 = *reinterpret_cast<const double2 *>(a14 + MAT_BYTE_OFFSET + g * 640 + 16);
    const double a4 = *reinterpret_cast<const double *>(a5 + MAT_BYTE_OFFSET + g * 640);
    const double a1[4] = {a01.x, a01.y, a23.x, a23.y};
    double acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1[m], bk[m], 0.0, 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 4; ++m)
      acc[m] = __builtin_amdgcn_mfma_f64_4x4x4f64(q == (unsigned int)m ? a4 : 0.0, bk[4], acc[m], 0, 0, 0);
    o[g] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    // The sum is wanted HERE: without a use at this point the optimiser sinks the three adds to
    // where o[g] is consumed -- after the other child's products -- and keeps all four
    // accumulators of every output alive until then (the kernel then spills 300-500 bytes per
    // lane and takes 3.3 x the time).  And nothing moves across: left alone the scheduler
    // fetches the A operands of several row groups ahead.
    asm volatile("" : "+v"(o[g]));
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- the same contraction in the order of the reference's TIP-INNER kernel, which is what a tip-inner op reaches
// under the AVX2 flag too (core_partials.c:427-441 dispatches it to pll_core_update_partial_ti_avx, whose 20-state
// kernel is core_partials_avx.c:1097-1340): the four accumulators strided by j mod 4 and the pairwise tree as above,
// but every step is a multiplication and THEN an addition (:1239-1262) -- two roundings.  No matrix-core instruction
// does that (its products are never rounded on their own), so this mat-vec runs on the vector unit:
//   * the lane needs all 20 entries of its column (site s, this rate) and holds five of them as B operands
//     (states 4q..4q+3 and 16+q); the other fifteen come from the three lanes s + 16 q' through the LDS crossbar
//     (ds_bpermute: no LDS memory involved, the tile image may be refilled meanwhile);
//   * row 4g+q of the rate's matrix is 160 contiguous bytes of LDS: ten 16-byte reads, the sixteen lanes of a q
//     read the same address (broadcast), the four q's hit different banks;
//   * 20 multiplications + 23 additions per output, 215 per lane and rate: 3440 issue cycles per wave and 16-site
//     tile, about a third of the time HBM needs for the tile's 20.7 KB at two waves per SIMD.
// prow: the lane's row q of matrix 0 of the child (bytes).  Result: the reference's bits.
__device__ __forceinline__ double lane_fetch_f64(double v, unsigned int src_lane_x4)
{
  const int lo = __builtin_amdgcn_ds_bpermute((int)src_lane_x4, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute((int)src_lane_x4, __double2hiint(v));
  return __hiloint2double(hi, lo);
}

template <int MAT_BYTE_OFFSET>
__device__ __forceinline__ void rate_matvec_plain(const char * prow, const double (&bk)[5], unsigned int lane,
                                                  double (&o)[5])
{
  const unsigned int s = lane & 15u;
  double c[20];
#pragma unroll
  for (int qq = 0; qq < 4; ++qq)
  {
    const unsigned int src = (s + 16u * qq) * 4u;
#pragma unroll
    for (int t = 0; t < 4; ++t) c[4 * qq + t] = lane_fetch_f64(bk[t], src);
    c[16 + qq] = lane_fetch_f64(bk[4], src);
  }
#pragma unroll
  for (int g = 0; g < 5; ++g)
  {
    // (chains 0 and 1, then chains 2 and 3: five 16-byte reads in flight instead of ten)
    double a[4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
    {
      double2 row[5];
#pragma unroll
      for (int j = 0; j < 5; ++j) row[j] = *reinterpret_cast<const double2 *>(prow + MAT_BYTE_OFFSET + g * 640 + j * 32 + h * 16);
      double e0 = 0.0, e1 = 0.0;
#pragma unroll
      for (int j = 0; j < 5; ++j)
      {
        e0 = e0 + row[j].x * c[4 * j + 2 * h];
        e1 = e1 + row[j].y * c[4 * j + 2 * h + 1];
      }
      a[2 * h] = e0;
      a[2 * h + 1] = e1;
      asm volatile("" : "+v"(a[2 * h]), "+v"(a[2 * h + 1]));
      asm volatile("" ::: "memory");
    }
    const double a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
    o[g] = (a0 + a1) + (a2 + a3);
    // (the sum is wanted here and one row is in flight at a time: see rate_matvec_chain)
    asm volatile("" : "+v"(o[g]));
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
}

// x[k][g] for all rates of one child whose matrices start CHILD_BYTE_OFFSET bytes behind matrix 0
template <int RC, int CHILD_BYTE_OFFSET>
__device__ __forceinline__ void tile_matvec_chain(const char * a14, const char * a5, const double (&b)[RC][5],
                                                  unsigned int lane, double (&x)[RC][5])
{
  if (RC >= 1) rate_matvec_chain<CHILD_BYTE_OFFSET>(a14, a5, b[0], lane, x[0]);
  if (RC >= 2) rate_matvec_chain<CHILD_BYTE_OFFSET + 3200>(a14, a5, b[RC >= 2 ? 1 : 0], lane, x[RC >= 2 ? 1 : 0]);
  if (RC >= 4)
  {
    rate_matvec_chain<CHILD_BYTE_OFFSET + 6400>(a14, a5, b[RC >= 4 ? 2 : 0], lane, x[RC >= 4 ? 2 : 0]);
    rate_matvec_chain<CHILD_BYTE_OFFSET + 9600>(a14, a5, b[RC >= 4 ? 3 : 0], lane, x[RC >= 4 ? 3 : 0]);
  }
}

// all 4 lanes of this lane's tile column (s, s+16, s+32, s+48) have the flag set
__device__ __forceinline__ bool column_all(bool f, unsigned int s)
{
  const unsigned long long b = __ballot(f);
  const unsigned long long m = 0x0001000100010001ull << s;
  return (b & m) == m;
}
} // namespace

