// partials_aa_fused.hip -- a whole op list of 20-state CLV updates in ONE kernel, site-blocked.
//
// Replaces, for pll_update_partials (partials.c:177-213) on 20-state data with 4 rate
// categories, the one-launch-per-tree-level path of partials.hip / partials_aa_mfma.hip, in
// which every child CLV makes a round trip through HBM (1932 B per inner-inner site-update:
// core_partials_avx2.c:568-803 reads two children and writes the parent).  As in the 4-state
// whole-list kernel (partials_fused.hip) a tile of sites is taken through the WHOLE list and
// what an op produced stays on chip for the op that consumes it: the list becomes a write
// stream of 644 B per site-update.
//
//   tile      a wave owns 8 sites = 2 sub-tiles of (4 sites x 4 rates) = the 16 columns of one
//             v_mfma_f64_4x4x4_4b: block b of the instruction is rate b, so ONE A operand
//             serves all four rates (each block has its own A) and an A operand fetched from
//             LDS is used for both sub-tiles.  (With the per-level kernel's tile of 16 sites x
//             one rate a value kept on chip would cost 10 KB per slot.)
//   slots     values live in REGISTERS, not LDS: the accumulator layout of the MFMA (lane =
//             (q, rate, site), states 4q..4q+3 and 16+q) is also its B-operand layout, so a
//             parent is consumed as it was produced -- 20 VGPRs per value and tile, five
//             values per wave (a balanced 64-taxon tree needs four, 128 taxa five); the slot
//             number of an op is wave-uniform and selects by a scalar branch.
//   matrices  the two 12.8 KB P-matrix blocks of an op are staged in LDS once per WORKGROUP
//             (four waves in lock-step: two barriers per op) by LDS-DMA, double-buffered by
//             halves -- the left block of op i+1 arrives while op i multiplies by its right
//             block -- in "operand order": the 64 lanes of one MFMA read 512 contiguous bytes
//             (k_af_prepare permutes every matrix of the list once per call).
//   order     the reference's (core_partials_avx2.c:632-750): four FMA chains strided by
//             j mod 4, then (a0+a1)+(a2+a3); the MFMA adds its four products as a chain of
//             FMAs in k order (tools/mfma_order_probe.hip), so a k-chunk {m, m+4, m+8, m+12}
//             is four steps of chain m -- CLVs and scaler counts are the reference's bit for bit
//             (inner-inner ops; the reference's tip-inner kernel does not fuse, core_partials_avx.c:1097).
//   tip-tip   ops run ahead of the list as before (k_aa_tt_rounds: a pure write stream at the
//             write ceiling already); an inner-inner op over two of them -- or a tip-inner op over
//             one -- is a table lookup over character pairs (partials_aa_mfma.hip), done here per
//             tile in the 16-bytes-per-lane layout of the stores and handed on through LDS.
//   reload    an operand that has no slot (written by an earlier call, or by a tip-tip op whose
//             reader is not a lookup, or evicted) is copied from HBM by LDS-DMA during the op
//             before its reader.
//   stores    a finished tile goes through a per-wave LDS stage into the 16-bytes-per-lane
//             layout: 5 KB contiguous per wave and op.
//
// The order of the list and the slots come from the planner of partials_fused.hip (depth
// first, heavier subtree first; Belady for evictions).
//
// Roofline: HBM writes, 644 B per site-update (640 B CLV + 4 B scaler count) + tip
// characters; the matrix cores run 80 MFMAs per 16 columns and child (16 cycles each):
// about 0.6 of the time the stores need.  DESIGN.md 2.2c has the budget and the measurements.
#include <algorithm>
#include <stdlib.h>
#include <type_traits>
#include <vector>

#include "ctx.hpp"
#include "numerics.hpp"
#include "partials_fused.hpp"

#define PLL_LDS __attribute__((address_space(3)))
#define PLL_GLOBAL __attribute__((address_space(1)))

namespace
{
constexpr int AF_J = 2;                    // sub-tiles per wave
constexpr int AF_WS = 4 * AF_J;            // sites of a wave's tile
constexpr int AF_WGS = 4 * AF_WS;          // sites of a workgroup's tile (four waves)
constexpr int AF_TILE_B = AF_WS * 640;     // bytes of a wave's tile of a CLV
constexpr int AF_NSLOT = 5;                // values a wave keeps in registers
constexpr int AF_MAT_PIECES = 13;          // 1 KB pieces of a matrix block in operand order (12.5, padded)
constexpr int AF_MAT_B = AF_MAT_PIECES * 1024;
constexpr int AF_LDS_B = 2 * AF_MAT_B + 4 * 2 * AF_TILE_B + 16;

// One record per op, read through the scalar cache: everything the wave needs while the op runs.
struct AaRec
{
  unsigned long long parent;      // CLV the op writes
  unsigned long long pscaler;     // its scale buffer (0: none)
  unsigned long long tab_l;       // lookup: table of pair 1; tip-inner: the tip's table [code][rate][state]
  unsigned long long tab_r;       // lookup: table of pair 2
  unsigned long long row[4];      // tip rows: lookup (t1, t2), (t3, t4); tip-inner row[0]; else rows of zeros
  unsigned int xoff, yoff;        // byte offsets of the op's left / right matrix blocks (operand order)
  unsigned int flags;             // AF_* below
  unsigned int slots;             // lslot | rslot << 4 | pslot << 8 | ra_slot << 12 | rb_slot << 16
  unsigned long long ra_src, ra_cnt; // reload A: done during this op for the NEXT op's left operand
  unsigned long long rb_src, rb_cnt; // reload B: the next op's right operand
  unsigned int pad[4];
};
static_assert(sizeof(AaRec) == 128, "thirty-two words per op");
constexpr unsigned int AF_KIND_MASK = 3u;   // 0 inner-inner, 1 tip-inner, 2 lookup
constexpr unsigned int AF_HAS_PSLOT = 4u;
constexpr unsigned int AF_SCALING = 8u;
constexpr unsigned int AF_LCNT = 16u;
constexpr unsigned int AF_RCNT = 32u;
constexpr unsigned int AF_RELOAD_A = 64u;
constexpr unsigned int AF_RELOAD_B = 128u;

struct AfMatJob
{
  const double * src;           // [rate][row][column]
  unsigned long long dst_off;   // bytes into the operand-order buffer
};
struct AfTipJob
{
  const double * lmat;          // the tip's matrices
  unsigned long long dst_off;   // bytes into the tip-table buffer
};

// ---- per list: every matrix block of the list in operand order, every tip table
// Operand order: 25 blocks of 64 doubles, block (t, b), t = row group, b = 0..3 the first MFMA of
// chain b, b = 4 the source of the second; lane (q, rate, i) of block (t, b) holds
// P_rate[G_t(i)][4q + b] (b < 4) or P_rate[G_t(i)][16 + q], G_t(i) = 4i + t (t < 4), 16 + i:
// the accumulator of group t then puts state 4q + t (or 16 + q) in lane q.
__global__ __launch_bounds__(256) void k_af_prepare(const AfMatJob * __restrict__ mj, unsigned int nmat,
                                                    const AfTipJob * __restrict__ tj, unsigned int ntip,
                                                    char * aorder, char * titab,
                                                    const unsigned int * __restrict__ tipmap, unsigned int ms)
{
  const unsigned int b = blockIdx.x;
  if (b < nmat)
  {
    const double * src = mj[b].src;
    double * out = reinterpret_cast<double *>(aorder + mj[b].dst_off);
    for (unsigned int idx = threadIdx.x; idx < AF_MAT_B / 8; idx += blockDim.x)
    {
      double v = 0.0;
      if (idx < 1600)
      {
        const unsigned int blk = idx >> 6, lane = idx & 63u, t = blk / 5, bb = blk - 5 * t;
        const unsigned int q = lane >> 4, rate = (lane >> 2) & 3u, i = lane & 3u;
        const unsigned int row = t < 4 ? 4 * i + t : 16 + i, col = bb < 4 ? 4 * q + bb : 16 + q;
        v = src[rate * 400 + row * 20 + col];
      }
      out[idx] = v;
    }
  }
  else if (b - nmat < ntip)
  {
    // tab[code][k][i] = sum_{j in tipmap[code]} P[k][i][j] (core_partials_avx.c:1140-1177), as k_aa_tip_tables
    const double * lmat = tj[b - nmat].lmat;
    double * out = reinterpret_cast<double *>(titab + tj[b - nmat].dst_off);
    for (unsigned int t = threadIdx.x; t < ms * 80; t += blockDim.x)
    {
      const unsigned int code = t / 80, ki = t - 80 * code;
      out[t] = masksum_seq(lmat + (size_t)ki * 20, tipmap[code], 20);
    }
  }
}

// ---- device helpers
typedef const unsigned int __attribute__((address_space(4))) * af_words;
// words [FIRST, FIRST + N) of record i, through the scalar cache
template <int FIRST, int N>
struct AfW
{
  unsigned int w[N];
  __device__ __forceinline__ unsigned int operator[](int t) const { return w[t - FIRST]; }
  __device__ __forceinline__ unsigned long long quad(int t) const
  {
    return (unsigned long long)w[t - FIRST] | ((unsigned long long)w[t - FIRST + 1] << 32);
  }
};
template <int FIRST, int N>
__device__ __forceinline__ AfW<FIRST, N> af_load(const AaRec * plan, unsigned int i)
{
  const af_words p = (af_words)(unsigned long long)(plan + i) + FIRST;
  AfW<FIRST, N> r;
#pragma unroll
  for (int t = 0; t < N; ++t) r.w[t] = p[t];
  return r;
}

// Every global access of the kernel is "wave-uniform 64-bit base (SGPRs) + 32-bit lane offset": one
// VGPR per address instead of two, and nothing per-lane to keep (or spill) across the op loop.
typedef char PLL_GLOBAL * af_gptr;
__device__ __forceinline__ af_gptr af_base(unsigned long long uniform_address)
{
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)uniform_address);
  const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(uniform_address >> 32));
  return (af_gptr)(((unsigned long long)hi << 32) | lo);
}

// 64 lanes x 16 bytes from global memory straight into LDS (lane l lands at lds_b + 16 l).  Inline
// assembly on purpose: the compiler neither counts it nor waits for it; the waits are ours.
__device__ __forceinline__ void af_dma16(unsigned int lds_b, unsigned long long uniform_src, unsigned int lane16)
{
  unsigned int m0_saved;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(m0_saved)
               : "s"(lds_b), "v"(lane16), "s"(uniform_src)
               : "memory");
}

struct AfSlot
{
  double v[AF_J][5];
  unsigned int c[AF_J];
};

// x[j][t] = (MUL ? x[j][t] : 1) * (P . column)[state 4q + t | 16 + q], reference order;
// mat_lane: the block in LDS (operand order) + 8 lane
template <bool MUL>
__device__ __forceinline__ void af_matvec(const char * mat_lane, unsigned int q, const double (&b)[AF_J][5],
                                          double (&x)[AF_J][5])
{
#pragma unroll
  for (int t = 0; t < 5; ++t)
  {
    double a1[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) a1[m] = *reinterpret_cast<const double *>(mat_lane + (t * 5 + m) * 512);
    const double a4 = *reinterpret_cast<const double *>(mat_lane + (t * 5 + 4) * 512);
    double acc[AF_J][4];
#pragma unroll
    for (int j = 0; j < AF_J; ++j)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[j][m] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1[m], b[j][m], 0.0, 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 4; ++m)
    {
      const double am = (q == (unsigned int)m) ? a4 : 0.0;
#pragma unroll
      for (int j = 0; j < AF_J; ++j) acc[j][m] = __builtin_amdgcn_mfma_f64_4x4x4f64(am, b[j][4], acc[j][m], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < AF_J; ++j)
    {
      const double sum = (acc[j][0] + acc[j][1]) + (acc[j][2] + acc[j][3]);
      x[j][t] = MUL ? x[j][t] * sum : sum;
      // (the value is wanted here, see rate_matvec_chain in aa_mfma.hpp)
      asm volatile("" : "+v"(x[j][t]));
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// a tile in the lane layout of the MFMA operands, out of a stage in LDS
__device__ __forceinline__ void af_read_tile(const char * stage_lane, const char * stage_lane5, double (&v)[AF_J][5])
{
#pragma unroll
  for (int j = 0; j < AF_J; ++j)
  {
    const double2 v0 = *reinterpret_cast<const double2 *>(stage_lane + j * 2560);
    const double2 v1 = *reinterpret_cast<const double2 *>(stage_lane + j * 2560 + 16);
    v[j][0] = v0.x; v[j][1] = v0.y; v[j][2] = v1.x; v[j][3] = v1.y;
    v[j][4] = *reinterpret_cast<const double *>(stage_lane5 + j * 2560);
  }
}

// The five slots are sixty scalar variables that never move.  Three things it took:
//  - a slot is read and written by INLINE ASSEMBLY that compares the (wave-uniform) slot number and
//    branches over a dozen moves.  Written as a C++ switch the optimiser first merged the arms into
//    one access through a pointer that depends on the slot number (the slots became 448 bytes of
//    scratch per lane); with scalars instead of structs it kept them in registers but threaded all
//    sixty values through every arm of every switch as phi nodes, and the register allocator answered
//    with 1150 moves and a slot in scratch.  Behind the asm there is no control flow to see.
//  - every asm names the PHYSICAL registers of the slot it touches (v146..v255): left to itself the
//    allocator gave the same slot different registers in different parts of the op loop and copied
//    all fifty pairs from one set to the other on the way (45 v_mov_b64 in a row).
//  - no lambda may touch them: a reference capture takes their address, and they would be memory.
#define AF_SLOT_VARS(N)                                                                                           \
  double N##_00 = 0.0, N##_01 = 0.0, N##_02 = 0.0, N##_03 = 0.0, N##_04 = 0.0, N##_10 = 0.0, N##_11 = 0.0,        \
         N##_12 = 0.0, N##_13 = 0.0, N##_14 = 0.0;                                                                \
  unsigned int N##_c0 = 0u, N##_c1 = 0u;
#define AF_MOVE6_ASM(K)                                                                                           \
  "s_cmp_lg_u32 %12, " #K "\n\ts_cbranch_scc1 .Laf_skip_%=\n\t"                                                   \
  "v_mov_b64 %0, %6\n\tv_mov_b64 %1, %7\n\tv_mov_b64 %2, %8\n\tv_mov_b64 %3, %9\n\t"                              \
  "v_mov_b64 %4, %10\n\tv_mov_b32 %5, %11\n.Laf_skip_%=:"

#define AF_PUT_0_0(idx, L) asm volatile(AF_MOVE6_ASM(0) : "+{v[146:147]}"(s0_00), "+{v[148:149]}"(s0_01), "+{v[150:151]}"(s0_02), "+{v[152:153]}"(s0_03), "+{v[154:155]}"(s0_04), "+{v166}"(s0_c0) : "v"(L.v[0][0]), "v"(L.v[0][1]), "v"(L.v[0][2]), "v"(L.v[0][3]), "v"(L.v[0][4]), "v"(L.c[0]), "s"(idx) : "scc");
#define AF_GET_0_0(idx, L) asm volatile(AF_MOVE6_ASM(0) : "+v"(L.v[0][0]), "+v"(L.v[0][1]), "+v"(L.v[0][2]), "+v"(L.v[0][3]), "+v"(L.v[0][4]), "+v"(L.c[0]) : "{v[146:147]}"(s0_00), "{v[148:149]}"(s0_01), "{v[150:151]}"(s0_02), "{v[152:153]}"(s0_03), "{v[154:155]}"(s0_04), "{v166}"(s0_c0), "s"(idx) : "scc");
#define AF_PUT_0_1(idx, L) asm volatile(AF_MOVE6_ASM(0) : "+{v[156:157]}"(s0_10), "+{v[158:159]}"(s0_11), "+{v[160:161]}"(s0_12), "+{v[162:163]}"(s0_13), "+{v[164:165]}"(s0_14), "+{v167}"(s0_c1) : "v"(L.v[1][0]), "v"(L.v[1][1]), "v"(L.v[1][2]), "v"(L.v[1][3]), "v"(L.v[1][4]), "v"(L.c[1]), "s"(idx) : "scc");
#define AF_GET_0_1(idx, L) asm volatile(AF_MOVE6_ASM(0) : "+v"(L.v[1][0]), "+v"(L.v[1][1]), "+v"(L.v[1][2]), "+v"(L.v[1][3]), "+v"(L.v[1][4]), "+v"(L.c[1]) : "{v[156:157]}"(s0_10), "{v[158:159]}"(s0_11), "{v[160:161]}"(s0_12), "{v[162:163]}"(s0_13), "{v[164:165]}"(s0_14), "{v167}"(s0_c1), "s"(idx) : "scc");
#define AF_PUT_1_0(idx, L) asm volatile(AF_MOVE6_ASM(1) : "+{v[168:169]}"(s1_00), "+{v[170:171]}"(s1_01), "+{v[172:173]}"(s1_02), "+{v[174:175]}"(s1_03), "+{v[176:177]}"(s1_04), "+{v188}"(s1_c0) : "v"(L.v[0][0]), "v"(L.v[0][1]), "v"(L.v[0][2]), "v"(L.v[0][3]), "v"(L.v[0][4]), "v"(L.c[0]), "s"(idx) : "scc");
#define AF_GET_1_0(idx, L) asm volatile(AF_MOVE6_ASM(1) : "+v"(L.v[0][0]), "+v"(L.v[0][1]), "+v"(L.v[0][2]), "+v"(L.v[0][3]), "+v"(L.v[0][4]), "+v"(L.c[0]) : "{v[168:169]}"(s1_00), "{v[170:171]}"(s1_01), "{v[172:173]}"(s1_02), "{v[174:175]}"(s1_03), "{v[176:177]}"(s1_04), "{v188}"(s1_c0), "s"(idx) : "scc");
#define AF_PUT_1_1(idx, L) asm volatile(AF_MOVE6_ASM(1) : "+{v[178:179]}"(s1_10), "+{v[180:181]}"(s1_11), "+{v[182:183]}"(s1_12), "+{v[184:185]}"(s1_13), "+{v[186:187]}"(s1_14), "+{v189}"(s1_c1) : "v"(L.v[1][0]), "v"(L.v[1][1]), "v"(L.v[1][2]), "v"(L.v[1][3]), "v"(L.v[1][4]), "v"(L.c[1]), "s"(idx) : "scc");
#define AF_GET_1_1(idx, L) asm volatile(AF_MOVE6_ASM(1) : "+v"(L.v[1][0]), "+v"(L.v[1][1]), "+v"(L.v[1][2]), "+v"(L.v[1][3]), "+v"(L.v[1][4]), "+v"(L.c[1]) : "{v[178:179]}"(s1_10), "{v[180:181]}"(s1_11), "{v[182:183]}"(s1_12), "{v[184:185]}"(s1_13), "{v[186:187]}"(s1_14), "{v189}"(s1_c1), "s"(idx) : "scc");
#define AF_PUT_2_0(idx, L) asm volatile(AF_MOVE6_ASM(2) : "+{v[190:191]}"(s2_00), "+{v[192:193]}"(s2_01), "+{v[194:195]}"(s2_02), "+{v[196:197]}"(s2_03), "+{v[198:199]}"(s2_04), "+{v210}"(s2_c0) : "v"(L.v[0][0]), "v"(L.v[0][1]), "v"(L.v[0][2]), "v"(L.v[0][3]), "v"(L.v[0][4]), "v"(L.c[0]), "s"(idx) : "scc");
#define AF_GET_2_0(idx, L) asm volatile(AF_MOVE6_ASM(2) : "+v"(L.v[0][0]), "+v"(L.v[0][1]), "+v"(L.v[0][2]), "+v"(L.v[0][3]), "+v"(L.v[0][4]), "+v"(L.c[0]) : "{v[190:191]}"(s2_00), "{v[192:193]}"(s2_01), "{v[194:195]}"(s2_02), "{v[196:197]}"(s2_03), "{v[198:199]}"(s2_04), "{v210}"(s2_c0), "s"(idx) : "scc");
#define AF_PUT_2_1(idx, L) asm volatile(AF_MOVE6_ASM(2) : "+{v[200:201]}"(s2_10), "+{v[202:203]}"(s2_11), "+{v[204:205]}"(s2_12), "+{v[206:207]}"(s2_13), "+{v[208:209]}"(s2_14), "+{v211}"(s2_c1) : "v"(L.v[1][0]), "v"(L.v[1][1]), "v"(L.v[1][2]), "v"(L.v[1][3]), "v"(L.v[1][4]), "v"(L.c[1]), "s"(idx) : "scc");
#define AF_GET_2_1(idx, L) asm volatile(AF_MOVE6_ASM(2) : "+v"(L.v[1][0]), "+v"(L.v[1][1]), "+v"(L.v[1][2]), "+v"(L.v[1][3]), "+v"(L.v[1][4]), "+v"(L.c[1]) : "{v[200:201]}"(s2_10), "{v[202:203]}"(s2_11), "{v[204:205]}"(s2_12), "{v[206:207]}"(s2_13), "{v[208:209]}"(s2_14), "{v211}"(s2_c1), "s"(idx) : "scc");
#define AF_PUT_3_0(idx, L) asm volatile(AF_MOVE6_ASM(3) : "+{v[212:213]}"(s3_00), "+{v[214:215]}"(s3_01), "+{v[216:217]}"(s3_02), "+{v[218:219]}"(s3_03), "+{v[220:221]}"(s3_04), "+{v232}"(s3_c0) : "v"(L.v[0][0]), "v"(L.v[0][1]), "v"(L.v[0][2]), "v"(L.v[0][3]), "v"(L.v[0][4]), "v"(L.c[0]), "s"(idx) : "scc");
#define AF_GET_3_0(idx, L) asm volatile(AF_MOVE6_ASM(3) : "+v"(L.v[0][0]), "+v"(L.v[0][1]), "+v"(L.v[0][2]), "+v"(L.v[0][3]), "+v"(L.v[0][4]), "+v"(L.c[0]) : "{v[212:213]}"(s3_00), "{v[214:215]}"(s3_01), "{v[216:217]}"(s3_02), "{v[218:219]}"(s3_03), "{v[220:221]}"(s3_04), "{v232}"(s3_c0), "s"(idx) : "scc");
#define AF_PUT_3_1(idx, L) asm volatile(AF_MOVE6_ASM(3) : "+{v[222:223]}"(s3_10), "+{v[224:225]}"(s3_11), "+{v[226:227]}"(s3_12), "+{v[228:229]}"(s3_13), "+{v[230:231]}"(s3_14), "+{v233}"(s3_c1) : "v"(L.v[1][0]), "v"(L.v[1][1]), "v"(L.v[1][2]), "v"(L.v[1][3]), "v"(L.v[1][4]), "v"(L.c[1]), "s"(idx) : "scc");
#define AF_GET_3_1(idx, L) asm volatile(AF_MOVE6_ASM(3) : "+v"(L.v[1][0]), "+v"(L.v[1][1]), "+v"(L.v[1][2]), "+v"(L.v[1][3]), "+v"(L.v[1][4]), "+v"(L.c[1]) : "{v[222:223]}"(s3_10), "{v[224:225]}"(s3_11), "{v[226:227]}"(s3_12), "{v[228:229]}"(s3_13), "{v[230:231]}"(s3_14), "{v233}"(s3_c1), "s"(idx) : "scc");
#define AF_PUT_4_0(idx, L) asm volatile(AF_MOVE6_ASM(4) : "+{v[234:235]}"(s4_00), "+{v[236:237]}"(s4_01), "+{v[238:239]}"(s4_02), "+{v[240:241]}"(s4_03), "+{v[242:243]}"(s4_04), "+{v254}"(s4_c0) : "v"(L.v[0][0]), "v"(L.v[0][1]), "v"(L.v[0][2]), "v"(L.v[0][3]), "v"(L.v[0][4]), "v"(L.c[0]), "s"(idx) : "scc");
#define AF_GET_4_0(idx, L) asm volatile(AF_MOVE6_ASM(4) : "+v"(L.v[0][0]), "+v"(L.v[0][1]), "+v"(L.v[0][2]), "+v"(L.v[0][3]), "+v"(L.v[0][4]), "+v"(L.c[0]) : "{v[234:235]}"(s4_00), "{v[236:237]}"(s4_01), "{v[238:239]}"(s4_02), "{v[240:241]}"(s4_03), "{v[242:243]}"(s4_04), "{v254}"(s4_c0), "s"(idx) : "scc");
#define AF_PUT_4_1(idx, L) asm volatile(AF_MOVE6_ASM(4) : "+{v[244:245]}"(s4_10), "+{v[246:247]}"(s4_11), "+{v[248:249]}"(s4_12), "+{v[250:251]}"(s4_13), "+{v[252:253]}"(s4_14), "+{v255}"(s4_c1) : "v"(L.v[1][0]), "v"(L.v[1][1]), "v"(L.v[1][2]), "v"(L.v[1][3]), "v"(L.v[1][4]), "v"(L.c[1]), "s"(idx) : "scc");
#define AF_GET_4_1(idx, L) asm volatile(AF_MOVE6_ASM(4) : "+v"(L.v[1][0]), "+v"(L.v[1][1]), "+v"(L.v[1][2]), "+v"(L.v[1][3]), "+v"(L.v[1][4]), "+v"(L.c[1]) : "{v[244:245]}"(s4_10), "{v[246:247]}"(s4_11), "{v[248:249]}"(s4_12), "{v[250:251]}"(s4_13), "{v[252:253]}"(s4_14), "{v255}"(s4_c1), "s"(idx) : "scc");
#define AF_SLOT_READ(idx_expr, L)                                                                                 \
  {                                                                                                               \
    const unsigned int idx_ = __builtin_amdgcn_readfirstlane(idx_expr);                                           \
    L.v[0][0] = s0_00; L.v[0][1] = s0_01; L.v[0][2] = s0_02; L.v[0][3] = s0_03; L.v[0][4] = s0_04;                \
    L.v[1][0] = s0_10; L.v[1][1] = s0_11; L.v[1][2] = s0_12; L.v[1][3] = s0_13; L.v[1][4] = s0_14;                \
    L.c[0] = s0_c0; L.c[1] = s0_c1;                                                                               \
    AF_GET_1_0(idx_, L) AF_GET_1_1(idx_, L) AF_GET_2_0(idx_, L) AF_GET_2_1(idx_, L)                               \
    AF_GET_3_0(idx_, L) AF_GET_3_1(idx_, L) AF_GET_4_0(idx_, L) AF_GET_4_1(idx_, L)                               \
  }
#define AF_SLOT_WRITE(idx_expr, L)                                                                                \
  {                                                                                                               \
    const unsigned int idx_ = __builtin_amdgcn_readfirstlane(idx_expr);                                           \
    AF_PUT_0_0(idx_, L) AF_PUT_0_1(idx_, L) AF_PUT_1_0(idx_, L) AF_PUT_1_1(idx_, L) AF_PUT_2_0(idx_, L)           \
    AF_PUT_2_1(idx_, L) AF_PUT_3_0(idx_, L) AF_PUT_3_1(idx_, L) AF_PUT_4_0(idx_, L) AF_PUT_4_1(idx_, L)           \
  }
static_assert(AF_J == 2 && AF_NSLOT == 5, "the slot macros spell out two sub-tiles and five slots");

// MODE: SCALE_NONE (no op of the list has a scale buffer) or SCALE_SITE
template <int MODE, bool NT>
__global__ __launch_bounds__(256, 2) void k_aa_fused(const AaRec * __restrict__ plan, unsigned int nops, unsigned int sites,
                                                     const char * aorder, unsigned int ms, double2 * sink,
                                                     unsigned int * next_tile, unsigned int static_rounds)
{
  extern __shared__ double2 lds_af[];
  const unsigned int lane = threadIdx.x & 63u;
  const unsigned int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned int q = lane >> 4, rate = (lane >> 2) & 3u, n = lane & 3u;
  // LDS: [left block][right block][per wave: out stage, in stage][tile word]
  char * lds = reinterpret_cast<char *>(lds_af);
  const unsigned int lds_b = __builtin_amdgcn_readfirstlane((unsigned int)(uintptr_t)(PLL_LDS char *)lds_af);
  const unsigned int xbuf_b = lds_b, ybuf_b = lds_b + AF_MAT_B;
  const unsigned int in_b = lds_b + 2 * AF_MAT_B + wave * 2 * AF_TILE_B + AF_TILE_B;
  char * outs = lds + 2 * AF_MAT_B + wave * 2 * AF_TILE_B;
  char * ins = outs + AF_TILE_B;
  unsigned int * tile_word = reinterpret_cast<unsigned int *>(lds + 2 * AF_MAT_B + 8 * AF_TILE_B);
  // what never changes for a lane (five registers)
  const unsigned int lane16 = lane * 16u;
  const unsigned int boff = n * 640u + rate * 160u + q * 32u;        // its four states 4q.. of sub-tile 0 in a stage
  const unsigned int boff5 = n * 640u + rate * 160u + 128u + q * 8u; // state 16 + q
  const char * xlane = lds + lane * 8u, * ylane = lds + AF_MAT_B + lane * 8u;
  const unsigned long long sink_a = (unsigned long long)(uintptr_t)(sink + ((size_t)blockIdx.x * 4u + wave) * 4u);
  const unsigned long long aorder_a = (unsigned long long)(uintptr_t)aorder;

  AF_SLOT_VARS(s0)
  AF_SLOT_VARS(s1)
  AF_SLOT_VARS(s2)
  AF_SLOT_VARS(s3)
  AF_SLOT_VARS(s4)

  // a matrix block into LDS: 13 pieces of 1 KB dealt to the four waves
  auto stage_matrix = [&](unsigned int buf_b, unsigned int off) __attribute__((always_inline)) {
#pragma unroll
    for (unsigned int r = 0; r < 4; ++r)
    {
      const unsigned int piece = r * 4u + wave;
      if (piece < (unsigned int)AF_MAT_PIECES) af_dma16(buf_b + piece * 1024u, aorder_a + off + piece * 1024u, lane16);
    }
  };

  const size_t tiles = ((size_t)sites + AF_WGS - 1) / AF_WGS;
  // the first op's left block (later tiles: requested by the last op of the tile before)
  {
    const AfW<16, 3> r1 = af_load<16, 3>(plan, 1);
    if ((r1[18] & AF_KIND_MASK) == 0u) stage_matrix(xbuf_b, r1[16]);
  }
  for (size_t round = 0;; ++round)
  {
    // a workgroup's first tiles are its own by a fixed stride, the last rounds' worth come from a
    // counter (the XCDs do not write at the same rate, partials_fused.hip)
    size_t tile;
    if (round < static_rounds || !next_tile)
      tile = (size_t)blockIdx.x + round * gridDim.x;
    else
    {
      if (threadIdx.x == 0) *tile_word = atomicAdd(next_tile, 1u);
      __syncthreads();
      tile = (size_t)static_rounds * gridDim.x + (unsigned int)__builtin_amdgcn_readfirstlane((int)*tile_word);
    }
    if (tile >= tiles) break;
    const size_t site0 = tile * AF_WGS + (size_t)wave * AF_WS;
    const unsigned long long clv_off = (unsigned long long)site0 * 640u;
    const unsigned long long cnt_off = (unsigned long long)site0 * 4u;

    // an operand without a slot: from HBM through the wave's in stage into a slot (AF_RELOAD_TAKE)
    auto reload_issue = [&](unsigned long long src) __attribute__((always_inline)) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (the stage's last readers are done)
#pragma unroll
      for (unsigned int it = 0; it < 5; ++it) af_dma16(in_b + it * 1024u, src + clv_off + it * 1024u, lane16);
    };
    auto reload_counts = [&](unsigned long long cnt, unsigned int (&cj)[AF_J]) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < AF_J; ++j)
      {
        cj[j] = 0u;
        if (MODE != SCALE_NONE && cnt) cj[j] = *(const unsigned int PLL_GLOBAL *)(af_base(cnt + cnt_off) + (4u * j + n) * 4u);
      }
    };
    // (a macro, not a lambda: a lambda would capture the slot variables by reference, and
    // variables whose address is taken anywhere stay in memory)
#define AF_RELOAD_TAKE(slot, cj)                                   \
  {                                                                \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               \
    AfSlot tmp;                                                    \
    af_read_tile(ins + boff, ins + boff5, tmp.v);                  \
    tmp.c[0] = cj[0];                                              \
    tmp.c[1] = cj[1];                                              \
    AF_SLOT_WRITE(slot, tmp)                                       \
  }
    // the tip characters of an op (record words 8..15: four rows): lane l holds those of site l & 7
    auto request_chars = [&](const AfW<8, 11> & r, unsigned int (&c)[4]) __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < 4; ++k) c[k] = *(const unsigned char PLL_GLOBAL *)(af_base(r.quad(8 + 2 * k) + site0) + (lane & 7u));
    };

    // ---- prologue: what the op before op 0 would have done for it
    unsigned int ch[4];
    {
      const AfW<16, 12> h = af_load<16, 12>(plan, 0);
      const unsigned int hf = __builtin_amdgcn_readfirstlane(h[18]);
      if (hf & AF_RELOAD_A)
      {
        unsigned int cj[AF_J];
        reload_issue(h.quad(20));
        reload_counts(h.quad(22), cj);
        AF_RELOAD_TAKE((h[19] >> 12) & 15u, cj)
      }
      if (hf & AF_RELOAD_B)
      {
        unsigned int cj[AF_J];
        reload_issue(h.quad(24));
        reload_counts(h.quad(26), cj);
        AF_RELOAD_TAKE((h[19] >> 16) & 15u, cj)
      }
      request_chars(af_load<8, 11>(plan, 1), ch);
    }
    AfW<0, 28> rc = af_load<0, 28>(plan, 1);

    for (unsigned int i = 0; i < nops; ++i)
    {
      const unsigned int fl = __builtin_amdgcn_readfirstlane(rc[18]);
      const unsigned int kind = fl & AF_KIND_MASK;
      const unsigned int slots = __builtin_amdgcn_readfirstlane(rc[19]);
      const bool scaling = MODE != SCALE_NONE && (fl & AF_SCALING);

      // ---- barrier 1: the left block has landed everywhere, everybody is done with op i - 1.
      // The block was requested BEFORE the previous op's six stores: they may stay in flight.
      if (i == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kind <= 1u) stage_matrix(ybuf_b, rc[17]);
      unsigned int ra_cj[AF_J], rb_cj[AF_J];
      if (fl & AF_RELOAD_A)
      {
        reload_issue(rc.quad(20));
        reload_counts(rc.quad(22), ra_cj);
      }

      double x[AF_J][5];        // the left factor, then the product
      unsigned int lc[AF_J], rcn[AF_J];
#pragma unroll
      for (int j = 0; j < AF_J; ++j) lc[j] = rcn[j] = 0u;
      if (kind == 0u)
      {
        AfSlot l;
        AF_SLOT_READ(slots & 15u, l)
#pragma unroll
        for (int j = 0; j < AF_J; ++j) lc[j] = l.c[j];
        af_matvec<false>(xlane, q, l.v, x);
      }
      else if (kind == 1u)
      {
        // tip-inner: the tip's factor is a row of its table (requested now, used after the products)
        const af_gptr tab = af_base(rc.quad(4));
#pragma unroll
        for (int j = 0; j < AF_J; ++j)
        {
          unsigned int code = (unsigned int)__shfl((int)ch[0], 4 * j + (int)n, 64);
          if (code >= ms) code = 0;
          const unsigned int e = (code * 4u + rate) * 160u;
          const pll_v2d v0 = *(const pll_v2d PLL_GLOBAL *)(tab + (e + q * 32u));
          const pll_v2d v1 = *(const pll_v2d PLL_GLOBAL *)(tab + (e + q * 32u + 16u));
          x[j][0] = v0.x; x[j][1] = v0.y; x[j][2] = v1.x; x[j][3] = v1.y;
          x[j][4] = *(const double PLL_GLOBAL *)(tab + (e + 128u + q * 8u));
        }
      }
      if (fl & AF_RELOAD_A) AF_RELOAD_TAKE((slots >> 12) & 15u, ra_cj)

      // ---- barrier 2: the right block has landed, everybody is done with the left one
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // what the next op needs: its characters, its left block (the six stores of this op
      // follow: barrier 1 of the next op lets exactly those stay in flight).  (The record behind
      // the last op's is a copy of op 0's: the next tile begins with it.)
      unsigned int nch[4];
      {
        const AfW<8, 11> rn = af_load<8, 11>(plan, i + 2);
        request_chars(rn, nch);
        asm volatile("" ::: "memory");
        if ((rn[18] & AF_KIND_MASK) == 0u) stage_matrix(xbuf_b, rn[16]);
      }
      if (fl & AF_RELOAD_B)
      {
        reload_issue(rc.quad(24));
        reload_counts(rc.quad(26), rb_cj);
      }

      double2 g[5];            // the finished tile, 16 bytes per lane: granule it * 64 + lane
      unsigned int pc[AF_J];   // the parent's counts, lane's own site of each sub-tile
#pragma unroll
      for (int j = 0; j < AF_J; ++j) pc[j] = 0u;
      AfSlot p;                // the parent in the operand layout (what its slot takes)
#pragma unroll
      for (int j = 0; j < AF_J; ++j)
      {
#pragma unroll
        for (int t = 0; t < 5; ++t) p.v[j][t] = 0.0;
        p.c[j] = 0u;
      }
      if (kind <= 1u)
      {
        {
          AfSlot r;
          AF_SLOT_READ((slots >> 4) & 15u, r)
#pragma unroll
          for (int j = 0; j < AF_J; ++j) rcn[j] = r.c[j];
          af_matvec<true>(ylane, q, r.v, x);
        }
#pragma unroll
        for (int j = 0; j < AF_J; ++j)
        {
          bool small = true;
#pragma unroll
          for (int t = 0; t < 5; ++t)
          {
            p.v[j][t] = x[j][t];
            small = small && (p.v[j][t] < PLLHIP_SCALE_THRESHOLD);
          }
          // scaling rule of core_partials_avx2.c:752-800: every entry of the site below the threshold
          unsigned int scaled = 0u;
          if (scaling)
          {
            const unsigned long long bal = __ballot(small), m = 0x1111111111111111ull << n;
            scaled = (bal & m) == m ? 1u : 0u;
            if (__ballot(scaled != 0u))
            {
              const double f = scaled ? PLLHIP_SCALE_FACTOR : 1.0;
#pragma unroll
              for (int t = 0; t < 5; ++t) p.v[j][t] *= f;
            }
          }
          pc[j] = scaling ? ((fl & AF_LCNT) ? lc[j] : 0u) + ((fl & AF_RCNT) ? rcn[j] : 0u) + scaled : 0u;
          p.c[j] = pc[j];
        }
        // through the wave's out stage into the layout of the stores
#pragma unroll
        for (int j = 0; j < AF_J; ++j)
        {
          *reinterpret_cast<double2 *>(outs + boff + j * 2560) = make_double2(p.v[j][0], p.v[j][1]);
          *reinterpret_cast<double2 *>(outs + boff + j * 2560 + 16) = make_double2(p.v[j][2], p.v[j][3]);
          *reinterpret_cast<double *>(outs + boff5 + j * 2560) = p.v[j][4];
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int it = 0; it < 5; ++it) g[it] = *reinterpret_cast<const double2 *>(outs + it * 1024 + lane16);
        asm volatile("" ::: "memory");
      }
      else
      {
        // lookup: parent = TL[pair 1] (.) TR[pair 2] (k_aa_cherry_rounds, partials_aa_mfma.hip)
        // (the lane number through an empty asm: what is derived from it below -- site and column
        // of five granules -- is recomputed here instead of living in ten registers all along)
        unsigned int lane_l = lane;
        asm volatile("" : "+v"(lane_l));
        unsigned int c1 = ch[0], c2 = ch[1], c3 = ch[2], c4 = ch[3];
        if (c1 >= ms) c1 = 0;
        if (c2 >= ms) c2 = 0;
        if (c3 >= ms) c3 = 0;
        if (c4 >= ms) c4 = 0;
        const unsigned int p1 = c1 * ms + c2, p2 = c3 * ms + c4;
        const af_gptr tl = af_base(rc.quad(4)), tr = af_base(rc.quad(6));
        unsigned long long bal[5];
        // in two halves the scheduler may not mix (it would keep all five iterations' indices,
        // addresses and operands alive at once, on top of the slots): where, fetch, multiply
        auto lookup_part = [&](auto first_c, auto count_c) __attribute__((always_inline)) {
          constexpr unsigned int FIRST = decltype(first_c)::value, COUNT = decltype(count_c)::value;
          unsigned int o1[COUNT], o2[COUNT];
#pragma unroll
          for (unsigned int u = 0; u < COUNT; ++u)
          {
            const unsigned int gi = (FIRST + u) * 64u + lane_l, sl = gi / 40u, rr = gi - 40u * sl;
            const unsigned int q1 = (unsigned int)__shfl((int)p1, (int)sl, 64);
            const unsigned int q2 = (unsigned int)__shfl((int)p2, (int)sl, 64);
            o1[u] = (q1 * 40u + rr) * 16u;
            o2[u] = (q2 * 40u + rr) * 16u;
          }
          __builtin_amdgcn_sched_barrier(0);
          pll_v2d a[COUNT], b[COUNT];
#pragma unroll
          for (unsigned int u = 0; u < COUNT; ++u)
          {
            a[u] = *(const pll_v2d PLL_GLOBAL *)(tl + o1[u]);
            b[u] = *(const pll_v2d PLL_GLOBAL *)(tr + o2[u]);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (unsigned int u = 0; u < COUNT; ++u)
          {
            g[FIRST + u] = make_double2(a[u].x * b[u].x, a[u].y * b[u].y);
            bal[FIRST + u] = __ballot((g[FIRST + u].x < PLLHIP_SCALE_THRESHOLD) & (g[FIRST + u].y < PLLHIP_SCALE_THRESHOLD));
          }
          __builtin_amdgcn_sched_barrier(0);
        };
        lookup_part(std::integral_constant<unsigned int, 0>{}, std::integral_constant<unsigned int, 3>{});
        lookup_part(std::integral_constant<unsigned int, 3>{}, std::integral_constant<unsigned int, 2>{});
        // bit s: every entry of site s of the tile below the threshold.  The 320 flags of the tile
        // are five wave masks; site s owns bits [40 s, 40 s + 40).  (Nearly always no flag is set.)
        unsigned int scaled = 0u;
        if (scaling && (bal[0] | bal[1] | bal[2] | bal[3] | bal[4]) != 0ull)
        {
#pragma unroll
          for (int s = 0; s < AF_WS; ++s)
          {
            constexpr unsigned long long F40 = (1ull << 40) - 1ull;
            const int w = (40 * s) / 64, off = (40 * s) % 64;
            unsigned long long field = bal[w] >> off;
            if (off > 24) field |= bal[w + 1 < 5 ? w + 1 : 4] << (64 - off);
            scaled |= ((field & F40) == F40) ? 1u << s : 0u;
          }
          scaled = __builtin_amdgcn_readfirstlane(scaled);
          if (scaled)
#pragma unroll
            for (unsigned int it = 0; it < 5; ++it)
            {
              const unsigned int sl = (it * 64u + lane_l) / 40u;
              const double f = ((scaled >> sl) & 1u) ? PLLHIP_SCALE_FACTOR : 1.0;
              g[it].x *= f;
              g[it].y *= f;
            }
        }
#pragma unroll
        for (int j = 0; j < AF_J; ++j) pc[j] = (scaled >> (4 * j + n)) & 1u; // (both children are tip-tip results: nothing to inherit)
        if (fl & AF_HAS_PSLOT)
        {
#pragma unroll
          for (int it = 0; it < 5; ++it) *reinterpret_cast<double2 *>(outs + it * 1024 + lane16) = g[it];
          asm volatile("" ::: "memory");
          af_read_tile(outs + boff, outs + boff5, p.v);
          asm volatile("" ::: "memory");
#pragma unroll
          for (int j = 0; j < AF_J; ++j) p.c[j] = pc[j];
        }
      }
      // the parent's slot (ONE place for both kinds, and no branch around it: slot 15 is nobody's)
      AF_SLOT_WRITE((fl & AF_HAS_PSLOT) ? (slots >> 8) & 15u : 15u, p)
      if (fl & AF_RELOAD_B) AF_RELOAD_TAKE((slots >> 16) & 15u, rb_cj)

      // ---- the six stores (always six: barrier 1 of the next op counts on it): the counts --
      // to the wave's sink when the op has no scale buffer --, then the tile, 5 KB contiguous
      {
        const unsigned int mine = (lane & 4u) ? pc[1] : pc[0];
        const af_gptr cdst = af_base(scaling ? rc.quad(2) + cnt_off : sink_a);
        asm volatile("" ::: "memory");
        if (lane < (unsigned int)AF_WS) *(unsigned int PLL_GLOBAL *)(cdst + lane * 4u) = mine;
        asm volatile("" ::: "memory");
        const af_gptr out = af_base(rc.quad(0) + clv_off);
#pragma unroll
        for (unsigned int it = 0; it < 5; ++it)
        {
          const pll_v2d v = {g[it].x, g[it].y};
          pll_v2d PLL_GLOBAL * dst = (pll_v2d PLL_GLOBAL *)(out + (lane16 + it * 1024u));
          if (NT) __builtin_nontemporal_store(v, dst);
          else *dst = v;
          asm volatile("" ::: "memory");
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) ch[k] = nch[k];
      // the next op's record: on its way during barrier 1
      rc = af_load<0, 28>(plan, i + 2);
    }
  }
}
} // namespace

// ---------------------------------------------------------------- host

struct pllhip_aa_fused_cache
{
  std::vector<pllhip_op_t> last_ops;
  unsigned int epoch = 0, maxstates = 0;
  // what runs ahead of the list kernel
  std::vector<PartialsArgs> tt_ops;   // tip-tip ops, grouped by scaling mode
  std::vector<int> tt_modes;
  std::vector<PartialsArgs> lk_ops, lk_k1, lk_k2; // lookup ops and the tip-tip ops that made their children
  unsigned int nops = 0, nmat = 0, ntip = 0;
  int mode = SCALE_NONE;
  size_t off_mat = 0, off_tip = 0;    // job arrays within d_plan
  void * d_plan = nullptr;
  void * h_plan = nullptr;
  size_t plan_cap = 0;
  hipEvent_t done = nullptr;
  bool pending = false;
  char * d_aorder = nullptr;
  size_t aorder_cap = 0;
  char * d_titab = nullptr;
  size_t titab_cap = 0;
};

void pllhip_aa_fused_free(pllhip_ctx * c)
{
  pllhip_aa_fused_cache * k = c->aa_fused;
  if (!k) return;
  if (k->d_plan) (void)hipFree(k->d_plan);
  if (k->h_plan) (void)hipHostFree(k->h_plan);
  if (k->done) (void)hipEventDestroy(k->done);
  if (k->d_aorder) (void)hipFree(k->d_aorder);
  if (k->d_titab) (void)hipFree(k->d_titab);
  delete k;
  c->aa_fused = nullptr;
}

// everything the kept plan describes, again: tip-tip ops, tables, matrices in operand order, the list
static int aa_fused_launch(pllhip_ctx * c, bool tables_built)
{
  pllhip_aa_fused_cache & k = *c->aa_fused;
  // tip-tip ops of one scaling mode per launch, PLLHIP_BATCH_MAX at a time
  for (size_t first = 0; first < k.tt_ops.size();)
  {
    PartialsBatch b;
    unsigned int nb = 0;
    const int mode = k.tt_modes[first];
    while (first < k.tt_ops.size() && k.tt_modes[first] == mode && nb < PLLHIP_BATCH_MAX) b.op[nb++] = k.tt_ops[first++];
    const int rc = pllhip_launch_aa_batch(c, b, nb, 2, mode);
    if (rc) return rc;
  }
  if (!k.lk_ops.empty() && !tables_built)
  {
    // (same pool, same places: the records' addresses hold while the epoch does)
    std::vector<AaLookupTables> tabs(k.lk_ops.size());
    const int rc = pllhip_aa_lookup_tables(c, k.lk_ops.data(), k.lk_k1.data(), k.lk_k2.data(),
                                           (unsigned int)k.lk_ops.size(), tabs.data());
    if (rc) return rc;
  }
  const char * plan = static_cast<const char *>(k.d_plan);
  if (k.nmat + k.ntip)
  {
    k_af_prepare<<<k.nmat + k.ntip, 256, 0, c->stream>>>((const AfMatJob *)(plan + k.off_mat), k.nmat,
                                                         (const AfTipJob *)(plan + k.off_tip), k.ntip, k.d_aorder,
                                                         k.d_titab, c->tipmap, c->maxstates);
    HIP_TRY(hipGetLastError());
  }
  const size_t tiles = ((size_t)c->sh.sites + AF_WGS - 1) / AF_WGS;
  size_t grid = tiles;
  const size_t cap = (size_t)c->num_cus * 2;
  if (grid > cap) grid = cap;
  const size_t rounds = tiles / grid;
  const unsigned int dynamic_rounds = getenv("PLLHIP_FUSED_DYNAMIC_ROUNDS") ? (unsigned int)atoi(getenv("PLLHIP_FUSED_DYNAMIC_ROUNDS"))
                                                                            : (unsigned int)std::max<size_t>(2, rounds / 3);
  const unsigned int static_rounds = rounds > dynamic_rounds ? (unsigned int)(rounds - dynamic_rounds) : 1u;
  unsigned int * counter = getenv("PLLHIP_FUSED_STATIC_TILES") ? nullptr : c->d_tile_counter;
  HIP_TRY(hipMemsetAsync(c->d_tile_counter, 0, sizeof(unsigned int), c->stream));
  const bool nt = pllhip_use_nt(c);
#define AF_LAUNCH(MODEV, NTV)                                                                                          \
  do {                                                                                                                 \
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_aa_fused<MODEV, NTV>),                               \
                                hipFuncAttributeMaxDynamicSharedMemorySize, AF_LDS_B));                                \
    k_aa_fused<MODEV, NTV><<<(unsigned int)grid, 256, AF_LDS_B, c->stream>>>(                                          \
        (const AaRec *)k.d_plan, k.nops, c->sh.sites, k.d_aorder, c->maxstates, (double2 *)c->d_sink, counter,         \
        static_rounds);                                                                                                \
  } while (0)
  if (k.mode == SCALE_NONE) { if (nt) AF_LAUNCH(SCALE_NONE, true); else AF_LAUNCH(SCALE_NONE, false); }
  else { if (nt) AF_LAUNCH(SCALE_SITE, true); else AF_LAUNCH(SCALE_SITE, false); }
#undef AF_LAUNCH
  HIP_TRY(hipGetLastError());
  return 0;
}

// Returns 0 when the list has been enqueued, 1 when it is not one this path takes (the caller
// launches per level), < 0 on error.
int pllhip_aa_fused_update(pllhip_ctx * c, const pllhip_op_t * ops, unsigned int count)
{
  if (c->sh.states != 20 || c->sh.rate_cats != 4 || c->sh.rate_scalers || c->sh.asc_states || !c->rows.empty() ||
      c->aa_exact || count > PLLHIP_FUSED_MAX_OPS)
    return 1;
  if (c->sh.pattern_tip && (c->maxstates < 1 || c->maxstates > 32)) return 1;
  if (!c->aa_fused) c->aa_fused = new pllhip_aa_fused_cache();
  pllhip_aa_fused_cache & k = *c->aa_fused;
  if (k.last_ops.size() == count && k.epoch == c->layout_epoch && k.maxstates == c->maxstates &&
      !getenv("PLLHIP_FUSED_DEBUG") && memcmp(k.last_ops.data(), ops, (size_t)count * sizeof(pllhip_op_t)) == 0)
    return aa_fused_launch(c, false);
  k.last_ops.clear();

  // ---- classify.  Tip-tip ops run ahead of the list: allowed only if nothing earlier in the list
  // wrote or read what they write (they read tips only).  An inner-inner op over two tip-tip
  // results of this list, or a tip-inner op over one, is a lookup.
  std::vector<PartialsArgs> args(count);
  std::vector<int> kinds(count), modes(count);
  const size_t nclv = c->clv.size(), nsc = c->sh.scale_buffers;
  std::vector<int> clv_touched(nclv, 0), sc_touched(nsc, 0), tt_writer(nclv, -1);
  std::vector<std::pair<int, int>> lk_kids(count, {-1, -1});
  bool any_scaler = false;
  const bool lookups_ok = pllhip_aa_cherry_covers(c, SCALE_SITE);
  for (unsigned int i = 0; i < count; ++i)
  {
    const int rc = pllhip_resolve_op(c, ops[i], args[i], kinds[i], modes[i]);
    if (rc) return rc;
    const pllhip_op_t & op = ops[i];
    if (kinds[i] == 2 && (clv_touched[op.parent_clv] || (op.parent_scaler >= 0 && sc_touched[op.parent_scaler]))) return 1;
    any_scaler = any_scaler || modes[i] != SCALE_NONE;
    const int plain = kinds[i];
    if (lookups_ok && plain == 0 && tt_writer[op.child1_clv] >= 0 && tt_writer[op.child2_clv] >= 0)
    {
      lk_kids[i] = {tt_writer[op.child1_clv], tt_writer[op.child2_clv]};
      kinds[i] = 3;
    }
    else if (lookups_ok && plain == 1)
    {
      const unsigned int inner = pllhip_is_tip(c, op.child1_clv) ? op.child2_clv : op.child1_clv;
      if (tt_writer[inner] >= 0)
      {
        lk_kids[i] = {-2, tt_writer[inner]};
        kinds[i] = 3;
      }
    }
    tt_writer[op.parent_clv] = plain == 2 ? (int)i : -1;
    clv_touched[op.parent_clv] = clv_touched[op.child1_clv] = clv_touched[op.child2_clv] = 1;
    if (op.parent_scaler >= 0) sc_touched[op.parent_scaler] = 1;
    if (op.child1_scaler >= 0) sc_touched[op.child1_scaler] = 1;
    if (op.child2_scaler >= 0) sc_touched[op.child2_scaler] = 1;
  }

  // ---- the list the kernel walks: everything but the tip-tip ops, ordered and given slots by the
  // planner of the 4-state kernel (a lookup has no inner operands: a "tip-tip" op to the planner)
  std::vector<pllhip_op_t> rops;
  std::vector<PartialsArgs> rargs;
  std::vector<int> rkinds, orig;
  k.tt_ops.clear();
  k.tt_modes.clear();
  for (int pass = 0; pass < 2; ++pass) // (tip-tip ops grouped by mode: without a scale buffer first)
    for (unsigned int i = 0; i < count; ++i)
      if (kinds[i] == 2 && (modes[i] != SCALE_NONE) == (pass == 1))
      {
        k.tt_ops.push_back(args[i]);
        k.tt_modes.push_back(modes[i]);
      }
  for (unsigned int i = 0; i < count; ++i)
    if (kinds[i] != 2)
    {
      rops.push_back(ops[i]);
      rargs.push_back(args[i]);
      rkinds.push_back(kinds[i] == 3 ? 2 : kinds[i]);
      orig.push_back((int)i);
    }
  const unsigned int n = (unsigned int)rops.size();
  if (n == 0)
  {
    // nothing but tip-tip ops
    k.nops = 0;
    return 1;
  }
  std::vector<FusedOp> fplan;
  unsigned int reloads = 0;
  const FusedGeom geom = {nclv, nsc, c->sh.tips, c->sh.pattern_tip != 0};
  int rc = pllhip_fused_plan(geom, rops.data(), rargs.data(), rkinds.data(), n, AF_NSLOT, fplan, &reloads);
  if (rc) return rc;

  // ---- encode
  if (!c->fused_zero_row)
  {
    const size_t bytes = (size_t)c->sh.sites + PLLHIP_TAIL_SITES + 256;
    HIP_TRY(hipMalloc((void **)&c->fused_zero_row, bytes));
    HIP_TRY(hipMemsetAsync(c->fused_zero_row, 0, bytes, c->stream));
  }
  if (!c->d_sink) HIP_TRY(hipMalloc(&c->d_sink, (size_t)c->num_cus * 16 * 80 * sizeof(double2)));
  if (!c->d_tile_counter) HIP_TRY(hipMalloc((void **)&c->d_tile_counter, sizeof(unsigned int)));
  k.lk_ops.clear();
  k.lk_k1.clear();
  k.lk_k2.clear();
  std::vector<int> lk_index(n, -1);
  for (unsigned int pos = 0; pos < n; ++pos)
  {
    const int ri = fplan[pos].list_pos, oi = orig[ri];
    if (kinds[oi] != 3) continue;
    lk_index[pos] = (int)k.lk_ops.size();
    k.lk_ops.push_back(args[oi]);
    if (lk_kids[oi].first >= 0) k.lk_k1.push_back(args[lk_kids[oi].first]);
    else
    {
      PartialsArgs none; // marks a tip-inner lookup op (no producing op on the tip's side)
      memset(&none, 0, sizeof(none));
      k.lk_k1.push_back(none);
    }
    k.lk_k2.push_back(args[lk_kids[oi].second]);
  }
  // the lookup tables' addresses are needed in the records: build them now (their pool may move)
  std::vector<AaLookupTables> tabs(k.lk_ops.size());
  if (!k.lk_ops.empty())
  {
    rc = pllhip_aa_lookup_tables(c, k.lk_ops.data(), k.lk_k1.data(), k.lk_k2.data(), (unsigned int)k.lk_ops.size(),
                                 tabs.data());
    if (rc) return rc;
  }
  std::vector<AaRec> recs(n + 2);
  std::vector<AfMatJob> mj;
  std::vector<AfTipJob> tj;
  memset(recs.data(), 0, recs.size() * sizeof(AaRec));
  const size_t tip_tab_b = (size_t)c->maxstates * 80 * sizeof(double);
  const unsigned long long zero_row = (unsigned long long)(uintptr_t)c->fused_zero_row;
  k.mode = any_scaler ? SCALE_SITE : SCALE_NONE;
  auto slot4 = [](int s) { return (unsigned int)(s > 0 ? s : 0) & 15u; };
  auto reloads_of = [&](AaRec & r, const FusedOp & f) {
    // what is done during the op before `f` for it
    if (f.dma_flags & 1)
    {
      r.flags |= AF_RELOAD_A;
      r.ra_src = (unsigned long long)(uintptr_t)f.left_hbm;
      r.ra_cnt = (unsigned long long)(uintptr_t)f.lsc_hbm;
      r.slots |= slot4(f.lslot) << 12;
    }
    if (f.dma_flags & 2)
    {
      r.flags |= AF_RELOAD_B;
      r.rb_src = (unsigned long long)(uintptr_t)f.right_hbm;
      r.rb_cnt = (unsigned long long)(uintptr_t)f.rsc_hbm;
      r.slots |= slot4(f.rslot) << 16;
    }
  };
  reloads_of(recs[0], fplan[0]);
  for (unsigned int pos = 0; pos < n; ++pos)
  {
    const FusedOp & f = fplan[pos];
    const int oi = orig[f.list_pos];
    const int kind = kinds[oi] == 3 ? 2 : kinds[oi];
    AaRec & r = recs[pos + 1];
    r.parent = (unsigned long long)(uintptr_t)f.parent;
    r.pscaler = (unsigned long long)(uintptr_t)f.pscaler;
    r.flags = (unsigned int)kind;
    if (f.pslot >= 0) r.flags |= AF_HAS_PSLOT;
    if (f.pscaler) r.flags |= AF_SCALING;
    if (kind == 0 && f.lsc_slot >= 0) r.flags |= AF_LCNT;
    if (kind <= 1 && f.rsc_slot >= 0) r.flags |= AF_RCNT;
    r.slots = slot4(f.lslot) | slot4(f.rslot) << 4 | slot4(f.pslot) << 8;
    for (int t = 0; t < 4; ++t) r.row[t] = zero_row;
    if (kind == 0)
    {
      r.xoff = (unsigned int)(mj.size() * AF_MAT_B);
      mj.push_back(AfMatJob{f.lmat, (unsigned long long)r.xoff});
    }
    if (kind <= 1)
    {
      r.yoff = (unsigned int)(mj.size() * AF_MAT_B);
      mj.push_back(AfMatJob{f.rmat, (unsigned long long)r.yoff});
    }
    if (kind == 1)
    {
      r.row[0] = (unsigned long long)(uintptr_t)f.ltip;
      r.tab_l = (unsigned long long)(tj.size() * tip_tab_b); // (made absolute below)
      tj.push_back(AfTipJob{f.lmat, (unsigned long long)(tj.size() * tip_tab_b)});
    }
    if (kind == 2)
    {
      const AaLookupTables & t = tabs[lk_index[pos]];
      r.tab_l = (unsigned long long)(uintptr_t)t.tl;
      r.tab_r = (unsigned long long)(uintptr_t)t.tr;
      r.row[0] = (unsigned long long)(uintptr_t)t.t1;
      r.row[1] = (unsigned long long)(uintptr_t)t.t2;
      r.row[2] = (unsigned long long)(uintptr_t)t.t3;
      r.row[3] = (unsigned long long)(uintptr_t)t.t4;
    }
    if (pos + 1 < n) reloads_of(r, fplan[pos + 1]);
  }
  if ((mj.size() + 1) * (size_t)AF_MAT_B > 0xffffffffull) return 1;
  // buffers: matrices in operand order, tip tables
  if (k.aorder_cap < (mj.size() + 1) * (size_t)AF_MAT_B)
  {
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (k.d_aorder) HIP_TRY(hipFree(k.d_aorder));
    k.d_aorder = nullptr;
    k.aorder_cap = (mj.size() + 1) * (size_t)AF_MAT_B * 2;
    HIP_TRY(hipMalloc((void **)&k.d_aorder, k.aorder_cap));
  }
  if (k.titab_cap < (tj.size() + 1) * tip_tab_b)
  {
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (k.d_titab) HIP_TRY(hipFree(k.d_titab));
    k.d_titab = nullptr;
    k.titab_cap = (tj.size() + 1) * tip_tab_b * 2;
    HIP_TRY(hipMalloc((void **)&k.d_titab, k.titab_cap));
  }
  for (unsigned int pos = 0; pos < n; ++pos)
    if ((recs[pos + 1].flags & AF_KIND_MASK) == 1u) recs[pos + 1].tab_l += (unsigned long long)(uintptr_t)k.d_titab;
  recs[n + 1] = recs[1]; // (the next tile begins with op 0: its left block is requested by the last op)
  recs[n + 1].flags &= ~(AF_RELOAD_A | AF_RELOAD_B);

  const size_t rec_b = recs.size() * sizeof(AaRec), mat_b = (mj.size() + 1) * sizeof(AfMatJob),
               tip_b = (tj.size() + 1) * sizeof(AfTipJob);
  const size_t bytes = rec_b + mat_b + tip_b;
  if (k.plan_cap < bytes)
  {
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (k.d_plan) HIP_TRY(hipFree(k.d_plan));
    if (k.h_plan) HIP_TRY(hipHostFree(k.h_plan));
    k.d_plan = k.h_plan = nullptr;
    k.plan_cap = bytes * 2;
    HIP_TRY(hipMalloc(&k.d_plan, k.plan_cap));
    HIP_TRY(hipHostMalloc(&k.h_plan, k.plan_cap, hipHostMallocDefault));
    if (!k.done) HIP_TRY(hipEventCreateWithFlags(&k.done, hipEventDisableTiming));
    k.pending = false;
  }
  if (k.pending) HIP_TRY(hipEventSynchronize(k.done));
  char * stage = static_cast<char *>(k.h_plan);
  memcpy(stage, recs.data(), rec_b);
  if (!mj.empty()) memcpy(stage + rec_b, mj.data(), mj.size() * sizeof(AfMatJob));
  if (!tj.empty()) memcpy(stage + rec_b + mat_b, tj.data(), tj.size() * sizeof(AfTipJob));
  HIP_TRY(hipMemcpyAsync(k.d_plan, k.h_plan, bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipEventRecord(k.done, c->stream));
  k.pending = true;
  k.off_mat = rec_b;
  k.off_tip = rec_b + mat_b;
  k.nmat = (unsigned int)mj.size();
  k.ntip = (unsigned int)tj.size();
  k.nops = n;
  if (getenv("PLLHIP_FUSED_DEBUG"))
    fprintf(stderr, "pllhip 20-state list kernel: %u ops = %zu tip-tip ahead + %zu lookups + %u on the matrix cores, %u operands reloaded\n",
            count, k.tt_ops.size(), k.lk_ops.size(), n - (unsigned int)k.lk_ops.size(), reloads);
  rc = aa_fused_launch(c, true);
  if (rc) return rc;
  k.last_ops.assign(ops, ops + count);
  k.epoch = c->layout_epoch;
  k.maxstates = c->maxstates;
  return 0;
}
