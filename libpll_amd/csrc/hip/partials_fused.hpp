// partials_fused.hpp -- plan of the site-blocked whole-list kernel (partials_fused.hip)
#ifndef PLLHIP_PARTIALS_FUSED_HPP_
#define PLLHIP_PARTIALS_FUSED_HPP_

#include <vector>

#include "ctx.hpp"

#ifndef PLLHIP_FUSED_J
#define PLLHIP_FUSED_J 2 /* sub-steps (64 lanes x 16 B) per tile */
#endif

// What the planner decides for one op (host side; the device gets FusedRec below).
struct FusedOp
{
  double * parent;
  const double * left_hbm;         // inner child 1 when it is copied into lslot from HBM (reload), else nullptr
  const double * right_hbm;        // the same for the inner child of "right"
  const unsigned char * ltip;      // tip child (tip-inner), tip child 1 (tip-tip), else nullptr
  const unsigned char * rtip;      // tip child 2 (tip-tip), else nullptr
  const double * lmat;
  const double * rmat;
  unsigned int * pscaler;          // nullptr: no scaling
  const unsigned int * lsc_hbm;    // counts copied into the slot together with left_hbm / right_hbm, else nullptr
  const unsigned int * rsc_hbm;
  int lslot, rslot, pslot;         // LDS slots of the two inner children / the parent; -1 = none
  int lsc_slot, rsc_slot;          // LDS slots the inherited counts are taken from; -1 = none
  int kind;                        // 0 inner-inner, 1 tip-inner, 2 tip-tip
  int list_pos;                    // position of the op in the caller's list (the plan is re-ordered)
  int dma_flags;                   // bit 0 / 1: left_hbm / right_hbm is copied into lslot / rslot by LDS-DMA
                                   // at the top of the op before this one
  const double * pair_tab;         // tip operands: [256 code pairs][rate][state] table, else nullptr
};

// The plan as the kernel reads it, through the scalar data cache: 32 bytes per op, indices
// instead of pointers; the kernel rebuilds the addresses from FusedBases with scalar
// arithmetic.  Every wave walks the whole plan once per tile, each at its own position, so the
// plan should stay in the 16 KB scalar cache that neighbouring compute units share: the
// 128-byte entries of round 1 did that for 62 ops (7.9 KB) but not for the 126 / 198 ops of
// the 128- / 200-taxon lists (16 / 25 KB; now 4 / 6 KB).  Measured effect on this kernel:
// none beyond noise (the look-ahead already hid the misses; what slows the long lists down is
// the size of their CLVs, see launch_fused_rc) -- kept because it halves the plan upload, frees
// SGPRs (114-125 VGPRs, no spills into them) and bounds the scalar-cache footprint for any list.
struct FusedRec
{
  // words 0, 1: what the look-ahead loads of an op need (read three ops ahead of it)
  unsigned short ltip, rtip;       // tip index, 0xffff: none
  unsigned short lmat, rmat;       // P-matrix index
  unsigned int list_pos, pad1;
  // words 4..7: what the op itself needs (read two ops ahead: the op before it looks at
  // pair, dma_flags and the slots to gather from the pair table and to reload)
  unsigned short parent, pscaler;  // CLV / scale-buffer index (pscaler 0xffff: no scaling)
  unsigned short pair, src;        // pair-table number / FusedSrc number, 0xffff: none
  signed char lslot, rslot, pslot, kind;
  signed char lsc_slot, rsc_slot;
  unsigned char dma_flags, pad0;
};
static_assert(sizeof(FusedRec) == 32, "eight words per op");
#define PLLHIP_FUSED_NONE 0xffffu
#define PLLHIP_FUSED_MAX_INDEX 0xfffeu /* lists with larger buffer indices run per level */

// sources of the operands an op reloads (few ops have any: kept out of the records)
struct FusedSrc
{
  const double * left_hbm;
  const double * right_hbm;
  const unsigned int * lsc_hbm;
  const unsigned int * rsc_hbm;
};

// what the indices of a FusedRec are relative to (kernel argument; the FusedSrc entries follow
// the records in the plan buffer)
struct FusedBases
{
  double * clv;                    // CLV of arena position i at clv + i * site_stride * (states * rate_cats)
  unsigned int * scaler;           // scale buffer i at scaler + i * site_stride (* rate_cats with per-rate scalers)
  const unsigned char * tips;      // tip i at tips + i * tip_stride
  const double * pmat;
  double * pairtab;
  unsigned int site_stride;        // sites + slack of every per-site buffer
  unsigned int tip_stride;         // bytes
};

// (The planner is host logic and needs no device: what it must know of the partition is here.)
struct FusedGeom
{
  size_t nclv;            // CLV slots (tips + clv_buffers)
  size_t nsc;             // scale buffers
  unsigned int tips;
  bool pattern_tip;       // tips are character rows, not CLVs
  bool is_tip(unsigned int clv_index) const { return pattern_tip && clv_index < tips; }
};

// Order the list, assign slots.  args/kinds are resolve_op's results per op.  Returns 0 and
// fills plan (one entry per op, in the order the kernel runs them) and *reloads (operands copied
// back from HBM), 1 if the list is of a shape the kernel does not take (the caller then
// launches per level), < 0 on error.
int pllhip_fused_plan(const FusedGeom & geom, const pllhip_op_t * ops, const PartialsArgs * args,
                      const int * kinds, unsigned int count, unsigned int nslots,
                      std::vector<FusedOp> & plan, unsigned int * reloads);
unsigned int pllhip_fused_slots(const pllhip_ctx * c, unsigned int workgroups_per_cu);
int pllhip_launch_fused(pllhip_ctx * c, const std::vector<FusedOp> & plan, unsigned int nslots);
int pllhip_relaunch_fused(pllhip_ctx * c); // the same op list as in the previous whole-list call of this context

#endif
