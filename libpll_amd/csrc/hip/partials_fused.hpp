// partials_fused.hpp -- plan entry of the site-blocked whole-list kernel (partials_fused.hip)
#ifndef PLLHIP_PARTIALS_FUSED_HPP_
#define PLLHIP_PARTIALS_FUSED_HPP_

#include <vector>

#include "ctx.hpp"

#ifndef PLLHIP_FUSED_J
#define PLLHIP_FUSED_J 2 /* sub-steps (64 lanes x 16 B) per tile */
#endif

struct FusedOp
{
  double * parent;
  const double * left_hbm;         // inner child 1 when it must come from HBM (no LDS slot), else nullptr
  const double * right_hbm;        // the same for the inner child of "right"
  const unsigned char * ltip;      // tip child (tip-inner), tip child 1 (tip-tip), else nullptr
  const unsigned char * rtip;      // tip child 2 (tip-tip), else nullptr
  const double * lmat;
  const double * rmat;
  unsigned int * pscaler;          // nullptr: no scaling
  const unsigned int * lsc_hbm;    // inherited counts that must come from HBM, else nullptr
  const unsigned int * rsc_hbm;
  int lslot, rslot, pslot;         // LDS slots of the two inner children / the parent; -1 = none
  int lsc_slot, rsc_slot;          // LDS slots the inherited counts are taken from; -1 = none / HBM
  int kind;                        // 0 inner-inner, 1 tip-inner, 2 tip-tip
  int hbm_flags;                   // bit 0 / 1: lsc_hbm / rsc_hbm present
  int list_pos;                    // position of the op in the caller's list (the plan is re-ordered)
  const double * pair_tab;         // tip-tip ops: [256 code pairs][rate][state] parent entries, else nullptr
  int dma_flags;                   // reload plan: bit 0 / 1: left_hbm / right_hbm is copied into lslot / rslot
                                   // by LDS-DMA one op ahead (with lsc_hbm / rsc_hbm into the slot's counts)
  int pad2;
#ifdef PLLHIP_FUSED_PLAN_PAD /* experiment: a plan that does not fit the scalar cache */
  char pad_experiment[PLLHIP_FUSED_PLAN_PAD];
#endif
};

// Order the list, assign slots.  args/kinds/modes are resolve_op's results per op.
// Returns 0 and fills plan (one look-ahead entry more than there are ops) and *ext (some
// operand comes from HBM), 1 if the list is of a shape the kernel does not take (the caller
// then launches per level), < 0 on error.
// (The planner is host logic and needs no device: what it must know of the partition is here.)
struct FusedGeom
{
  size_t nclv;            // CLV slots (tips + clv_buffers)
  size_t nsc;             // scale buffers
  unsigned int tips;
  bool pattern_tip;       // tips are character rows, not CLVs
  bool is_tip(unsigned int clv_index) const { return pattern_tip && clv_index < tips; }
};
int pllhip_fused_plan(const FusedGeom & geom, const pllhip_op_t * ops, const PartialsArgs * args,
                      const int * kinds, unsigned int count, unsigned int nslots, bool reload,
                      std::vector<FusedOp> & plan, bool * ext, unsigned int * evictions);
unsigned int pllhip_fused_slots(const pllhip_ctx * c, unsigned int workgroups_per_cu);
int pllhip_launch_fused(pllhip_ctx * c, const std::vector<FusedOp> & plan, unsigned int nslots, bool ext);
int pllhip_relaunch_fused(pllhip_ctx * c); // the same op list as in the previous whole-list call of this context

#endif
