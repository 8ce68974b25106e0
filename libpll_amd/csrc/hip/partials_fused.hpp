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

// The plan as the kernel reads it, through the scalar data cache: ONE 64-byte record per op (one
// scalar load, one cache line), holding everything the wave does WHILE that op runs, ready
// to use -- absolute addresses and LDS byte offsets, no indices to multiply out:
//   - what it requests for the op two ahead (P-matrix offsets),
//   - the pair table it gathers from for the op one ahead and where that op's tip characters are
//     (which lanes of the wave's character registers), what that op reloads and which of its
//     matrices must be staged,
//   - the op itself: parent, scale buffer, LDS places of operands, parent and counts.
// Round 2 first had 32-byte records of 16-bit indices, decoded by the kernel; the counters say
// a wave-op then cost 124 scalar + 124 vector instructions of which 30 were arithmetic, and that
// the wave, not HBM, was the limit (tools/pmc_instmix.sh, tools/fused_timing.sh) -- so the
// host now does the decoding once per list.  The records hold addresses: they are rebuilt when an
// arena moves (layout_epoch).
struct FusedRec
{
  unsigned int chars;                    // tip characters of op + 1, PLLHIP_FUSED_CH_* below
  unsigned int pad[3];
  unsigned int req_lmat, req_rmat;       // byte offsets of its P-matrices in the matrix arena
  unsigned int gather_off;               // byte offset of the pair table of op + 1 (0: the table of zeros)
  unsigned int flags;                    // PLLHIP_FUSED_* below
  unsigned long long parent;             // CLV the op writes
  unsigned long long pscaler;            // its scale buffer (0: none)
  unsigned short lslot_b, rslot_b;       // LDS byte offsets (within the wave's slots) of the operands,
  unsigned short pslot_b, src;           //   of the parent; FusedSrc number of what op + 1 reloads
  unsigned short lcnt_b, rcnt_b;         // LDS byte offsets (within the wave's counts) the inherited
  unsigned short pcnt_b, list_pos;       //   counts are read from / the parent's are kept at
};
static_assert(sizeof(FusedRec) == 64, "sixteen words per op");
#define PLLHIP_FUSED_KIND_MASK 3u      /* 0 inner-inner, 1 tip-inner, 2 tip-tip */
#define PLLHIP_FUSED_HAS_PSLOT 4u      /* the parent is kept in a slot */
#define PLLHIP_FUSED_SCALING 8u        /* the op has a scale buffer */
#define PLLHIP_FUSED_LCNT 16u          /* counts are inherited from the left / right operand's slot */
#define PLLHIP_FUSED_RCNT 32u
#define PLLHIP_FUSED_STAGE_SHIFT 6     /* two bits: matrices op + 1 needs (2 both, 1 right only, 0 none) */
#define PLLHIP_FUSED_RELOAD_NEXT 256u  /* op + 1 reloads operands: FusedSrc number `src` */
#define PLLHIP_FUSED_MAX_OPS 60000u    /* longer lists run per level */
// FusedRec::chars -- a wave holds the characters of up to 1024 / (tile sites) tip rows at its tile, 16 bytes per
// lane, fetched with ONE load per tile (rows in the order the list uses them; lists with more tip operands fetch
// the next batch of rows when they get there):
#define PLLHIP_FUSED_CH_LPOS(x) ((x) & 0xffu)         /* first lane of the left tip's row (op + 1) */
#define PLLHIP_FUSED_CH_RPOS(x) (((x) >> 8) & 0xffu)  /* ... of the right tip's */
#define PLLHIP_FUSED_CH_LTIP (1u << 16)                /* op + 1 has a left / right tip */
#define PLLHIP_FUSED_CH_RTIP (1u << 17)
#define PLLHIP_FUSED_CH_LOAD (1u << 18)                /* op + 2's rows are in another batch: fetch batch (x >> 24) */

// sources and LDS destinations of the operands an op reloads (few ops have any: kept out of the records)
struct FusedSrc
{
  const double * left_hbm;
  const double * right_hbm;
  const unsigned int * lsc_hbm;
  const unsigned int * rsc_hbm;
  unsigned long long where; // lslot_b | rslot_b << 16 | lcnt_b << 32 | rcnt_b << 48
  unsigned long long pad;
};

// one pair table to build (k_dna_pair_tables)
struct FusedPairJob
{
  const double * lmat;
  const double * rmat;
  double * tab;
  unsigned long long tip_tip;
};

unsigned int pllhip_fused_char_batches(const unsigned int * tips, unsigned int count, unsigned int lpr,
                                       unsigned int * chars_out, unsigned int * batch_out);

// Round 5: SEGMENTS.  Ops that share no buffer any of them writes are independent lists -- the two sides of the root
// edge of a full traversal, above all -- and a tile of sites may be taken through each of them by a different wave at
// the same time.  A launch whose tiles do not fill the chip a few times over (the 25-125 k sites an 8-way split of
// the BASELINE alignments leaves a GPU, real protein data) hands out (tile, segment) pairs instead of tiles: twice
// the work items of half the length, so the time of a launch with fewer tiles than wave slots -- one tile's serial
// walk of the list, ~1 us per op whatever the site count -- halves, and the last round's idle slots shrink with the
// items.  The reference has no counterpart (its cost per op is proportional to the sites, partials.c:177-213).
// Each segment is a plan of its own (order, slots, header records); reload sources, pair tables and character rows
// are numbered across the segments.  pllhip_fused_segments is host logic (CPU tests: tests/test_host.py).
struct FusedSeg
{
  unsigned int rec_first;   // the segment's two header records begin here (in records)
  unsigned int nops;
  unsigned int first_batch; // the batch of character rows its first ops use
  unsigned int pad;
};
#define PLLHIP_FUSED_MAX_SEGS 8u

// what the offsets of a FusedRec are relative to (kernel argument)
struct FusedBases
{
  const double * pmat;
  const double * pairtab;
  const unsigned long long * rowtab; // [batch][64]: the address each lane fetches its 16 bytes of tip characters from (+ site)
  const struct FusedSrc * srcs;      // reload sources, numbered across the segments
  const FusedSeg * segs;             // [nsegs] (nsegs == 1: not read, the kernel's own arguments say it all)
  unsigned int nsegs;
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
// The list as up to `max_segments` independent sub-lists of at least two ops each: seg_of[i] = segment of op i
// (segment 0 the longest; ops keep their relative order within a segment).  Components -- ops connected through a
// buffer one of them writes -- are dealt to the segments longest first, each to the segment that is shortest then.
// Returns the number of segments (1: the list does not split).
unsigned int pllhip_fused_segments(const FusedGeom & geom, const pllhip_op_t * ops, unsigned int count,
                                   unsigned int max_segments, std::vector<unsigned int> & seg_of);
// one plan per segment (pllhip_fused_plan of its sub-list)
int pllhip_launch_fused(pllhip_ctx * c, const std::vector<std::vector<FusedOp>> & plans, unsigned int nslots);
int pllhip_relaunch_fused(pllhip_ctx * c); // the same op list as in the previous whole-list call of this context

#endif
