#!/bin/bash
# developer tool (round 6): the 4-state whole-list kernel with FOUR workgroups per CU (sixteen waves of 128 registers --
# the variant spills five -- and four LDS slots per wave instead of six: more operands copied back from HBM) against
# the three it runs with, in a build that has both (-DPLLHIP_FUSED_WPS4; round 3 measured it on lists of 2-15 ops only):
#   make -j8 BUILD=build/wps4 OUT=build/wps4/libpll_amd.so EXTRA_HIPFLAGS=-DPLLHIP_FUSED_WPS4 lib     (here)
#   bash tools/fused_wps4.sh                                                                          (on the GPU box)
export PLL_AMD_AUTO_MIRROR_MB=0 PLLHIP_DEVELOPER=1
cd "$(dirname "$0")/.." || exit 1
export PLL_AMD_LIB=$PWD/build/wps4/libpll_amd.so
for shape in "--sites 1000000" "--sites 1000000 --taxa 128" "--sites 500000 --taxa 200 --tree random" "--sites 125000"; do
  for rep in 1 2; do
    for w in 3 4; do
      PLLHIP_FUSED_WGS=$w python3 bench.py $shape --cpu-sites 20000 --steps 20 --warmup 2 --no-vary --no-c4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-44s workgroups per CU %d  step %7.3f ms  launch %8.1f us  frac %.3f  value %8.1f  lnL err %s  %s' % ('$shape', $w, d['ms_per_step'], r['avg_launch_us'], r['frac'], d['value'], d.get('lnl_rel_err_vs_reference'), r['kernel'][:40]))"
    done
  done
done
