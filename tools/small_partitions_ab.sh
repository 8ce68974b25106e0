#!/bin/bash
export PLL_AMD_AUTO_MIRROR_MB=0   # (the device path is what is measured: no host mirrors kept for partitions below 64 MB)
export PLLHIP_DEVELOPER=1   # developer switches are honoured only under this one
# small partitions (20 states by default; argument: 4): the whole-list kernel forced (PLLHIP_FUSED=2) against the per-level launches (=0): full
# traversals (the step) and the varying-list leg's partial traversals of 3 / 7 / 15 ops.  bash tools/small_partitions_ab.sh [states]  (SITES="2000 6000" to choose the sizes)
ST=${1:-20}
for shape in "--taxa 64" "--taxa 200 --tree random"; do
for sites in ${SITES:-2000 4000 6000 8000 12000 16000}; do
 for f in default 0 2; do
  if [ $f = default ]; then unset PLLHIP_FUSED; else export PLLHIP_FUSED=$f; fi
  python3 bench.py --states $ST --sites $sites $shape --cpu-sites 0 --no-c4 --steps 50 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=d['varying_lists']['ms_per_step']
print('$shape sites %6d FUSED=$f step %7.1f us  new full %7.1f  partial 3/7/15 ops %6.1f %6.1f %6.1f us' % ($sites, d['ms_per_step']*1e3, v['full traversal']['median']*1e3, *[v[k]['median']*1e3 for k in v if 'partial' in k]))"
 done
done; done
