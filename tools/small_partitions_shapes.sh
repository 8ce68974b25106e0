#!/bin/bash
export PLL_AMD_AUTO_MIRROR_MB=0   # (the device path is what is measured: no host mirrors kept for partitions below 64 MB)
# the small-partition rule on other tree shapes (random 64, balanced 128, a 100-taxon ladder), 20 and 4 states, 3,000 and 10,000
# sites: default choice against both paths forced.  bash tools/small_partitions_shapes.sh
for ST in 20 4; do
for shape in "--taxa 64 --tree random" "--taxa 128" "--taxa 100 --tree caterpillar"; do
for sites in 3000 10000; do
 for f in default 0 2; do
  if [ $f = default ]; then unset PLLHIP_FUSED; else export PLLHIP_FUSED=$f; fi
  python3 bench.py --states $ST --sites $sites $shape --cpu-sites 0 --no-c4 --steps 50 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=d['varying_lists']['ms_per_step']
print('states $ST $shape sites %6d FUSED=$f step %7.1f us  new full %7.1f  partial %s us' % ($sites, d['ms_per_step']*1e3, v['full traversal']['median']*1e3, ' '.join('%6.1f' % (v[k]['median']*1e3) for k in v if 'partial' in k)))"
 done
done; done; done
