#!/bin/bash
export PLL_AMD_AUTO_MIRROR_MB=0   # (the device path is what is measured: no host mirrors kept for partitions below 64 MB)
export PLLHIP_DEVELOPER=1   # developer switches are honoured only under this one
# On the GPU box: interleaved A/B of one environment variable's values on bench.py shapes
#   bash tools/ab_env.sh PLLHIP_FUSED_TILE_GROUPS "1 8 96" "--sites 1000000 --taxa 64" ...
var=$1; vals=$2; shift 2
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-44s %-28s step %7.3f ms  launch %8.3f ms  frac %.3f  value %.1f' % ('$1', '$2', d['ms_per_step'], r['avg_launch_us']/1e3, r['frac'], d['value']))"; }
for shape in "$@"; do
  for rep in 1 2; do
    for v in $vals; do
      env $var=$v python3 bench.py $shape --cpu-sites 0 --steps 20 --warmup 2 --no-vary --no-c4 2>/dev/null | line "$shape" "$var=$v"
    done
  done
done
