#!/bin/bash
# On the GPU box: interleaved A/B of two builds of the library on bench.py shapes (boxes and processes differ by
# several per cent, so only interleaved runs on one box compare).
#   bash tools/ab_libs.sh build/prev/libpll_amd.so libpll_amd/libpll_amd.so "--sites 1000000 --taxa 64" "--states 20 --sites 200000" ...
a=$1; b=$2; shift 2
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-44s %-28s step %7.3f ms  launch %8.3f ms  frac %.3f  value %.1f' % ('$1', '$2', d['ms_per_step'], r['avg_launch_us']/1e3, r['frac'], d['value']))"; }
for shape in "$@"; do
  for rep in 1 2 3; do
    for lib in $a $b; do
      PLL_AMD_LIB=$lib python3 bench.py $shape --cpu-sites 0 --steps 20 --warmup 2 --no-vary --no-c4 2>/dev/null | line "$shape" $lib
    done
  done
done
