#!/bin/bash
# several builds of the library on ONE box, interleaved: bash tools/ab_many_libs.sh "lib1 lib2 ..." [bench args...]
#   ("this" = the in-tree build)
libs=$1; shift
for rep in 1 2; do
for lib in $libs; do
  l=$lib; [ "$lib" = this ] && l=""
  PLL_AMD_LIB=$l python3 bench.py --cpu-sites 0 --steps 20 --warmup 3 --no-c4 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-28s %-28s launch_us %8.1f frac %.3f value %8.1f lnl %.6f' % ('$lib', '$*', r['avg_launch_us'], r['frac'], d['value'], d['lnl']))"
done; done
