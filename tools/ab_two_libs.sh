#!/bin/bash
# A/B of two builds of the library on ONE box: bash tools/ab_two_libs.sh <other .so> [bench args...]
other=$1; shift
for rep in 1 2; do
for lib in "" "$other"; do
  PLL_AMD_LIB=$lib python3 bench.py --cpu-sites 0 --steps 20 --warmup 3 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-30s %-22s launch_us %8.1f frac %.3f value %8.1f' % ('${lib:-this build}', '$*', r['avg_launch_us'], r['frac'], d['value']))"
done; done
