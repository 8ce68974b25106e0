#!/bin/bash
# On the GPU box: duration of the derivative and final-sum kernels (rocprofv3 --kernel-trace --stats) for one or more builds
#   bash tools/deriv_kernel_time.sh libpll_amd/libpll_amd.so build/prev/libpll_amd.so
export TMPDIR=/tmp
for lib in "$@"; do
  rm -rf /tmp/dk
  PLL_AMD_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dk -- python3 bench.py --steps 5 --warmup 1 --cpu-sites 0 --no-c4 --no-vary --sites 500000 --taxa 200 --tree random --newton 20 > /dev/null 2>&1
  f=$(find /tmp/dk -name "*kernel_stats.csv" | head -1)
  echo "== $lib"
  [ -n "$f" ] && grep -E "k_derivatives|k_final_sum" "$f" | cut -d, -f1-4,6,7 | sed 's/^"\([^("]*\)[^"]*"/\1/'
done
