#!/bin/bash
# Where do a wave's cycles go in the whole-list kernel?  A build of the library with s_memtime
# stamps between the phases of an op (-DPLLHIP_FUSED_TIMING: four waves print their averages).
#   bash tools/fused_timing.sh build      (here: hipcc cross-compiles; build/ travels to the GPU box)
#   bash tools/fused_timing.sh            (on the GPU box)
lib=build/timing/libpll_amd.so
if [ "$1" = build ]; then
  make -s BUILD=build/timing OUT=$lib EXTRA_HIPFLAGS=-DPLLHIP_FUSED_TIMING lib
  exit $?
fi
for cfg in "c2 --taxa 64 --sites 1000000" "c2_noscale --taxa 64 --sites 1000000 --no-scalers" "c4s --taxa 128 --sites 1000000" "tipclv --taxa 64 --sites 500000 --tip-clv"; do
  set -- $cfg; tag=$1; shift
  echo "== $tag"
  PLL_AMD_LIB=$lib python3 bench.py --steps 2 --warmup 1 --cpu-sites 0 --no-c4 "$@" 2>/dev/null | grep -v '^{' | sed 's/^wave [0-9]* //' | sort | uniq -c | sort -rn | head -6
done
