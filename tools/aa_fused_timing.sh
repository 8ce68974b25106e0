#!/bin/bash
# Where do a wave's cycles go in the 20-state whole-list kernel?  A build of the library with cycle
# stamps between the phases of an op (-DPLLHIP_AF_TIMING: four waves print their averages per op kind).
#   bash tools/aa_fused_timing.sh build      (here: hipcc cross-compiles; build/ travels to the GPU box)
#   bash tools/aa_fused_timing.sh            (on the GPU box)
lib=build/aftiming/libpll_amd.so
if [ "$1" = build ]; then
  make -s -j8 BUILD=build/aftiming OUT=$lib EXTRA_HIPFLAGS=-DPLLHIP_AF_TIMING AGPR_CHECK_FLAGS="--allow-scratch --report-only" lib
  exit $?
fi
for cfg in "c3 --states 20 --taxa 64 --sites 200000" "c3_random --states 20 --taxa 64 --sites 200000 --tree random"; do
  set -- $cfg; tag=$1; shift
  echo "== $tag"
  PLL_AMD_LIB=$lib python3 bench.py --steps 2 --warmup 1 --cpu-sites 0 --no-c4 "$@" 2>/dev/null | grep -v '^{' | sort -k2,2n -s | awk '{c[$2]++; if (c[$2] <= 3) print}'
done
