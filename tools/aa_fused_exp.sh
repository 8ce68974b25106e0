#!/bin/bash
export PLLHIP_DEVELOPER=1   # developer switches are honoured only under this one
# On the GPU box: the 20-state list kernel with parts switched off (tool build -DPLLHIP_AF_TIMING -DPLLHIP_AF_NOTICKS,
# PLLHIP_AF_EXP bits: 1 no matrix-core products, 2 no stores, 4 no gathers, 8 no block staging, 16 no barriers):
# what the time of an update is made of.  Results are wrong by construction.
lib=build/afexp/libpll_amd.so
for m in 0 1 2 4 8 16 3 7 15 31; do
  PLLHIP_AF_EXP=$m PLL_AMD_LIB=$lib python3 bench.py --steps 10 --warmup 2 --cpu-sites 0 --no-c4 --states 20 --sites 200000 "$@" 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('mask %2d: update_partials %.3f ms' % ($m, d['api_calls']['update_partials_ms_hip_events']['median']))"
done
