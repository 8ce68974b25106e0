#!/bin/bash
export PLL_AMD_AUTO_MIRROR_MB=0   # (the device path is what is measured: no host mirrors kept for partitions below 64 MB)
export PLLHIP_DEVELOPER=1   # developer switches are honoured only under this one
# On the GPU box: the floor of a result-returning call (C5 shape: 500 k sites, 200-taxon random tree): how the
# sum is finished (PLLHIP_FUSE_REDUCE), how the host waits (PLLHIP_SPIN), the derivative kernel's grid and cache hint
for env in "PLLHIP_SPIN=1 PLLHIP_FUSE_REDUCE=0" "PLLHIP_SPIN=1 PLLHIP_FUSE_REDUCE=0 PLLHIP_DERIV_NT=1" "PLLHIP_SPIN=1 PLLHIP_FUSE_REDUCE=0 PLLHIP_DERIV_GRID=1024" "PLLHIP_SPIN=1 PLLHIP_FUSE_REDUCE=0 PLLHIP_DERIV_GRID=512" \
           "PLLHIP_SPIN=1 PLLHIP_FUSE_REDUCE=1 PLLHIP_DERIV_GRID=512" "PLLHIP_SPIN=1 PLLHIP_FUSE_REDUCE=1 PLLHIP_DERIV_GRID=256" "PLLHIP_SPIN=1 PLLHIP_FUSE_REDUCE=1" "PLLHIP_SPIN=0 PLLHIP_FUSE_REDUCE=0 PLLHIP_DERIV_NT=1"; do
  env $env python3 bench.py --steps 20 --warmup 2 --cpu-sites 0 --no-c4 --no-vary --sites 500000 --taxa 200 --tree random --newton 10 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-70s derivatives %.1f us/call, sumtable %.1f us, lnL call %.1f us wall' % ('$env', d['newton']['derivatives_us_per_call'], d['newton']['sumtable_us'], d['api_calls']['edge_loglikelihood_ms_wall']['median']*1e3))"
done
