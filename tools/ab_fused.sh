#!/bin/bash
# A/B of the whole-list kernel's plans on the GPU box: reload plan (LDS-DMA) against the EXT plan
# (operands without a slot into registers) on the BASELINE shapes.  bash tools/ab_fused.sh <tag>
tag=${1:-ab}
out=gpurun_out/$tag
mkdir -p $out
b() { # name, env..., -- args
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" python3 bench.py --cpu-sites 0 --steps 20 --warmup 3 "$@" > $out/$name.json 2> $out/$name.err
  python3 - "$out/$name.json" "$name" <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%-28s value %9.1f  ms/step %7.3f  launch_us %9.1f  frac %.3f  lnl %.6f" % (sys.argv[2], d["value"], d["ms_per_step"], r["avg_launch_us"], r["frac"], d["lnl"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
}
b c2_reload PLLHIP_FUSED_RELOAD=1 --
b c2_ext PLLHIP_FUSED_RELOAD=0 --
b c4s_reload PLLHIP_FUSED_RELOAD=1 -- --taxa 128
b c4s_ext PLLHIP_FUSED_RELOAD=0 -- --taxa 128
b c5s_reload PLLHIP_FUSED_RELOAD=1 -- --sites 500000 --taxa 200 --tree random
b c5s_ext PLLHIP_FUSED_RELOAD=0 -- --sites 500000 --taxa 200 --tree random
b tipclv_reload PLLHIP_FUSED_RELOAD=1 -- --tip-clv
b tipclv_ext PLLHIP_FUSED_RELOAD=0 -- --tip-clv
b ratesc_reload PLLHIP_FUSED_RELOAD=1 -- --rate-scalers --taxa 128
b ratesc_ext PLLHIP_FUSED_RELOAD=0 -- --rate-scalers --taxa 128
