#!/bin/bash
# The whole-list kernel on the BASELINE shapes (bench.py lines without the CPU baseline).  bash tools/ab_fused.sh <tag> [env...]
tag=${1:-ab}; shift
out=gpurun_out/$tag
mkdir -p $out
b() { # name, -- args
  local name=$1; shift; shift
  env "${ENVS[@]}" python3 bench.py --cpu-sites 0 --steps 20 --warmup 3 "$@" > $out/$name.json 2> $out/$name.err
  python3 - "$out/$name.json" "$name" <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%-22s value %9.1f  ms/step %7.3f  launch_us %9.1f  frac %.3f  lnl %.6f" % (sys.argv[2], d["value"], d["ms_per_step"], r["avg_launch_us"], r["frac"], d["lnl"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
}
ENVS=("X=1" "$@")
b c2 --
b c4s -- --taxa 128
b c5s -- --sites 500000 --taxa 200 --tree random
b tipclv -- --tip-clv
b ratesc -- --rate-scalers --taxa 128
b c2_100k -- --sites 100000
b c256 -- --taxa 256 --sites 500000
