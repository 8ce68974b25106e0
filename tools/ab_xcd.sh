#!/bin/bash
out=gpurun_out/xcd
mkdir -p $out
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
b() {
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
for x in 1 16 64 256 1024 4096 1000000; do
b c2_xcd$x PLLHIP_FUSED_XCD=$x --
b c4s_xcd$x PLLHIP_FUSED_XCD=$x -- --taxa 128
b c5s_xcd$x PLLHIP_FUSED_XCD=$x -- --sites 500000 --taxa 200 --tree random
done
