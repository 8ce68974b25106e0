#!/bin/bash
# experiment: does a plan that exceeds the scalar cache slow the whole-list kernel down?
# build_pad/libpll_amd.so = the same library with 224-byte plan entries (make ... EXTRA_HIPFLAGS=-DPLLHIP_FUSED_PLAN_PAD=112)
out=gpurun_out/pad
mkdir -p $out
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
b c2_plain X=1 --
b c2_padded PLL_AMD_LIB=$PWD/build_pad/libpll_amd.so --
b c2_32taxa X=1 -- --taxa 32
b c4s_plain X=1 -- --taxa 128
b c4s_padded PLL_AMD_LIB=$PWD/build_pad/libpll_amd.so -- --taxa 128
b c4s_nopairs PLLHIP_FUSED_PAIRS=0 -- --taxa 128
b c2_nopairs PLLHIP_FUSED_PAIRS=0 --
b c4s_250k X=1 -- --taxa 128 --sites 250000
b c2_250k X=1 -- --sites 250000
b c256_500k X=1 -- --taxa 256 --sites 500000
