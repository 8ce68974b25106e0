#!/bin/bash
out=gpurun_out/dbg2
mkdir -p $out
T='tests/test_golden.py::test_hip_matches_golden[fused-dna_balanced16_tipclv_site]'
echo "== default (VGPR address form)"; python -m pytest "$T" -x -q 2>&1 | tail -2
echo "== SADDR form"; PLL_AMD_LIB=$PWD/build_drain/libpll_amd.so python -m pytest "$T" -x -q 2>&1 | tail -2
python -m pytest tests -m gpu -x -q 2>&1 | tail -8
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
b c4s X=1 -- --taxa 128
b c4s_nt0 PLLHIP_NT=0 -- --taxa 128
b c4s_static PLLHIP_FUSED_STATIC_TILES=1 -- --taxa 128
b c4s_random X=1 -- --taxa 128 --alignment random
b c2_random X=1 -- --alignment random
b c4s_500k X=1 -- --taxa 128 --sites 500000
b c4s_750k X=1 -- --taxa 128 --sites 750000
b tipclv X=1 -- --tip-clv
b tipclv_saddr PLL_AMD_LIB=$PWD/build_drain/libpll_amd.so -- --tip-clv
b c5s X=1 -- --sites 500000 --taxa 200 --tree random
export TMPDIR=/tmp
cd /tmp
rocprofv3 --list-avail > $OLDPWD/$out/counters.txt 2>&1
cd $OLDPWD
grep -i -c "name" $out/counters.txt
