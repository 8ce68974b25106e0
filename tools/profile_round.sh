#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root:  bash tools/profile_round.sh <tag>
# Takes the bench line and the rocprofv3 evidence for the two headline workloads
# (4-state C2, 20-state C3) into gpurun_out/<tag>/; tools/summarize_rocprof.py then
# condenses those directories into profiles/.  Counters are collected in their own
# passes (no trace domains next to --pmc).
set -u
tag=${1:-r1}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp

run() { # name, bench args...
  local name=$1; shift
  (cd "$root" && python3 bench.py "$@" > "$out/bench_$name.json" 2> "$out/bench_$name.err")
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_$name" -- \
      python3 "$root/bench.py" --steps 20 --warmup 3 --cpu-sites 0 "$@" > "$out/bench_${name}_under_rocprof.json" 2> "$out/trace_$name.err"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch_$name" -- \
      python3 "$root/bench.py" --steps 5 --warmup 1 --cpu-sites 0 "$@" > /dev/null 2> "$out/fetch_$name.err"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write_$name" -- \
      python3 "$root/bench.py" --steps 5 --warmup 1 --cpu-sites 0 "$@" > /dev/null 2> "$out/write_$name.err"
  # keep what travels back small: the per-dispatch trace is not needed
  find "$out/trace_$name" -name '*_kernel_trace.csv' -delete
}

run c2
run c3 --states 20 --sites 200000
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$out/mfma_c3" -- \
    python3 "$root/bench.py" --steps 5 --warmup 1 --cpu-sites 0 --states 20 --sites 200000 > /dev/null 2> "$out/mfma_c3.err"
ls "$out"
# BASELINE config 5's shape (200-taxon random tree, 500 k sites) with the Newton inner loop
(cd "$root" && python3 bench.py --sites 500000 --taxa 200 --tree random --newton 5 --cpu-sites 0 > "$out/bench_c5shape.json" 2> "$out/bench_c5shape.err")
