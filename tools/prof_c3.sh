#!/bin/bash
# On the GPU box: kernel statistics of the C3 bench (20 states, 200 k sites) -> gpurun_out/<tag>_c3_kernel_stats.csv
tag=${1:-r3}
root=$(pwd)
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/trace_$tag" -- \
  python3 "$root/bench.py" --steps 20 --warmup 3 --cpu-sites 0 --states 20 --sites 200000 > "$root/gpurun_out/${tag}_c3_under_rocprof.json" 2> "$root/gpurun_out/trace_$tag.err"
python3 "$root/tools/summarize_rocprof.py" stats "$root/gpurun_out/trace_$tag" "$root/gpurun_out/${tag}_c3_kernel_stats.csv"
rm -rf "$root/gpurun_out/trace_$tag"
cat "$root/gpurun_out/${tag}_c3_kernel_stats.csv"
