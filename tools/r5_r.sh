#!/bin/bash
export TMPDIR=/tmp; root=$(pwd); cd /tmp; mkdir -p $root/gpurun_out/r5r
for sites in 20000 100000; do for lib in "" build/ab_pipe/libpll_amd.so; do
PLL_AMD_LIB=${lib:+$root/$lib} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -- python3 $root/bench.py --steps 50 --warmup 2 --cpu-sites 0 --no-vary --no-c4 --sites $sites > /dev/null 2>&1
echo "== $sites sites, ${lib:-this build}"; python3 $root/tools/kernel_stats.py /tmp/kp 3 | grep lnl; rm -rf /tmp/kp
done; done > $root/gpurun_out/r5r/lnl_small.txt 2>&1; cat $root/gpurun_out/r5r/lnl_small.txt
