#!/bin/bash
mkdir -p gpurun_out/r5l
python3 -m pytest tests/test_gpu_aa_whole_list.py tests/test_gpu_sharded.py tests/test_gpu_thresholds.py tests/test_gpu_baseline_configs.py -x -q -k "not config4" > gpurun_out/r5l/tests.txt 2>&1; tail -3 gpurun_out/r5l/tests.txt
{
echo "== C3: this build (k_af_prepare: four first characters per workgroup) against round 4's library"; bash tools/ab_two_libs.sh build/ab_head/libpll_amd.so --no-vary --no-c4 --states 20 --sites 200000
echo "== 20 states, 12,500 sites"; bash tools/ab_two_libs.sh build/ab_head/libpll_amd.so --no-vary --no-c4 --states 20 --sites 12500
echo "== 20 states, 200 taxa random, 12,000 sites"; bash tools/ab_two_libs.sh build/ab_head/libpll_amd.so --no-vary --no-c4 --states 20 --sites 12000 --taxa 200 --tree random
} > gpurun_out/r5l/ab_prepare.txt 2>&1; cat gpurun_out/r5l/ab_prepare.txt
export TMPDIR=/tmp; root=$(pwd); cd /tmp
for lib in "" build/ab_head/libpll_amd.so; do
PLL_AMD_LIB=${lib:+$root/$lib} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -- python3 $root/bench.py --steps 20 --warmup 2 --cpu-sites 0 --no-vary --no-c4 --states 20 --sites 200000 > /dev/null 2>&1
echo "== kernels, C3, ${lib:-this build}"; python3 $root/tools/kernel_stats.py /tmp/kp 4; rm -rf /tmp/kp
done > $root/gpurun_out/r5l/prepare_kernels.txt 2>&1; cat $root/gpurun_out/r5l/prepare_kernels.txt
