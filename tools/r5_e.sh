#!/bin/bash
export PLLHIP_DEVELOPER=1   # developer switches are honoured only under this one
mkdir -p gpurun_out/r5e
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_core_api.py tests/test_gpu_thresholds.py -x -q -k "not full_size" > gpurun_out/r5e/tests.txt 2>&1; tail -3 gpurun_out/r5e/tests.txt
bash tools/segments_ab.sh "20 12500" "20 25000" "20 50000" "20 100000" "20 200000" > gpurun_out/r5e/segments_ab_20.txt 2>&1; cat gpurun_out/r5e/segments_ab_20.txt
{
echo "== C2: this build against round 4's (build/ab_head)"; bash tools/ab_two_libs.sh build/ab_head/libpll_amd.so --no-vary --no-c4
echo "== C3"; bash tools/ab_two_libs.sh build/ab_head/libpll_amd.so --no-vary --no-c4 --states 20 --sites 200000
echo "== random 200 x 100k, 20 states"; bash tools/ab_two_libs.sh build/ab_head/libpll_amd.so --no-vary --no-c4 --states 20 --sites 100000 --taxa 200 --tree random
} > gpurun_out/r5e/ab_r4.txt 2>&1; cat gpurun_out/r5e/ab_r4.txt
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-52s derivatives %.1f us/call, sumtable %.1f us, lnL call %.1f us wall, lnl kernel %s' % ('$1', d['newton']['derivatives_us_per_call'], d['newton']['sumtable_us'], d['api_calls']['edge_loglikelihood_ms_wall']['median']*1e3, d['kernels'].get('lnl')))"; }
{
for rep in 1 2; do
for lib in "" build/ab_head/libpll_amd.so; do
  PLL_AMD_LIB=$lib python3 bench.py --steps 20 --warmup 2 --cpu-sites 0 --no-c4 --no-vary --sites 500000 --taxa 200 --tree random --newton 20 2>/dev/null | line "C5 shape ${lib:-this build}"
  PLL_AMD_LIB=$lib python3 bench.py --steps 20 --warmup 2 --cpu-sites 0 --no-c4 --no-vary 2>/dev/null --newton 20 | line "C2 ${lib:-this build}"
done
for g in 1024 2048 4096; do
  PLLHIP_LNL_GRID=$g python3 bench.py --steps 20 --warmup 2 --cpu-sites 0 --no-c4 --no-vary --newton 5 2>/dev/null | line "C2 PLLHIP_LNL_GRID=$g"
done
done
} > gpurun_out/r5e/result_calls.txt 2>&1; cat gpurun_out/r5e/result_calls.txt
bash tools/deriv_kernel_time.sh libpll_amd/libpll_amd.so build/ab_head/libpll_amd.so > gpurun_out/r5e/deriv_kernel.txt 2>&1; cat gpurun_out/r5e/deriv_kernel.txt
