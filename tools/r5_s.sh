#!/bin/bash
mkdir -p gpurun_out/r5s
python3 -m pytest tests/test_gpu_aa_whole_list.py tests/test_gpu_thresholds.py -x -q > gpurun_out/r5s/tests.txt 2>&1; tail -3 gpurun_out/r5s/tests.txt
bash tools/ab_env.sh PLLHIP_AA_TI_MFMA "0 1" "--states 20 --sites 100000 --taxa 200 --tree random" "--states 20 --sites 200000 --taxa 64 --tree random" "--states 20 --sites 100000 --taxa 100 --tree caterpillar" "--states 20 --sites 200000" > gpurun_out/r5s/ti_mfma_ab.txt 2>&1; cat gpurun_out/r5s/ti_mfma_ab.txt
python3 tools/soak_aa_fused_at_size.py 90000 300 100000 200 20 > gpurun_out/r5s/soak.txt 2>&1; tail -1 gpurun_out/r5s/soak.txt
