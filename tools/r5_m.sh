#!/bin/bash
mkdir -p gpurun_out/r5m
python3 -m pytest tests/test_gpu_sharded.py tests/test_gpu_aa_whole_list.py -x -q > gpurun_out/r5m/tests.txt 2>&1; tail -3 gpurun_out/r5m/tests.txt
{
python3 tools/soak_aa_fused_at_size.py 40000 3000 100000 200 20
python3 tools/soak_aa_fused_at_size.py 50000 1500 100000 200 4
python3 tools/soak_aa_fused_at_size.py 60000 1500 40000 120 20
python3 tools/soak_aa_fused_at_size.py 70000 1500 30000 150 4
python3 tools/soak_aa_fused.py 20000 2000
python3 tools/soak_fused.py 9000 1500
python3 tools/soak.py 6000 400
} > gpurun_out/r5m/soaks.txt 2>&1; grep -v "^  \.\.\." gpurun_out/r5m/soaks.txt | tail -12
KERNEL=k_lnl_dna bash tools/pmc_instmix.sh --no-vary > gpurun_out/r5m/lnl_instmix.txt 2>&1; cat gpurun_out/r5m/lnl_instmix.txt
