#!/bin/bash
mkdir -p gpurun_out/r5af
timeout 900 python3 tools/soak_aa_fused_at_size.py 500000 2500 100000 200 20 > gpurun_out/r5af/soak_final_aa_fused_at_size.log 2>&1; tail -2 gpurun_out/r5af/soak_final_aa_fused_at_size.log
timeout 600 python3 tools/soak_aa_fused_at_size.py 600000 1500 100000 200 4 > gpurun_out/r5af/soak_final_dna_fused_at_size.log 2>&1; tail -2 gpurun_out/r5af/soak_final_dna_fused_at_size.log
timeout 600 python3 tools/soak_aa_fused_at_size.py 700000 1500 30000 64 20 > gpurun_out/r5af/soak_final_aa_fused_segments_size.log 2>&1; tail -2 gpurun_out/r5af/soak_final_aa_fused_segments_size.log
timeout 600 python3 tools/soak_repeats_at_size.py 1000 300 > gpurun_out/r5af/soak_final_repeats_at_size.log 2>&1; tail -1 gpurun_out/r5af/soak_final_repeats_at_size.log
