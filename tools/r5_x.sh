#!/bin/bash
mkdir -p gpurun_out/r5x
timeout 1500 python3 tools/soak_repeats_at_size.py 0 120 > gpurun_out/r5x/soak_repeats.txt 2>&1; tail -30 gpurun_out/r5x/soak_repeats.txt
