#!/bin/bash
mkdir -p gpurun_out/r5v
timeout 900 python3 -m pytest tests/test_gpu_repeats.py -m gpu -x -q --durations=5 2>&1 | tail -15 > gpurun_out/r5v/tests.txt; cat gpurun_out/r5v/tests.txt
