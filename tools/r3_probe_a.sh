#!/bin/bash
# one GPU call: (1) counters of the 16 GB and 133 GB partitions, (2) the 133 GB alignment as eight partitions of
# 16.6 GB on the SAME device (address locality per launch), (3) short op lists, whole list vs per level + phase stamps
bash tools/footprint_blocks.sh pmc > gpurun_out/r3_footprint_pmc.txt 2>&1
{
echo "== 8 M sites x 128 taxa as ONE partition"
python3 bench.py --total-sites 8000000 --taxa 128 --cpu-sites 0 --steps 5 --warmup 1 --no-vary --no-c4 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['lnl'])"
echo "== the same as eight site ranges on device 0 (PLL_AMD_DEVICES=0,0,0,0,0,0,0,0)"
python3 bench.py --gpus 8 --in-process --devices 0,0,0,0,0,0,0,0 --total-sites 8000000 --taxa 128 --cpu-sites 0 --steps 5 --warmup 1 --no-vary --no-c4 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['lnl'])"
echo "== four ranges"
python3 bench.py --gpus 4 --in-process --devices 0,0,0,0 --total-sites 8000000 --taxa 128 --cpu-sites 0 --steps 5 --warmup 1 --no-vary --no-c4 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['lnl'])"
} > gpurun_out/r3_same_device_ranges.txt 2>&1
{
for f in 2 0; do PLLHIP_FUSED=$f python3 tools/partial_traversal_timing.py 1000000; done
echo "== phase stamps (timing build), whole list forced"
PLL_AMD_LIB=build/timing/libpll_amd.so PLLHIP_FUSED=2 python3 tools/partial_traversal_timing.py 1000000 | sed 's/^wave [0-9]* //' | sort | uniq -c | sort -rn | head -40
} > gpurun_out/r3_short_lists.txt 2>&1
tail -30 gpurun_out/r3_footprint_pmc.txt; cat gpurun_out/r3_same_device_ranges.txt gpurun_out/r3_short_lists.txt
