#!/bin/bash
export PLLHIP_DEVELOPER=1
export TMPDIR=/tmp
mkdir -p gpurun_out/r5ae
root=$(pwd)
{
bash tools/ab_env.sh PLLHIP_AA_TT_PAIRS "1 0" "--states 20 --sites 200000" "--states 20 --sites 100000 --taxa 200 --tree random"
# HBM traffic of C3 with the two-table form
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  PLLHIP_AA_TT_PAIRS=0 rocprofv3 --pmc $ctr --output-format csv -d /tmp/tt0_$ctr -- python3 $root/bench.py --steps 5 --warmup 1 --cpu-sites 0 --no-vary --no-c4 --states 20 --sites 200000 > /dev/null 2>&1
done
python3 $root/tools/summarize_rocprof.py hbm /tmp/tt0_FETCH_SIZE /tmp/tt0_WRITE_SIZE $root/gpurun_out/r5ae/pmc_hbm_traffic_c3_two_tip_tables.csv "PLLHIP_AA_TT_PAIRS=0 python3 bench.py --steps 5 --warmup 1 --cpu-sites 0 --no-vary --no-c4 --states 20 --sites 200000"
cat $root/gpurun_out/r5ae/pmc_hbm_traffic_c3_two_tip_tables.csv | head -12
} > gpurun_out/r5ae/tt_pairs_ab.txt 2>&1; cat gpurun_out/r5ae/tt_pairs_ab.txt
