#!/bin/bash
# On the GPU box: the whole-list test files, then the kernel statistics of three 20-state shapes (k_af_prepare beside
# k_aa_fused).  bash tools/prep_time.sh
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd $R && timeout 600 python -m pytest tests/test_gpu_aa_whole_list.py tests/test_golden.py -q -x 2>&1 | grep -E "passed|failed|rror" | tail -2)
for shape in "--sites 200000" "--sites 100000 --taxa 200 --tree random" "--sites 30000"; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prep_trace -- python3 $R/bench.py --states 20 $shape --cpu-sites 0 --no-vary --no-c4 --steps 20 > /tmp/line.json 2> /dev/null
python3 $R/tools/summarize_rocprof.py stats $R/gpurun_out/prep_trace /tmp/stats.csv
echo "== $shape"; head -4 /tmp/stats.csv | cut -c1-60,200-400 | sed 's/  */ /g'; python3 -c "
import csv
for r in list(csv.reader(open('/tmp/stats.csv')))[1:5]: print(r[0][:40], r[1], float(r[3])/1e3)"
rm -rf $R/gpurun_out/prep_trace
done
