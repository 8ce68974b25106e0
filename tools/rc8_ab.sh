#!/bin/bash
export PLLHIP_DEVELOPER=1   # developer switches are honoured only under this one
# 20 states x R rate categories (R != 1, 2, 4): chunk launches of the matrix-core kernels (default) against the
# all-vector kernels (PLLHIP_AA_CHUNKS=0), three shapes, one box; then the kernel statistics of the first shape.
# Run through gpurun from the repo root:  bash tools/rc8_ab.sh [R]
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
RC=${1:-8}
out=$R/gpurun_out/rc$RC
mkdir -p $out
rm -f $out/lines_*.json
for v in 1 0; do
 for shape in "--taxa 64" "--taxa 200 --tree random --sites 100000" "--taxa 64 --tip-clv"; do
  (cd $R && PLLHIP_AA_CHUNKS=$v python3 bench.py --states 20 --rate-cats $RC --sites 200000 $shape --cpu-sites 20000 --no-vary --no-c4 --steps 10 2>/dev/null) | tee -a $out/lines_chunks_$v.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('CHUNKS=$v', 'R=$RC', '$shape', d['ms_per_step'], d.get('lnl_rel_err_vs_reference'), d['roofline']['frac'], d['kernels'])"
 done
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/bench.py --states 20 --rate-cats $RC --sites 200000 --cpu-sites 0 --no-vary --no-c4 --steps 10 > /dev/null 2> $out/trace.err
python3 $R/tools/summarize_rocprof.py stats $out/trace $out/r4_bench_c3_rates${RC}_kernel_stats.csv
rm -rf $out/trace
head -12 $out/r4_bench_c3_rates${RC}_kernel_stats.csv | cut -c1-150
