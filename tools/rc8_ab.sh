cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/rc8
for v in 1 0; do
 for shape in "--taxa 64" "--taxa 200 --tree random --sites 100000" "--taxa 64 --tip-clv"; do
  (cd $R && PLLHIP_AA_RC8=$v python3 bench.py --states 20 --rate-cats 8 --sites 200000 $shape --cpu-sites 20000 --no-vary --no-c4 --steps 10) | tee -a $R/gpurun_out/rc8/lines_rc8_$v.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('RC8=$v', '$shape', d['value'], d['ms_per_step'], d.get('lnl_rel_err_vs_reference'), d['roofline'].get('kernel'))"
 done
done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rc8/trace -- python3 $R/bench.py --states 20 --rate-cats 8 --sites 200000 --cpu-sites 0 --no-vary --no-c4 --steps 10 > /dev/null 2> $R/gpurun_out/rc8/trace.err
python3 $R/tools/summarize_rocprof.py stats $R/gpurun_out/rc8/trace $R/gpurun_out/rc8/r4_bench_c3_gamma8_kernel_stats.csv
rm -rf $R/gpurun_out/rc8/trace
head -12 $R/gpurun_out/rc8/r4_bench_c3_gamma8_kernel_stats.csv
