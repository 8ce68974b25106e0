#!/bin/bash
mkdir -p gpurun_out/r5w
timeout 1500 python3 -m pytest tests/ -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r5w/tests.txt; cat gpurun_out/r5w/tests.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5w/smoke.txt 2>&1; tail -3 gpurun_out/r5w/smoke.txt
python3 bench.py > gpurun_out/r5w/bench.json 2> gpurun_out/r5w/bench.err; python3 -c "
import json; d=json.loads(open('gpurun_out/r5w/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['lnl_rel_err_vs_reference'])"
python3 bench.py --site-repeats --cpu-sites 0 --no-vary --no-c4 --steps 20 2>/dev/null | tail -1 > gpurun_out/r5w/bench_repeats_c2.json
python3 bench.py --site-repeats --tree random --taxa 200 --sites 500000 --cpu-sites 0 --no-vary --no-c4 --steps 20 2>/dev/null | tail -1 > gpurun_out/r5w/bench_repeats_c5.json
python3 -c "
import json
for f in ('c2','c5'):
    d=json.loads(open('gpurun_out/r5w/bench_repeats_%s.json'%f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['first_evaluation_ms'], d['config']['site_repeats'])"
