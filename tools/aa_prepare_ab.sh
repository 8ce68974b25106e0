#!/bin/bash
# developer tool (round 6): what k_af_prepare -- the tables and the matrices in operand order, one small launch ahead of
# every 20-state whole-list launch -- costs, and what the list costs at the sizes where that launch is 10-15 % of it.
#   bash tools/aa_prepare_ab.sh [out dir] [another build of the library]     (prints the kernel's average from a kernel trace of BASELINE config 3 and of the
#                                           200-taxon random tree, then the 20-state lines of tools/size_sweep2.sh)
export PLL_AMD_AUTO_MIRROR_MB=0
export PLLHIP_DEVELOPER=1
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=${1:-gpurun_out/prep_ab}
mkdir -p "$out"
lib=${2:-}
export PLL_AMD_LIB=$lib
for w in "c3 --states 20 --sites 200000" "random200 --states 20 --sites 100000 --tree random --taxa 200"; do
  set -- $w; tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$tag" -o "$tag" -- python3 bench.py "$@" --cpu-sites 0 --no-vary --no-c4 --steps 50 > "$out/$tag.json" 2> "$out/$tag.err"
  f=$(find "$out/$tag" -name "*kernel_stats.csv" | head -1)
  echo "== $tag"; python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'k_af_prepare' in r['Name'] or 'k_aa_fused' in r['Name']:
        print('  %-14s calls %5s  average %9.1f us  min %9.1f  max %9.1f' % ('k_af_prepare' if 'k_af_prepare' in r['Name'] else 'k_aa_fused', r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))"
done
for cfg in "20 12500" "20 25000" "20 50000" "20 100000"; do
  set -- $cfg
  python3 bench.py --states $1 --sites $2 --cpu-sites 0 --no-vary --no-c4 --steps 50 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; a=d['api_calls']
print('states %2d sites %7d  step %8.1f us  update_partials %8.1f us (events)  lnl call %6.1f us  frac %.3f  value %.1f' % ($1, $2, d['ms_per_step']*1e3, a['update_partials_ms_hip_events']['median']*1e3, a['edge_loglikelihood_ms_wall']['median']*1e3, r['frac'], d['value']))"
done
