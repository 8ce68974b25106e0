#!/bin/bash
# developer tool (round 6): k_dna_pair_tables -- the small launch ahead of every 4-state whole-list launch -- and the
# step at the sizes where it is a sixth of a call.  bash tools/dna_pair_tables_ab.sh [out dir] [another build of the library]
export PLL_AMD_AUTO_MIRROR_MB=0
export PLLHIP_DEVELOPER=1
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=${1:-gpurun_out/pair_ab}
export PLL_AMD_LIB=${2:-}
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/c2" -o c2 -- python3 bench.py --cpu-sites 0 --no-vary --no-c4 --steps 50 > "$out/c2.json" 2> "$out/c2.err"
f=$(find "$out/c2" -name "*kernel_stats.csv" | head -1)
python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'k_dna_pair_tables' in r['Name'] or 'k_dna_fused' in r['Name']:
        print('  %-18s calls %5s  average %9.1f us  min %9.1f  max %9.1f' % ('k_dna_pair_tables' if 'pair' in r['Name'] else 'k_dna_fused', r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))"
rm -rf "$out/c2"
for cfg in "4 20000" "4 31250" "4 50000" "4 62500" "4 125000"; do
  set -- $cfg
  python3 bench.py --states $1 --sites $2 --cpu-sites 0 --no-vary --no-c4 --steps 50 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; a=d['api_calls']
print('states %2d sites %7d  step %8.1f us  update_partials %8.1f us (events)  lnl call %6.1f us  frac %.3f  value %.1f' % ($1, $2, d['ms_per_step']*1e3, a['update_partials_ms_hip_events']['median']*1e3, a['edge_loglikelihood_ms_wall']['median']*1e3, r['frac'], d['value']))"
done
