#!/bin/bash
# developer tool (round 6): what each kind of job costs k_af_prepare (PLLHIP_AF_PREP_SKIP, wrong results): kernel trace
# of BASELINE config 3 with all jobs / without one kind at a time / with one kind only.   bash tools/aa_prepare_parts.sh
export PLL_AMD_AUTO_MIRROR_MB=0 PLLHIP_DEVELOPER=1 TMPDIR=/tmp
cd "$(dirname "$0")/.." || exit 1
for skip in 0 1 4 8 12 9 5 13 15; do
  rm -rf /tmp/prep_parts
  PLLHIP_AF_PREP_SKIP=$skip rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prep_parts -o p -- python3 bench.py --states 20 --sites 200000 --cpu-sites 0 --no-vary --no-c4 --steps 30 > /dev/null 2>&1
  f=$(find /tmp/prep_parts -name "*kernel_stats.csv" | head -1)
  python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'k_af_prepare' in r['Name']:
        print('skip %2d (1 matrices, 4 pair tables, 8 lookup tables): k_af_prepare average %6.1f us  min %6.1f' % ($skip, float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))"
done
