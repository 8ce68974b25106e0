#!/bin/bash
export PLLHIP_DEVELOPER=1   # developer switches are honoured only under this one
# On the GPU box: partitions beyond 32 GB (BASELINE config 4 whole on one GPU: 8 M sites x 128 taxa, 133 GB;
# 2 M sites x 256 taxa, 67 GB) with the whole-list launch walking the alignment in blocks of sites.
#   bash tools/footprint_blocks.sh          timings per block size
#   bash tools/footprint_blocks.sh pmc      HBM traffic and address-translation counters of the 133 GB partition
root=$(pwd)
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-34s %-18s launch %8.2f ms  %6.1f GB/s  frac %.3f  lnL %.4f' % ('$1', '$2', r['avg_launch_us']/1e3, r['achieved'], r['frac'], d['lnl']))"; }
if [ "$1" != pmc ]; then
  for shape in "--total-sites 8000000 --taxa 128" "--total-sites 2000000 --taxa 256"; do
    for b in 0 4000000 2000000 1000000 500000 auto; do
      if [ $b = auto ]; then unset PLLHIP_FUSED_BLOCK_SITES; else export PLLHIP_FUSED_BLOCK_SITES=$b; fi
      python3 bench.py $shape --cpu-sites 0 --steps 5 --warmup 1 --no-vary 2>/dev/null | line "$shape" "block $b"
    done
  done
  exit 0
fi
out=$root/gpurun_out/footprint
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
# the 16 GB shard (1 M sites) beside the 133 GB whole (8 M sites): same list, same tiles per workgroup slot
for shape in "--sites 1000000" "--total-sites 8000000"; do
  for ctr in FETCH_SIZE WRITE_SIZE "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum" "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum" "TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum"; do
    tag=$(echo $shape $ctr | tr ' ' '_' | tr -d '-')
    rocprofv3 --pmc $ctr --output-format csv -d $out/$tag -- python3 $root/bench.py $shape --taxa 128 --cpu-sites 0 --steps 3 --warmup 1 --no-vary --no-c4 > /dev/null 2> $out/$tag.err
    python3 $root/tools/summarize_rocprof.py pmc $out/$tag $out/$tag.csv "python3 bench.py $shape --taxa 128 --cpu-sites 0 --steps 3 --warmup 1 --no-vary" 2>/dev/null
    grep -h "k_dna_fused" $out/$tag.csv | sed "s/^\"[^\"]*\"/$shape k_dna_fused/"
    rm -rf $out/$tag
  done
done
