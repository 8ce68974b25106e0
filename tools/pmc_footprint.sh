#!/bin/bash
# counters of the whole-list kernel on the 128-taxon list at 500 k sites (8 GB of CLVs, fast)
# and 1 M sites (16 GB, slow): what differs?   bash tools/pmc_footprint.sh
root=$(pwd)
out=$root/gpurun_out/pmcfp
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
pass() { # tag, sites, counters...
  local tag=$1 sites=$2; shift 2
  rocprofv3 --pmc "$@" --output-format csv -d $out/${tag}_$sites -- python3 $root/bench.py --steps 4 --warmup 1 --cpu-sites 0 --taxa 128 --sites $sites > /dev/null 2> $out/${tag}_$sites.err
  python3 $root/tools/summarize_rocprof.py pmc $out/${tag}_$sites $out/${tag}_$sites.csv
  grep k_dna_fused $out/${tag}_$sites.csv | sed "s/^\"[^\"]*\"/$sites/"
  rm -rf $out/${tag}_$sites
}
for sites in 500000 1000000; do
  pass tlb $sites TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE
  pass lat $sites TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
  pass tcc $sites TCC_HIT_sum TCC_MISS_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum
  pass fetch $sites FETCH_SIZE
  pass write $sites WRITE_SIZE
  pass sqc $sites SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM
  pass sq $sites SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS
  pass tlb2 $sites TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum
done
