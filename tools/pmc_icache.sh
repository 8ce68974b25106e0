#!/bin/bash
# Instruction-cache counters of k_aa_fused (47 KB of code per instance): requests, hits, misses, waves' fetch stalls.
# On the GPU box:  bash tools/pmc_icache.sh   (counters in a pass of their own, the program directly behind "--")
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/icache; mkdir -p $out
rocprofv3 --list-avail 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_FETCH|SQ_WAIT_INST|SQ_INSTS_ISSUED|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_ACTIVE_INST|SQ_WAIT_ANY|SQ_INST_CYCLES" | sed 's/  */ /g' | cut -c1-160 | sort -u | head -40 > $out/avail.txt
cat $out/avail.txt
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $out/$name -- python3 $R/bench.py --states 20 --sites 200000 --steps 5 --warmup 1 --cpu-sites 0 --no-vary --no-c4 > /dev/null 2> $out/$name.err
  python3 $R/tools/summarize_rocprof.py pmc $out/$name $out/pmc_$name.csv "bench.py --states 20 --sites 200000" 2>/dev/null
  grep "k_aa_fused" $out/pmc_$name.csv | sed 's/.*unsigned int)",//' 
  rm -rf $out/$name
done
