#!/bin/bash
# On the GPU box: a partition the library shards over "eight devices" (ordinal 0 eight times) of ONE GPU -- what the
# host side of the in-process mode costs per shard: one enqueueing thread per shard against the calling thread
# visiting the shards in turn (PLLHIP_SHARD_THREADS), results polled in host-mapped memory against a
# hipStreamSynchronize per shard (PLLHIP_SHARD_POLL).   bash tools/shards_ab.sh
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-58s %-44s step %8.1f us' % ('$1', '$2', d['ms_per_step']*1e3))"; }
for shape in "--total-sites 1000000" "--total-sites 200000" "--total-sites 500000 --taxa 200 --tree random" "--total-sites 200000 --states 20"; do
for rep in 1 2; do
for env in "PLLHIP_SHARD_THREADS=0 PLLHIP_SHARD_POLL=0" "PLLHIP_SHARD_THREADS=0 PLLHIP_SHARD_POLL=1" "PLLHIP_SHARD_THREADS=1 PLLHIP_SHARD_POLL=1"; do
  env $env python3 bench.py --gpus 8 --in-process --devices 0,0,0,0,0,0,0,0 $shape --cpu-sites 0 --no-vary --no-c4 --steps 30 2>/dev/null | line "8 shards on one device: $shape" "$env"
done
done
python3 bench.py $(echo $shape | sed 's/--total-sites/--sites/') --cpu-sites 0 --no-vary --no-c4 --steps 30 2>/dev/null | line "one partition: $shape" ""
done
