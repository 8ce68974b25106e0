#!/usr/bin/env python3
"""developer tool: random op sequences (every kind of dependency between neighbouring ops, shared scale buffers, slots
written more than once) on 20-state partitions with a number of rate categories other than 1, 2, 4 -- the chunk
launches of partials_aa_mfma.hip on the DEFAULT path -- against the oracle, CLVs and scaler counts bit for bit.
python3 tools/soak_aa_chunks.py [first] [count]"""
import os, sys
os.environ.setdefault("PLLHIP_DEVELOPER", "1")  # the switches set below are developer's ones
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
os.environ.pop("PLLHIP_AA_EXACT", None)
import numpy as np
import libpll_amd
from helpers import random_sequence_case, build_partition, oracle_run, bits_equal
from oracle_api import Oracle

amd = libpll_amd.load()
orc = Oracle(os.path.join(root, "oracle", "liboracle.so"))
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
COUNTS = (3, 5, 6, 7, 8, 10, 12, 16)
bad = 0
for seed in range(first, first + count):
    R = COUNTS[seed % len(COUNTS)]
    os.environ["PLLHIP_AA_GRID_CAP"] = str(1 + seed % 5)
    case, attrs, ops, rng = random_sequence_case(seed, states=20, rate_cats=R)
    case["alpha"] = 0.3 + 0.1 * (seed % 7)
    plan = case["plan"]
    plan.branch_lengths = rng.uniform(0.05, 1.5, len(plan.matrix_indices))  # long branches: scaling events
    p = build_partition(amd, case, attrs)
    o = oracle_run(orc, amd, p, case, attrs)
    p.update_partials(ops)
    o.update_partials(ops)
    nodes = sorted(set(int(x) for x in ops["parent_clv_index"]))
    ok = all(bits_equal(p.get_clv(n), o.clv[n]) for n in nodes) and \
        all((p.get_scaler(sc) == o.scalers[sc]).all() for sc in range(plan.scale_buffers))
    events = int(sum(int(o.scalers[sc].sum()) for sc in range(plan.scale_buffers)))
    p.destroy()
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, "rate_cats", R, "attrs", attrs, flush=True)
    if (seed - first) % 50 == 49:
        print("... seed", seed, "mismatches so far", bad, "(scaling events in the last case:", events, ")", flush=True)
print("soak_aa_chunks: %d seeds from %d, %d mismatches" % (count, first, bad))
sys.exit(1 if bad else 0)
