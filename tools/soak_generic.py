#!/usr/bin/env python3
"""developer tool: random op sequences (tests/helpers.random_sequence_case) for the state counts
of partials_gen_tile.hip, HIP vs oracle bitwise.  python tools/soak_generic.py [first] [count]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import libpll_amd
from helpers import random_sequence_case, build_partition, oracle_run, bits_equal
from oracle_api import Oracle

amd = libpll_amd.load()
orc = Oracle(os.path.join(root, "oracle", "liboracle.so"))
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
SHAPES = [(2, 4), (3, 2), (5, 4), (5, 3), (7, 1), (8, 8), (11, 16), (13, 3), (16, 2), (21, 4), (32, 2), (48, 1), (61, 2)]
bad = 0
for seed in range(first, first + count):
    states, rc = SHAPES[seed % len(SHAPES)]
    case, attrs, ops, rng = random_sequence_case(seed, states, rc)
    plan = case["plan"]
    p = build_partition(amd, case, attrs)
    o = oracle_run(orc, amd, p, case, attrs)
    p.update_partials(ops)
    o.update_partials(ops)
    nodes = sorted(set(int(x) for x in ops["parent_clv_index"]))
    ok = all(bits_equal(p.get_clv(n), o.clv[n]) for n in nodes) and \
        all((p.get_scaler(sc) == o.scalers[sc]).all() for sc in range(plan.scale_buffers))
    if not ok:
        bad += 1
        print("MISMATCH seed %d states %d rate_cats %d attrs %#x" % (seed, states, rc, attrs))
    p.destroy()
print("soak_generic: %d seeds from %d, %d mismatches" % (count, first, bad))
