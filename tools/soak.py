#!/usr/bin/env python3
"""developer tool: many more seeds of the random op-sequence checks than the test suite
runs (HIP vs oracle bitwise; site repeats vs plain bitwise).  python tools/soak.py [first] [count]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
os.environ["PLLHIP_AA_EXACT"] = "1"
import numpy as np
import libpll_amd
from helpers import random_sequence_case, build_partition, oracle_run, bits_equal
from oracle_api import Oracle
from libpll_amd.pllapi import ATTRIB_PATTERN_TIP, ATTRIB_SITE_REPEATS

amd = libpll_amd.load()
orc = Oracle(os.path.join(root, "oracle", "liboracle.so"))
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
for seed in range(first, first + count):
    case, attrs, ops, rng = random_sequence_case(seed)
    plan = case["plan"]
    p = build_partition(amd, case, attrs)
    o = oracle_run(orc, amd, p, case, attrs)
    p.update_partials(ops)
    o.update_partials(ops)
    nodes = sorted(set(int(x) for x in ops["parent_clv_index"]))
    ok = all(bits_equal(p.get_clv(n), o.clv[n]) for n in nodes) and \
        all((p.get_scaler(sc) == o.scalers[sc]).all() for sc in range(plan.scale_buffers))
    if ok and case["states"] == 4 and (attrs & ATTRIB_PATTERN_TIP):
        sites = case["sites"]
        pool = rng.integers(0, sites, size=sites // 8 + 1)
        pick = pool[rng.integers(0, len(pool), size=sites)]
        case["seqs"] = [bytes(np.frombuffer(s, dtype=np.uint8)[pick]) for s in case["seqs"]]
        a = build_partition(amd, case, attrs)
        b = build_partition(amd, case, attrs | ATTRIB_SITE_REPEATS)
        a.update_partials(ops); b.update_partials(ops)
        ok = all(bits_equal(a.get_clv(n), b.get_clv(n)) for n in nodes)
        a.destroy(); b.destroy()
    p.destroy()
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, "states", case["states"], "attrs", attrs, flush=True)
print("soak: %d seeds from %d, %d mismatches" % (count, first, bad))
sys.exit(1 if bad else 0)
