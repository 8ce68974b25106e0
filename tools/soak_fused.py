#!/usr/bin/env python3
"""developer tool: the whole-list 4-state kernel (forced, PLLHIP_FUSED=2) on many random TREES -- shapes,
sizes, tip modes, scaling modes, rate categories -- as full traversals followed by partial traversals
after branch-length changes (operands written by earlier calls are then copied back from HBM by the
kernel's reload path), every CLV and scale buffer bitwise against the oracle.
python tools/soak_fused.py [first seed] [count]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
os.environ["PLLHIP_FUSED"] = "2"
import numpy as np
import libpll_amd
from helpers import make_case, build_partition, oracle_run, bits_equal
from oracle_api import Oracle
from libpll_amd.pllapi import ATTRIB_PATTERN_TIP, ATTRIB_RATE_SCALERS

amd = libpll_amd.load()
orc = Oracle(os.path.join(root, "oracle", "liboracle.so"))
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(7000 + seed)
    shape = ("random", "random", "balanced", "caterpillar")[seed % 4]
    tips = int(2 ** rng.integers(2, 8)) if shape == "balanced" else int(rng.integers(4, 140))
    sites = int(rng.integers(17, 400))
    rate_cats = int((1, 2, 4, 8)[seed % 4 if seed % 3 else 2])
    attrs = (ATTRIB_PATTERN_TIP if rng.random() < 0.7 else 0) | (ATTRIB_RATE_SCALERS if rng.random() < 0.3 else 0)
    case = make_case(4, shape, tips, sites, rate_cats=rate_cats, seed=seed)
    plan, R = case["plan"], rate_cats
    p = build_partition(amd, case, attrs)
    o = oracle_run(orc, amd, p, case, attrs)
    p.update_partials(plan.ops)
    o.update_partials()

    def same():
        return all(bits_equal(p.get_clv(int(op["parent_clv_index"])), o.clv[int(op["parent_clv_index"])]) and
                   (p.get_scaler(int(op["parent_scaler_index"])) == o.scalers[int(op["parent_scaler_index"])]).all()
                   for op in plan.ops)
    ok = same()
    for step in range(3):
        if not ok:
            break
        # change one branch below a random op, redo the ops on the path from there to the top
        k = int(rng.integers(0, len(plan.ops)))
        changed = int(plan.ops[k]["child1_matrix_index"])
        t = float(rng.uniform(0.01, 0.9))
        p.update_prob_matrices([0] * R, [changed], [t])
        slot = int(np.nonzero(plan.matrix_indices == changed)[0][0])
        plan.branch_lengths[slot] = t
        o.pmat[changed] = orc.pmatrix(4, R, o.m["rates"], t, o._ev, o._vc, o._iv, o._pinv)
        dirty, node = {int(plan.ops[k]["parent_clv_index"])}, int(plan.ops[k]["parent_clv_index"])
        while node in plan.parent_of and plan.parent_of[node] != node and plan.parent_of[node] not in dirty:
            node = plan.parent_of[node]
            dirty.add(node)
        sub = plan.ops[[int(op["parent_clv_index"]) in dirty for op in plan.ops]]
        p.update_partials(sub)
        o.update_partials(sub)
        ok = same()
    p.destroy()
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, shape, tips, sites, rate_cats, attrs, flush=True)
print("soak_fused: %d seeds from %d, %d mismatches" % (count, first, bad))
sys.exit(1 if bad else 0)
