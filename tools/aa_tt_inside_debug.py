#!/usr/bin/env python3
"""developer tool: hunt for an intermittent difference between the 20-state whole-list kernel and the per-level launches:
the same traversal repeated on the same and on fresh partitions, every CLV compared each time.
python tools/aa_tt_inside_debug.py TAXA SITES REPEATS"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
os.environ["PLLHIP_AA_EXACT"] = "0"
import numpy as np, libpll_amd
from libpll_amd import workload as W
from libpll_amd.pllapi import ATTRIB_PATTERN_TIP
from helpers import bits_equal
amd = libpll_amd.load()
T, sites, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
plan = W.random_tree(T, seed=42)
rates, freqs = amd.aa_model("lg")
seqs = W.simulated_alignment(plan, sites, rates, freqs, amd.compute_gamma_cats(W.GAMMA_ALPHA, 4), seed=42)
kind = lambda op: "tt" if op["child1_clv_index"] < T and op["child2_clv_index"] < T else "ti" if (op["child1_clv_index"] < T or op["child2_clv_index"] < T) else "ii"
writer = {int(op["parent_clv_index"]): i for i, op in enumerate(plan.ops)}
os.environ["PLLHIP_FUSED"] = "0"
p = W.setup_partition(amd, plan, seqs, 20, 4, ATTRIB_PATTERN_TIP)
p.update_partials(plan.ops)
ref = [p.get_clv(int(op["parent_clv_index"])) for op in plan.ops]
refs = [p.get_scaler(int(op["parent_scaler_index"])) for op in plan.ops]
p.destroy()
os.environ["PLLHIP_FUSED"] = "1"
bad_runs = 0
for rep in range(reps):
    p = W.setup_partition(amd, plan, seqs, 20, 4, ATTRIB_PATTERN_TIP)
    for again in range(3):
        p.update_partials(plan.ops)
        nbad = 0
        for i, op in enumerate(plan.ops):
            a = p.get_clv(int(op["parent_clv_index"]))
            s = p.get_scaler(int(op["parent_scaler_index"]))
            if not bits_equal(a, ref[i]) or not (s == refs[i]).all():
                d = np.argwhere(a.reshape(sites, -1) != ref[i].reshape(sites, -1))
                rows = np.unique(d[:, 0])
                kids = [kind(plan.ops[writer[int(c)]]) if int(c) in writer else "tip" for c in (op["child1_clv_index"], op["child2_clv_index"])]
                if nbad < 6:
                    print("partition %d call %d: op %d (%s over %s): %d sites differ %s, entries per site %s, scaler diff %d" %
                          (rep, again, i, kind(op), kids, len(rows), rows[:8], np.bincount(d[:, 0])[rows[:4]] if len(rows) else "", int((s != refs[i]).sum())))
                nbad += 1
        bad_runs += nbad > 0
print("evaluations with a difference: %d of %d" % (bad_runs, reps * 3))
