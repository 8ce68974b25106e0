#!/usr/bin/env python3
"""developer tool (round 6): the random op sequence of seed 3 (20 states, pattern tips, per-rate scale buffers) call by
call against the oracle: which op of which call goes wrong first."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
os.environ["PLLHIP_DEVELOPER"] = "1"
import numpy as np
import libpll_amd
from helpers import build_partition, random_sequence_case, oracle_run
from oracle_api import Oracle

amd = libpll_amd.load()
orc = Oracle(os.path.join(root, "oracle", "liboracle.so"))
os.environ["PLLHIP_FUSED"] = sys.argv[1] if len(sys.argv) > 1 else "2"
os.environ["PLLHIP_AA_TI_MFMA"] = sys.argv[2] if len(sys.argv) > 2 else "1"
os.environ["PLLHIP_FUSED_DEBUG"] = "1"
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 3
case, attrs, ops, rng = random_sequence_case(seed)
tips = case["plan"].tips
p = build_partition(amd, case, attrs)
o = oracle_run(orc, amd, p, case, attrs)
cut = sorted(int(x) for x in rng.integers(1, len(ops), size=5))
for lo, hi in zip([0] + cut, cut + [len(ops)]):
    if hi <= lo:
        continue
    p.update_partials(ops[lo:hi])
    o.update_partials(ops[lo:hi])
    # the final writer of each slot within this piece
    last = {}
    for k in range(lo, hi):
        last[int(ops[k]["parent_clv_index"])] = k
    for node, k in sorted(last.items(), key=lambda kv: kv[1]):
        a, b = p.get_clv(node), o.clv[node]
        sc = int(ops[k]["parent_scaler_index"])
        e = (np.abs(a - b) / np.maximum(np.abs(b), 1e-300)).max()
        cnt_ok = sc < 0 or (p.get_scaler(sc) == o.scalers[sc]).all()
        op = ops[k]
        kind = ("t" if int(op["child1_clv_index"]) < tips else "i") + ("t" if int(op["child2_clv_index"]) < tips else "i")
        if e > 1e-12 or not cnt_ok:
            i = np.unravel_index(np.argmax(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)), a.shape)
            print("piece [%d, %d) op %d (%s) slot %d children %d %d: rel %.3g at %s, counts %s, sc %d/%d/%d" % (
                lo, hi, k, kind, node, int(op["child1_clv_index"]), int(op["child2_clv_index"]), e, i, "ok" if cnt_ok else "DIFFER",
                sc, int(op["child1_scaler_index"]), int(op["child2_scaler_index"])))
            if not cnt_ok:
                d = np.nonzero(p.get_scaler(sc) != o.scalers[sc])[0]
                print("   count entries that differ:", d[:10], p.get_scaler(sc)[d[:10]], o.scalers[sc][d[:10]])
    print("piece [%d, %d) done" % (lo, hi))
