#!/usr/bin/env python3
"""developer tool: the 20-state whole-list kernel (k_aa_fused, PLLHIP_FUSED=2) on many random TREES -- shapes, sizes,
tip modes, with and without scale buffers -- as full traversals followed by partial traversals after branch-length
changes, every CLV and scale buffer bitwise against the per-level launches (PLLHIP_FUSED=0), whose inner-inner
CLVs and counts the test suite pins to the oracle.
python tools/soak_aa_fused.py [first seed] [count]
Round 6: PLLHIP_AA_TI_MFMA unset or 1 in the environment = the default path -- the whole-list kernel's tip-inner mat-vecs
on the matrix cores: scale buffers bit for bit (the scaling certificate), CLVs to 1e-12 entry by entry, lnL to 1e-12, no
uncertified scaling decision; PLLHIP_AA_TI_MFMA=0 = every CLV bit for bit, as until round 5."""
import os, sys
os.environ.setdefault("PLLHIP_DEVELOPER", "1")  # the switches set below are developer's ones
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
os.environ["PLLHIP_AA_EXACT"] = "0"
import numpy as np
import libpll_amd
from helpers import make_case, build_partition, bits_equal, clv_err
from libpll_amd.pllapi import ATTRIB_PATTERN_TIP, SCALE_BUFFER_NONE

amd = libpll_amd.load()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rates, freqs = amd.aa_model("lg")
loose = os.environ.get("PLLHIP_AA_TI_MFMA", "1") != "0"
bad = raised = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(9000 + seed)
    shape = ("random", "random", "balanced", "caterpillar")[seed % 4]
    tips = int(2 ** rng.integers(2, 8)) if shape == "balanced" else int(rng.integers(4, 160))
    sites = int(rng.integers(1, 700))
    attrs = ATTRIB_PATTERN_TIP if rng.random() < 0.7 else 0
    case = make_case(20, shape, tips, sites, seed=seed)
    case["rates"], case["freqs"] = rates, freqs
    plan = case["plan"]
    ops = plan.ops.copy()
    if rng.random() < 0.2:
        for f in ("parent_scaler_index", "child1_scaler_index", "child2_scaler_index"):
            ops[f] = SCALE_BUFFER_NONE
    os.environ["PLLHIP_AA_CHERRY"] = "2" if rng.random() < 0.7 else "0"
    # partial traversals: a few random suffixes of the list after a branch-length change
    cuts = [int(rng.integers(1, len(ops) + 1)) for _ in range(3)]
    lens = [float(rng.uniform(0.01, 1.5)) for _ in cuts]
    out = {}
    for fused in ("2", "0"):
        os.environ["PLLHIP_FUSED"] = fused
        p = build_partition(amd, case, attrs)
        p.update_partials(ops)
        for n, t in zip(cuts, lens):
            p.update_prob_matrices([0] * 4, [int(ops[-n]["child1_matrix_index"])], [t])
            p.update_partials(ops[-n:])
        lnl = p.compute_edge_loglikelihood(*plan.root_edge, [0] * 4)
        out[fused] = (lnl, [p.get_clv(int(op["parent_clv_index"])) for op in ops],
                      [p.get_scaler(int(op["parent_scaler_index"])) if int(op["parent_scaler_index"]) >= 0 else None for op in ops],
                      p.scaling_certificate())
        p.destroy()
    a, b = out["2"], out["0"]
    raised += a[3]["raised"]
    if loose:
        ok = abs(a[0] - b[0]) <= 1e-12 * abs(b[0]) and all(clv_err(x, y) <= 1e-12 for x, y in zip(a[1], b[1])) and \
            all((x is None and y is None) or (x == y).all() for x, y in zip(a[2], b[2])) and a[3]["uncertified"] == 0
    else:
        ok = a[0] == b[0] and all(bits_equal(x, y) for x, y in zip(a[1], b[1])) and \
            all((x is None and y is None) or (x == y).all() for x, y in zip(a[2], b[2]))
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, shape, tips, sites, attrs, os.environ["PLLHIP_AA_CHERRY"], cuts)
print("soak_aa_fused: %d seeds from %d, %s, %d mismatches, %d certificate flags raised" % (
    count, first, "default path (tip-inner mat-vecs on the matrix cores)" if loose else "PLLHIP_AA_TI_MFMA=0", bad, raised))
sys.exit(1 if bad else 0)
