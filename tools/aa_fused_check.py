#!/usr/bin/env python3
"""developer tool: the 20-state whole-list kernel (PLLHIP_FUSED=2) against the per-level launches
(PLLHIP_FUSED=0) and the oracle: every CLV and scale buffer bit for bit."""
import os
os.environ.setdefault("PLLHIP_DEVELOPER", "1")  # the switches set below are developer's ones
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["PLLHIP_AA_EXACT"] = "0"
os.environ.setdefault("PLLHIP_AA_CHERRY", "2")
import numpy as np
import libpll_amd
from libpll_amd.pllapi import ATTRIB_PATTERN_TIP
from helpers import make_case, build_partition, oracle_run, bits_equal
from oracle_api import Oracle

lib = libpll_amd.load()
orc = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
bad = 0
cases = [("balanced", 16, 333), ("balanced", 64, 1000), ("random", 12, 500), ("caterpillar", 40, 100),
         ("random", 50, 97), ("balanced", 128, 64), ("random", 30, 1), ("balanced", 8, 40000), ("random", 200, 40),
         ("caterpillar", 120, 33), ("random", 200, 20000)]
if len(sys.argv) > 1:
    cases = cases[:int(sys.argv[1])] if int(sys.argv[1]) > 0 else cases[int(sys.argv[1]):]
for shape, tips, sites in cases:
    for attrs in (ATTRIB_PATTERN_TIP, 0):
        case = make_case(20, shape, tips, sites, seed=11)
        case["rates"], case["freqs"] = lib.aa_model("lg")
        plan = case["plan"]
        res = {}
        for mode in ("2", "0"):
            os.environ["PLLHIP_FUSED"] = mode
            p = build_partition(lib, case, attrs)
            p.update_partials(plan.ops)
            # a partial traversal on top: the last few ops again (operands from the earlier call)
            p.update_partials(plan.ops[-3:])
            p.update_partials(plan.ops[-1:])
            lnl = p.compute_edge_loglikelihood(*plan.root_edge, [0] * 4)
            res[mode] = ([p.get_clv(int(op["parent_clv_index"])) for op in plan.ops],
                         [p.get_scaler(int(op["parent_scaler_index"])) for op in plan.ops], lnl)
            if mode == "0":
                o = oracle_run(orc, lib, p, case, attrs)
                o.update_partials()
                exact = sum(bits_equal(c, o.clv[int(op["parent_clv_index"])]) for c, op in zip(res["2"][0], plan.ops))
            p.destroy()
        nclv = sum(not bits_equal(a, b) for a, b in zip(res["2"][0], res["0"][0]))
        nsc = sum(not (a == b).all() for a, b in zip(res["2"][1], res["0"][1]))
        print("%-11s %3d tips %6d sites attrs %2d: %3d ops, CLVs differing from the per-level path %d, scalers %d; "
              "equal to the oracle %d; lnL %.6f / %.6f" % (shape, tips, sites, attrs, len(plan.ops), nclv, nsc, exact,
                                                           res["2"][2], res["0"][2]), flush=True)
        bad += nclv + nsc
print("differences:", bad)
sys.exit(1 if bad else 0)
