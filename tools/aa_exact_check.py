#!/usr/bin/env python3
"""developer tool: which 20-state ops of the DEFAULT (matrix-core) path are bit-identical to the oracle.
Runs random / balanced / caterpillar trees with tips as characters and as CLVs, both scaling modes."""
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
from libpll_amd.pllapi import ATTRIB_PATTERN_TIP, ATTRIB_RATE_SCALERS
from helpers import make_case, build_partition, oracle_run, bits_equal
from oracle_api import Oracle

lib = libpll_amd.load()
orc = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
bad = 0
for shape, tips, sites in (("random", 12, 500), ("balanced", 16, 333), ("caterpillar", 40, 100), ("balanced", 64, 40000)):
    for attrs in (0, ATTRIB_PATTERN_TIP, ATTRIB_PATTERN_TIP | ATTRIB_RATE_SCALERS, ATTRIB_RATE_SCALERS):
        case = make_case(20, shape, tips, sites, seed=7)
        case["rates"], case["freqs"] = lib.aa_model("lg")
        p = build_partition(lib, case, attrs)
        o = oracle_run(orc, lib, p, case, attrs)
        plan = case["plan"]
        p.update_partials(plan.ops)
        o.update_partials()
        kinds = {"ii": [0, 0], "ti": [0, 0], "tt": [0, 0]}
        for op in plan.ops:
            node, sc = int(op["parent_clv_index"]), int(op["parent_scaler_index"])
            t1 = (attrs & ATTRIB_PATTERN_TIP) and int(op["child1_clv_index"]) < tips
            t2 = (attrs & ATTRIB_PATTERN_TIP) and int(op["child2_clv_index"]) < tips
            kind = "tt" if (t1 and t2) else "ti" if (t1 or t2) else "ii"
            ok = bits_equal(p.get_clv(node), o.clv[node])
            kinds[kind][0] += 1
            kinds[kind][1] += 0 if ok else 1
            assert (p.get_scaler(sc) == o.scalers[sc]).all()
        lnl = p.compute_edge_loglikelihood(*plan.root_edge, [0] * 4)
        ref = o.edge_loglikelihood(*plan.root_edge)
        print("%-11s %3d tips %6d sites attrs %3d: " % (shape, tips, sites, attrs) +
              "  ".join("%s %d ops, %d differ" % (k, v[0], v[1]) for k, v in kinds.items()) +
              "  lnL rel err %.2e" % abs((lnl - ref) / ref))
        bad += kinds["ii"][1] + kinds["tt"][1]
        p.destroy()
print("ii/tt ops that differ from the oracle:", bad)
sys.exit(1 if bad else 0)
