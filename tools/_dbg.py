import os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
os.environ["PLLHIP_AA_EXACT"] = "0"; os.environ["PLLHIP_AA_CHERRY"] = "2"
import numpy as np, libpll_amd
from libpll_amd.pllapi import ATTRIB_PATTERN_TIP
from helpers import make_case, build_partition, oracle_run, bits_equal
from oracle_api import Oracle
lib = libpll_amd.load(); orc = Oracle(ROOT + "/oracle/liboracle.so")
for shape, tips, sites, attrs in (("balanced", 16, 333, ATTRIB_PATTERN_TIP), ("balanced", 8, 64, 0)):
    case = make_case(20, shape, tips, sites, seed=11)
    case["rates"], case["freqs"] = lib.aa_model("lg")
    plan = case["plan"]
    os.environ["PLLHIP_FUSED"] = "2"
    p = build_partition(lib, case, attrs)
    o = oracle_run(orc, lib, p, case, attrs); o.update_partials()
    p.update_partials(plan.ops)
    for k, op in enumerate(plan.ops):
        c = p.get_clv(int(op["parent_clv_index"])); r = o.clv[int(op["parent_clv_index"])]
        bad = np.argwhere(c.reshape(sites, -1) != r.reshape(sites, -1))
        sc = (p.get_scaler(int(op["parent_scaler_index"])) != o.scalers[int(op["parent_scaler_index"])]).sum()
        print(k, "children", int(op["child1_clv_index"]), int(op["child2_clv_index"]), "-> parent", int(op["parent_clv_index"]),
              "wrong entries", len(bad), "first", bad[:3].tolist(), "sites wrong", sorted(set(bad[:, 0].tolist()))[:12], "scaler diffs", sc)
    p.destroy()
