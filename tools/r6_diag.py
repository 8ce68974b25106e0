#!/usr/bin/env python3
"""developer tool (round 6): where the default 20-state path's CLVs differ from the reference-order path's by more than
1e-13 relative -- the entries' sizes against their block's largest."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
os.environ["PLLHIP_DEVELOPER"] = "1"
import numpy as np
import libpll_amd
from helpers import make_case, build_partition, random_sequence_case
from libpll_amd.pllapi import ATTRIB_PATTERN_TIP

amd = libpll_amd.load()


def show(a, b, what):
    a = np.asarray(a); b = np.asarray(b)
    err = np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
    i = np.unravel_index(np.argmax(err), err.shape)
    blockmax = np.abs(b[i[0]]).max()
    print("%s: worst relative %.3g at %s: %.17g vs %.17g, block max %.3g, worst |a-b|/block max %.3g" % (
        what, err[i], i, a[i], b[i], blockmax, (np.abs(a - b) / np.abs(b).max(axis=(1, 2), keepdims=True).clip(1e-300)).max()))


os.environ["PLLHIP_FUSED"] = "2"
case = make_case(20, "caterpillar", 300, 300, seed=5)
case["rates"], case["freqs"] = amd.aa_model("lg")
plan = case["plan"]
res = {}
for ti in ("0", "1"):
    os.environ["PLLHIP_AA_TI_MFMA"] = ti
    os.environ["PLLHIP_AA_GRID_CAP"] = "1"
    p = build_partition(amd, case, ATTRIB_PATTERN_TIP)
    p.update_partials(plan.ops)
    res[ti] = [p.get_clv(int(op["parent_clv_index"])) for op in plan.ops], [p.get_scaler(int(op["parent_scaler_index"])) for op in plan.ops]
    print("cert", p.scaling_certificate())
    p.destroy()
os.environ.pop("PLLHIP_AA_GRID_CAP")
worst = 0
for k, (a, b) in enumerate(zip(res["1"][0], res["0"][0])):
    e = (np.abs(a - b) / np.maximum(np.abs(b), 1e-300)).max()
    if e > 1e-13:
        show(a, b, "caterpillar 300, op %d" % k)
        break
print("scalers equal:", all((x == y).all() for x, y in zip(res["1"][1], res["0"][1])))

case, attrs, ops, rng = random_sequence_case(3)
print("random sequence case 3: states", case["states"], "ops", len(ops))
res = {}
for ti in ("0", "1"):
    os.environ["PLLHIP_AA_TI_MFMA"] = ti
    p = build_partition(amd, case, attrs)
    p.update_partials(ops)
    res[ti] = {n: p.get_clv(n) for n in sorted(set(int(x) for x in ops["parent_clv_index"]))}
    p.destroy()
for n in res["0"]:
    e = (np.abs(res["1"][n] - res["0"][n]) / np.maximum(np.abs(res["0"][n]), 1e-300)).max()
    if e > 1e-13:
        show(res["1"][n], res["0"][n], "random sequence, slot %d" % n)

# in pieces
os.environ["PLLHIP_AA_TI_MFMA"] = "1"
p = build_partition(amd, case, attrs)
p.update_partials(ops)
p2 = build_partition(amd, case, attrs)
cut = sorted(int(x) for x in rng.integers(1, len(ops), size=5))
print("cuts", cut)
for lo, hi in zip([0] + cut, cut + [len(ops)]):
    if hi > lo:
        p2.update_partials(ops[lo:hi])
for n in res["0"]:
    for name, q in (("whole", p), ("pieces", p2)):
        a = q.get_clv(n)
        e = (np.abs(a - res["0"][n]) / np.maximum(np.abs(res["0"][n]), 1e-300)).max()
        if e > 1e-13:
            show(a, res["0"][n], "random sequence %s vs reference order, slot %d" % (name, n))
print(p.scaling_certificate(), p2.scaling_certificate())

# the soak's seed 31100
from libpll_amd import workload as W
T, sites, R = 200, 100_000, 4
rates, freqs = amd.aa_model("lg")
plan0 = W.random_tree(T, seed=42)
seqs = W.simulated_alignment(plan0, sites, rates, freqs, amd.compute_gamma_cats(W.GAMMA_ALPHA, R), seed=42)
parts = {}
for fused in ("2", "0"):
    os.environ["PLLHIP_FUSED"] = fused
    parts[fused] = W.setup_partition(amd, plan0, seqs, 20, R, ATTRIB_PATTERN_TIP)
seed = 31100
plan = W.random_tree(T, seed=1000 + seed)
rng = np.random.default_rng(seed)
n = int(rng.integers(1, len(plan.ops)))
t_new = float(rng.uniform(0.01, 1.2))
out = {}
for fused, q in parts.items():
    q.update_prob_matrices([0] * R, plan.matrix_indices, plan.branch_lengths)
    q.update_partials(plan.ops)
    q.update_partials(plan.ops)
    q.update_prob_matrices([0] * R, [int(plan.ops[-n]["child1_matrix_index"])], [t_new])
    q.update_partials(plan.ops[-n:])
    out[fused] = [q.get_clv(int(op["parent_clv_index"])) for op in plan.ops[-3:]]
for a, b in zip(out["2"], out["0"]):
    show(a, b, "soak seed 31100 (n = %d)" % n)
