#!/usr/bin/env python3
"""developer tool: site-repeat identification (repeats.hip) at the sizes where its paths change -- merge sort up to
2^17 sites, Onesweep above; 32- and 64-bit keys; one pass or a carry in the prefix kernel -- on random trees with
alignments drawn from column pools of random size: every per-site lnL, the lnL and the top CLV + scale buffer of a
partition with the attribute bitwise against the same partition without it, then two subtree swaps (classes
identified again from the kept orders) and a tip replaced.   python tools/soak_repeats_at_size.py [first seed] [count]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import libpll_amd
from helpers import make_case, build_partition, bits_equal
from libpll_amd.pllapi import ATTRIB_PATTERN_TIP, ATTRIB_SITE_REPEATS, ATTRIB_RATE_SCALERS


def run(first, count, log=print, amd=None):
    amd = amd or libpll_amd.load()
    bad = 0
    for seed in range(first, first + count):
        rng = np.random.default_rng(31000 + seed)
        states = 20 if seed % 5 == 4 else 4
        shape = ("random", "balanced", "random", "caterpillar")[seed % 4]
        tips = int(2 ** rng.integers(3, 7)) if shape == "balanced" else int(rng.integers(6, 70 if states == 4 else 24))
        sites = int(rng.choice([rng.integers(100_000, 131_072), rng.integers(131_073, 300_000),
                                rng.integers(524_289, 800_000) if states == 4 else rng.integers(131_073, 200_000)]))
        attrs = ATTRIB_PATTERN_TIP | (ATTRIB_RATE_SCALERS if rng.random() < 0.25 else 0)
        case = make_case(states, shape, tips, sites, rate_cats=4, seed=seed, ambiguity=False, gap_frac=0.02)
        pool = rng.integers(0, sites, size=max(2, int(sites / rng.choice([3, 6, 20, 200, 5000]))))
        pick = pool[rng.integers(0, len(pool), size=sites)]
        case["seqs"] = [bytes(np.frombuffer(s, dtype=np.uint8)[pick]) for s in case["seqs"]]
        plan = case["plan"]
        os.environ["PLLHIP_AA_EXACT"] = "0"
        os.environ["PLLHIP_AA_TI_MFMA"] = "0"   # (plain against repeats bit for bit: the reference's order on both)
        plain = build_partition(amd, case, attrs)
        rep = build_partition(amd, case, attrs | ATTRIB_SITE_REPEATS)
        ops = plan.ops.copy()

        def same(what):
            for p in (plain, rep):
                p.update_partials(what)
            a = plain.compute_edge_loglikelihood(*plan.root_edge, [0] * 4, persite=True)
            b = rep.compute_edge_loglikelihood(*plan.root_edge, [0] * 4, persite=True)
            top, sc = int(ops[-1]["parent_clv_index"]), int(ops[-1]["parent_scaler_index"])
            return (a[0] == b[0] and bits_equal(a[1], b[1]) and bits_equal(plain.get_clv(top), rep.get_clv(top)) and
                    (sc < 0 or (plain.get_scaler(sc) == rep.get_scaler(sc)).all()))
        ok = same(ops)
        rows = [rep.repeats_classes(int(op["parent_clv_index"])) for op in ops]
        for step in range(2):
            if not ok or len(ops) < 6:
                break
            # exchange the second children of two ops that are not ancestors of each other: the last two ops below the
            # top that have inner second children
            cand = [k for k in range(len(ops) - 1) if int(ops[k]["child2_clv_index"]) >= plan.tips]
            if len(cand) < 2:
                break
            i, j = (int(x) for x in rng.choice(cand, size=2, replace=False))
            i, j = min(i, j), max(i, j)
            # j's subtree must not contain op i's parent, nor i's subtree op j's (else the swap makes a cycle)
            def below(k):
                seen, todo = set(), [int(ops[k]["parent_clv_index"])]
                by_parent = {int(o["parent_clv_index"]): o for o in ops}
                while todo:
                    n = todo.pop()
                    if n in seen:
                        continue
                    seen.add(n)
                    if n in by_parent:
                        todo += [int(by_parent[n]["child1_clv_index"]), int(by_parent[n]["child2_clv_index"])]
                return seen
            if int(ops[i]["parent_clv_index"]) in below(j) or int(ops[j]["parent_clv_index"]) in below(i):
                continue
            for f in ("child2_clv_index", "child2_matrix_index", "child2_scaler_index"):
                ops[i][f], ops[j][f] = ops[j][f], ops[i][f]
            # the swapped children must have been computed before their new parents: ops stay in a valid order only
            # if child(j) precedes op i; check, else undo
            pos = {int(o["parent_clv_index"]): k for k, o in enumerate(ops)}
            if any(int(ops[k][c]) in pos and pos[int(ops[k][c])] > k for k in (i, j) for c in ("child1_clv_index", "child2_clv_index")):
                for f in ("child2_clv_index", "child2_matrix_index", "child2_scaler_index"):
                    ops[i][f], ops[j][f] = ops[j][f], ops[i][f]
                continue
            ok = same(ops)
        if ok:
            new = bytes(np.frombuffer(case["seqs"][0], dtype=np.uint8)[::-1])
            for p in (plain, rep):
                p.set_tip_states(1, amd.map("nt" if states == 4 else "aa"), new)
            ok = same(ops)
        plain.destroy()
        rep.destroy()
        bad += 0 if ok else 1
        log("seed %d: %2d states %-11s %3d tips %7d sites, pool %7d, %2d of %2d ops by class (most rows %7d): %s"
            % (seed, states, shape, tips, sites, len(pool), sum(1 for r in rows if r), len(rows), max(rows), "ok" if ok else "MISMATCH"))
    log("%d seeds, %d mismatches" % (count, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 60) else 0)
