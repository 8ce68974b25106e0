"""Site repeats (PLL_ATTRIB_SITE_REPEATS, host/repeats.c -- an extension, the reference
snapshot has none): every observable result must equal the one obtained without the
attribute, bit for bit.  CLVs and scale buffers are compared after expansion to one row
per site (what the host mirrors show), per-site lnL, lnL, sumtable and derivatives
directly."""
import numpy as np
import pytest

from helpers import make_case, build_partition, bits_equal, oracle_run, rel_err
from libpll_amd import workload as W
from libpll_amd.pllapi import (ATTRIB_PATTERN_TIP, ATTRIB_RATE_SCALERS, ATTRIB_SITE_REPEATS,
                               ATTRIB_AB_LEWIS, OPS_DTYPE, SCALE_BUFFER_NONE, PllError)

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _reference_order_tip_inner(monkeypatch):
    """This file compares two PATHS of the library bit for bit (site repeats against the plain partition).  On the default path the 20-state whole-list
    kernel runs tip-inner mat-vecs on the matrix cores (round 6: CLVs to 1e-15 per op, scaler counts bit for bit behind
    the scaling certificate -- tests/test_gpu_cert.py, tests/test_gpu_aa_whole_list.py), so a path that takes that
    kernel and one that does not agree to rounding only; pinned to the reference's order here."""
    monkeypatch.setenv("PLLHIP_AA_TI_MFMA", "0")


def evaluate(p, plan, R):
    p.update_partials(plan.ops)
    e = plan.root_edge
    lnl, ps = p.compute_edge_loglikelihood(*e, [0] * R, persite=True)
    st = p.alloc_sumtable()
    p.update_sumtable(e[0], e[2], e[1], e[3], [0] * R, st)
    d = [p.compute_likelihood_derivatives(e[1], e[3], t, [0] * R, st) for t in (0.05, 0.7)]
    return lnl, ps, p.get_sumtable(st), d


def same_state(p, q, plan):
    for op in plan.ops:
        node, sc = int(op["parent_clv_index"]), int(op["parent_scaler_index"])
        assert bits_equal(p.get_clv(node), q.get_clv(node)), "CLV %d" % node
        if sc >= 0:
            assert (p.get_scaler(sc) == q.get_scaler(sc)).all(), "scaler %d" % sc


@pytest.mark.parametrize("states,shape,tips,sites,rate_cats,rate_scalers",
                         [(4, "balanced", 16, 3000, 4, False), (4, "random", 30, 2111, 4, False),
                          (4, "caterpillar", 40, 1500, 4, False), (4, "random", 24, 1777, 2, True),
                          (4, "balanced", 8, 65, 1, False), (4, "random", 50, 4000, 8, False),
                          (20, "balanced", 16, 2000, 4, False), (20, "random", 25, 1555, 4, True),
                          (20, "caterpillar", 20, 777, 2, False), (20, "random", 12, 333, 1, False)])
def test_repeats_equal_plain(gpu, monkeypatch, states, shape, tips, sites, rate_cats, rate_scalers):
    monkeypatch.setenv("PLLHIP_AA_EXACT", "0")   # 20 states: the matrix-core kernels follow row maps
    case = make_case(states, shape, tips, sites, rate_cats=rate_cats, seed=tips + sites, gap_frac=0.02)
    # few distinct columns near the tips, like real data: draw sites from a small pool
    rng = np.random.default_rng(sites)
    pool = rng.integers(0, sites, size=sites // 6 + 1)
    pick = pool[rng.integers(0, len(pool), size=sites)]
    case["seqs"] = [bytes(np.frombuffer(s, dtype=np.uint8)[pick]) for s in case["seqs"]]
    plan, R = case["plan"], rate_cats
    attrs = ATTRIB_PATTERN_TIP | (ATTRIB_RATE_SCALERS if rate_scalers else 0)
    plain = build_partition(gpu, case, attrs)
    rep = build_partition(gpu, case, attrs | ATTRIB_SITE_REPEATS)
    a = evaluate(plain, plan, R)
    b = evaluate(rep, plan, R)
    compressed = [rep.repeats_classes(int(op["parent_clv_index"])) for op in plan.ops]
    assert sum(1 for c in compressed if c) >= len(plan.ops) // 2, compressed
    assert all(c <= sites // 2 for c in compressed)
    same_state(plain, rep, plan)
    assert a[0] == b[0] and bits_equal(a[1], b[1])
    assert bits_equal(a[2], b[2])
    assert a[3] == b[3]
    # the root form on a CLV stored by class
    top = int(plan.ops[-1]["parent_clv_index"])
    ra = plain.compute_root_loglikelihood(top, int(plan.ops[-1]["parent_scaler_index"]), [0] * R, persite=True)
    rb = rep.compute_root_loglikelihood(top, int(plan.ops[-1]["parent_scaler_index"]), [0] * R, persite=True)
    assert ra[0] == rb[0] and bits_equal(ra[1], rb[1])
    # a second evaluation reuses the classes (nothing changed) and gives the same bits
    b2 = evaluate(rep, plan, R)
    assert b2[0] == b[0] and bits_equal(b2[1], b[1])
    # branch-length change + partial traversal: classes stay, values follow
    changed = int(plan.ops[0]["parent_clv_index"])
    slot = int(np.nonzero(plan.matrix_indices == changed)[0][0])
    plan.branch_lengths[slot] = 0.37
    for p in (plain, rep):
        p.update_prob_matrices([0] * R, [changed], [0.37])
        dirty, node = set(), changed
        while node in plan.parent_of and plan.parent_of[node] != node and plan.parent_of[node] not in dirty:
            dirty.add(plan.parent_of[node])
            node = plan.parent_of[node]
        p.update_partials(plan.ops[[int(op["parent_clv_index"]) in dirty for op in plan.ops]])
    same_state(plain, rep, plan)
    la = plain.compute_edge_loglikelihood(*plan.root_edge, [0] * R)
    lb = rep.compute_edge_loglikelihood(*plan.root_edge, [0] * R)
    assert la == lb and la != a[0]
    plain.destroy()
    rep.destroy()


@pytest.mark.parametrize("states,shape,tips,sites,rate_cats,rate_scalers",
                         [(4, "balanced", 16, 1200, 4, False), (4, "random", 30, 911, 4, True),
                          (4, "caterpillar", 60, 700, 4, False), (20, "random", 14, 400, 4, False),
                          (20, "balanced", 8, 300, 2, True)])
def test_repeats_against_the_oracle(gpu, orc, monkeypatch, states, shape, tips, sites, rate_cats, rate_scalers):
    """Not a self-comparison: the partition WITH site repeats against liboracle.so directly
    (which knows nothing of repeats) -- expanded CLVs and scale buffers of every op bitwise,
    per-site lnL to 1e-13, lnL, sumtable and derivatives to the tolerances of the plain tests."""
    monkeypatch.delenv("PLLHIP_AA_EXACT", raising=False)   # (the default path: CLVs are bit for bit there too)
    case = make_case(states, shape, tips, sites, rate_cats=rate_cats, seed=3 * tips + sites, gap_frac=0.02)
    rng = np.random.default_rng(sites + 1)
    pool = rng.integers(0, sites, size=sites // 5 + 1)
    pick = pool[rng.integers(0, len(pool), size=sites)]
    case["seqs"] = [bytes(np.frombuffer(s, dtype=np.uint8)[pick]) for s in case["seqs"]]
    plan, R = case["plan"], rate_cats
    attrs = ATTRIB_PATTERN_TIP | (ATTRIB_RATE_SCALERS if rate_scalers else 0)
    rep = build_partition(gpu, case, attrs | ATTRIB_SITE_REPEATS)
    o = oracle_run(orc, gpu, rep, case, attrs)
    rep.update_partials(plan.ops)
    o.update_partials()
    stored_by_class = 0
    for op in plan.ops:
        node, sc = int(op["parent_clv_index"]), int(op["parent_scaler_index"])
        stored_by_class += 1 if rep.repeats_classes(node) else 0
        assert bits_equal(rep.get_clv(node), o.clv[node]), "CLV %d" % node
        assert (rep.get_scaler(sc) == o.scalers[sc]).all(), "scaler %d" % sc
    if states == 4:
        assert stored_by_class >= len(plan.ops) // 2   # (20 states, exact kernels: stored per site)
    e = plan.root_edge
    lnl, ps = rep.compute_edge_loglikelihood(*e, [0] * R, persite=True)
    ref_lnl, ref_ps = o.edge_loglikelihood(*e, persite=True)
    assert rel_err(ps, ref_ps) < 1e-13
    assert abs(lnl - ref_lnl) <= 1e-12 * abs(ref_lnl)
    st = rep.alloc_sumtable()
    rep.update_sumtable(e[0], e[2], e[1], e[3], [0] * R, st)
    want = o.sumtable(e[0], e[2], e[1], e[3])
    got = rep.get_sumtable(st).reshape(want.shape)
    scale = np.abs(want).max(axis=2, keepdims=True) + 1e-300
    assert float(np.max(np.abs(got - want) / scale)) < 1e-12
    for t in (0.05, 0.7):
        d = rep.compute_likelihood_derivatives(e[1], e[3], t, [0] * R, st)
        assert rel_err(np.array(d), np.array(o.derivatives(want, t))) < 1e-10
    rep.destroy()


def test_repeats_follow_topology_and_tip_changes(gpu):
    """Classes are rebuilt when an op pairs different children into a slot, and when a
    tip's sequence is replaced."""
    case = make_case(4, "balanced", 8, 900, seed=4, gap_frac=0.0, ambiguity=False)
    rng = np.random.default_rng(1)
    pool = rng.integers(0, 900, size=60)
    pick = pool[rng.integers(0, len(pool), size=900)]
    case["seqs"] = [bytes(np.frombuffer(s, dtype=np.uint8)[pick]) for s in case["seqs"]]
    plan, T = case["plan"], 8
    plain = build_partition(gpu, case, ATTRIB_PATTERN_TIP)
    rep = build_partition(gpu, case, ATTRIB_PATTERN_TIP | ATTRIB_SITE_REPEATS)
    for p in (plain, rep):
        p.update_partials(plan.ops)
    same_state(plain, rep, plan)
    # re-pair the tips: (0,2) and (1,3) into the slots that held (0,1) and (2,3)
    ops = plan.ops.copy()
    ops[0]["child2_clv_index"], ops[1]["child1_clv_index"] = 2, 1
    ops[0]["child2_matrix_index"], ops[1]["child1_matrix_index"] = 2, 1
    for p in (plain, rep):
        p.update_partials(ops)
    same_state(plain, rep, plan)
    # new sequence for tip 5
    new = bytes(np.frombuffer(case["seqs"][3], dtype=np.uint8)[::-1])
    for p in (plain, rep):
        p.set_tip_states(5, gpu.map("nt"), new)
        p.update_partials(ops)
    same_state(plain, rep, plan)
    a = plain.compute_edge_loglikelihood(*plan.root_edge, [0] * 4)
    b = rep.compute_edge_loglikelihood(*plan.root_edge, [0] * 4)
    assert a == b
    plain.destroy()
    rep.destroy()


def test_repeats_attribute_contract(gpu, monkeypatch):
    odd = make_case(4, "balanced", 8, 50, seed=3)
    odd["states"] = 5
    with pytest.raises(PllError):
        gpu.partition_create(8, 6, 5, 50, 1, 14, 4, 6, ATTRIB_PATTERN_TIP | ATTRIB_SITE_REPEATS)   # 5 states
    # 20 states on the bit-exact vector kernels: accepted, everything stays stored per site
    monkeypatch.setenv("PLLHIP_AA_EXACT", "1")
    aa = make_case(20, "balanced", 8, 60, seed=3)
    p = build_partition(gpu, aa, ATTRIB_PATTERN_TIP | ATTRIB_SITE_REPEATS)
    q = build_partition(gpu, aa, ATTRIB_PATTERN_TIP)
    for x in (p, q):
        x.update_partials(aa["plan"].ops)
    assert all(p.repeats_classes(int(op["parent_clv_index"])) == 0 for op in aa["plan"].ops)
    same_state(p, q, aa["plan"])
    p.destroy()
    q.destroy()
    # 20 states x 8 (or 6, 3) rate categories on the default path (round 5; until then pll_errno 202): the ops run in
    # chunks of the categories, which follow no row maps -- accepted, stored per site, the plain partition's bits
    monkeypatch.delenv("PLLHIP_AA_EXACT", raising=False)
    for cats in (8, 6, 3):
        aa = make_case(20, "random", 10, 300, rate_cats=cats, seed=cats)
        p = build_partition(gpu, aa, ATTRIB_PATTERN_TIP | ATTRIB_SITE_REPEATS)
        q = build_partition(gpu, aa, ATTRIB_PATTERN_TIP)
        ea, eb = evaluate(p, aa["plan"], cats), evaluate(q, aa["plan"], cats)
        assert all(p.repeats_classes(int(op["parent_clv_index"])) == 0 for op in aa["plan"].ops)
        same_state(p, q, aa["plan"])
        assert ea[0] == eb[0] and bits_equal(ea[1], eb[1]) and bits_equal(ea[2], eb[2]) and ea[3] == eb[3]
        p.destroy()
        q.destroy()
    dna = make_case(4, "balanced", 8, 50, seed=3)
    with pytest.raises(PllError):
        build_partition(gpu, dna, ATTRIB_SITE_REPEATS)                           # tip CLVs
    with pytest.raises(PllError):
        gpu.partition_create(8, 6, 4, 50, 1, 14, 3, 6, ATTRIB_PATTERN_TIP | ATTRIB_SITE_REPEATS)   # 4 states x 3 categories
    with pytest.raises(PllError):
        build_partition(gpu, dna, ATTRIB_PATTERN_TIP | ATTRIB_SITE_REPEATS | ATTRIB_AB_LEWIS)


@pytest.mark.parametrize("seed", range(8))
def test_repeats_random_op_sequences(gpu, seed):
    """The arbitrary op sequences of test_gpu_api (slot reuse, repeated children, scale
    buffers changing owner) with site repeats on and off: same CLVs and counts."""
    from helpers import random_sequence_case
    seed = 3 * seed + 1 if seed % 2 else 3 * seed + 2          # 4-state cases of that generator
    case, attrs, ops, rng = random_sequence_case(seed)
    assert case["states"] == 4
    attrs |= ATTRIB_PATTERN_TIP
    sites = case["sites"]
    pool = rng.integers(0, sites, size=sites // 8 + 1)
    pick = pool[rng.integers(0, len(pool), size=sites)]
    case["seqs"] = [bytes(np.frombuffer(s, dtype=np.uint8)[pick]) for s in case["seqs"]]
    plain = build_partition(gpu, case, attrs)
    rep = build_partition(gpu, case, attrs | ATTRIB_SITE_REPEATS)
    plain.update_partials(ops)
    rep.update_partials(ops)
    written = sorted(set(int(x) for x in ops["parent_clv_index"]))
    assert any(rep.repeats_classes(n) for n in written)
    final_scaler = {}
    for op in ops:                                    # the last op that wrote each CLV owns its scaler
        final_scaler[int(op["parent_clv_index"])] = int(op["parent_scaler_index"])
    owner = {}
    for op in ops:
        if int(op["parent_scaler_index"]) >= 0:
            owner[int(op["parent_scaler_index"])] = int(op["parent_clv_index"])
    for node in written:
        assert bits_equal(plain.get_clv(node), rep.get_clv(node)), "CLV slot %d" % node
    for sc, node in owner.items():
        if final_scaler[node] == sc:
            assert (plain.get_scaler(sc) == rep.get_scaler(sc)).all(), "scale buffer %d" % sc
    plain.destroy()
    rep.destroy()


def test_repeats_refuse_foreign_scale_buffer(gpu):
    """A child's counts must come from the buffer written with that child."""
    case = make_case(4, "balanced", 8, 400, seed=9)
    plan = case["plan"]
    rep = build_partition(gpu, case, ATTRIB_PATTERN_TIP | ATTRIB_SITE_REPEATS)
    rep.update_partials(plan.ops)
    bad = plan.ops[4:5].copy()                        # first inner-inner op: children 8, 9
    bad[0]["child1_scaler_index"] = int(bad[0]["child2_scaler_index"])
    rep.update_partials(bad)
    assert gpu.errno() == 113 and "was not written together" in gpu.errmsg()
    rep.destroy()


@pytest.mark.parametrize("states", [4, 20])
def test_repeats_with_invariant_sites_and_weights(gpu, monkeypatch, states):
    """+I model (per-site invariant-state index) and pattern weights are per SITE
    quantities: unchanged by the way CLVs are stored."""
    monkeypatch.setenv("PLLHIP_AA_EXACT", "0")
    case = make_case(states, "random", 18, 1200, seed=states)
    rng = np.random.default_rng(states)
    pool = rng.integers(0, 1200, size=150)
    pick = pool[rng.integers(0, len(pool), size=1200)]
    case["seqs"] = [bytes(np.frombuffer(s, dtype=np.uint8)[pick]) for s in case["seqs"]]
    # some constant columns so that the invariant-site term is exercised
    for col in range(0, 1200, 37):
        case["seqs"] = [s[:col] + case["seqs"][0][col:col + 1] + s[col + 1:] for s in case["seqs"]]
    plan, R = case["plan"], case["rate_cats"]
    res = []
    for attrs in (ATTRIB_PATTERN_TIP, ATTRIB_PATTERN_TIP | ATTRIB_SITE_REPEATS):
        p = build_partition(gpu, case, attrs, pinv=0.23)
        res.append(evaluate(p, plan, R))
        if attrs & ATTRIB_SITE_REPEATS:
            assert sum(1 for op in plan.ops if p.repeats_classes(int(op["parent_clv_index"]))) >= 8
        p.destroy()
    a, b = res
    assert a[0] == b[0] and bits_equal(a[1], b[1]) and bits_equal(a[2], b[2]) and a[3] == b[3]


@pytest.mark.parametrize("tips,sites", [(16, 70), (16, 700), (16, 140_000), (64, 600_000)])
def test_identification_on_every_sort_path(gpu, tips, sites):
    """repeats.hip sorts (row at the major child, row at the minor child) keys: a merge sort up to 2^17 sites, Onesweep
    above; 32-bit keys while both parts fit, 64-bit beyond (both children above 2^16 rows: the 600 k case); class
    numbers from per-wave head counts whose prefixes take one pass up to 2^19 sites and a carry beyond.  Whatever the
    path, every observable equals the plain partition's."""
    case = make_case(4, "balanced", tips, sites, rate_cats=4, seed=sites)
    rng = np.random.default_rng(sites)
    pool = rng.integers(0, sites, size=sites // 6 + 1)
    pick = pool[rng.integers(0, len(pool), size=sites)]
    case["seqs"] = [bytes(np.frombuffer(s, dtype=np.uint8)[pick]) for s in case["seqs"]]
    plan = case["plan"]
    plain = build_partition(gpu, case, ATTRIB_PATTERN_TIP)
    rep = build_partition(gpu, case, ATTRIB_PATTERN_TIP | ATTRIB_SITE_REPEATS)
    a, b = evaluate(plain, plan, 4), evaluate(rep, plan, 4)
    rows = {int(op["parent_clv_index"]): rep.repeats_classes(int(op["parent_clv_index"])) for op in plan.ops}
    assert sum(1 for r in rows.values() if r) >= len(plan.ops) // 2, rows
    if sites == 600_000:
        top = plan.ops[-1]
        assert min(rows[int(top["child1_clv_index"])], rows[int(top["child2_clv_index"])]) > 65536, rows
    assert a[0] == b[0] and bits_equal(a[1], b[1])
    assert bits_equal(a[2], b[2]) and a[3] == b[3]
    for op in plan.ops[-3:]:
        node, sc = int(op["parent_clv_index"]), int(op["parent_scaler_index"])
        assert bits_equal(plain.get_clv(node), rep.get_clv(node)), "CLV %d" % node
        if sc >= 0:
            assert (plain.get_scaler(sc) == rep.get_scaler(sc)).all(), "scaler %d" % sc
    # a subtree swap below the root: the classes of the ops above it are identified again, from the kept orders
    ops = plan.ops.copy()
    i, j = len(ops) - 3, len(ops) - 2
    for f in ("child2_clv_index", "child2_matrix_index", "child2_scaler_index"):
        ops[i][f], ops[j][f] = ops[j][f], ops[i][f]
    for p in (plain, rep):
        p.update_partials(ops[-3:])
    la = plain.compute_edge_loglikelihood(*plan.root_edge, [0] * 4, persite=True)
    lb = rep.compute_edge_loglikelihood(*plan.root_edge, [0] * 4, persite=True)
    assert la[0] == lb[0] and bits_equal(la[1], lb[1]) and la[0] != a[0]
    plain.destroy()
    rep.destroy()
