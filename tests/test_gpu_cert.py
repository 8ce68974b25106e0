"""The scaling certificate of 20-state partitions (round 6; DESIGN.md 2.2d, include/pll_amd.h
pll_amd_scaling_certificate).

The whole-list kernel runs the mat-vec of tip-inner ops on the matrix cores by default: CLVs
agree with the reference's (core_partials_avx.c:1229-1284) to ~1e-15 per op, and every scaling
decision taken on such a value (core_partials_avx2.c:752-800) must still be the reference's --
north_star asks for scaler counts bit for bit.  These tests build the case the certificate
exists for: a branch length is bisected (with the oracle) until the largest entry of one site
of one tip-inner op sits at 2^-256 (1 +- 2^-40), i.e. closer to the threshold than the two
summation orders are to each other's neighbourhood.  The library must notice, run the list
again in the reference's order and hand back the reference's counts (and, the list being a
full traversal from the tips, its CLVs bit for bit).
"""
import numpy as np
import pytest

from libpll_amd.pllapi import ATTRIB_PATTERN_TIP, ATTRIB_RATE_SCALERS
from helpers import make_case, build_partition, oracle_run, bits_equal, clv_err

pytestmark = pytest.mark.gpu

THR = 2.0 ** -256


def _unscaled_block_max(o, ops, k, site, rate, per_rate):
    """Largest entry of (site[, rate]) of op k's parent BEFORE the scaling step, from the oracle's state."""
    op = ops[k]
    parent, psc = int(op["parent_clv_index"]), int(op["parent_scaler_index"])
    v = o.clv[parent][site] if not per_rate else o.clv[parent][site, rate]
    inherited = 0
    for ch, sc in ((int(op["child1_clv_index"]), int(op["child1_scaler_index"])),
                   (int(op["child2_clv_index"]), int(op["child2_scaler_index"]))):
        if sc >= 0:
            inherited += int(o.scalers[sc][site * o.R + rate] if per_rate else o.scalers[sc][site])
    mine = int(o.scalers[psc][site * o.R + rate] if per_rate else o.scalers[psc][site])
    m = float(v.max())
    return m / 2.0 ** 256 if mine > inherited else m


def _place_at_threshold(orc, o, case, per_rate, side):
    """Bisect branch lengths until some site's largest entry at an op where it scales is THR (1 + side 2^-40).
    Returns (op number, site, relative distance reached); case["plan"].branch_lengths is changed in place."""
    plan = case["plan"]
    ops = plan.ops
    o.update_partials()
    where = {int(mi): i for i, mi in enumerate(plan.matrix_indices)}
    target = THR * (1.0 + side * 2.0 ** -40)
    original = plan.branch_lengths.copy()

    def counts(sc):
        return o.scalers[sc] if sc >= 0 else 0

    # every (op, entry) at which a scaling event happens: the op's count exceeds what it inherits
    events = []
    for k, op in enumerate(ops):
        psc = int(op["parent_scaler_index"])
        if psc < 0:
            continue
        own = o.scalers[psc].astype(np.int64) - counts(int(op["child1_scaler_index"])) - counts(int(op["child2_scaler_index"]))
        for e in np.nonzero(own > 0)[0][:3]:
            events.append((k, int(e)))
    assert events, "the tree does not scale: make it deeper"

    def setlen(mi, t):
        plan.branch_lengths[where[mi]] = t
        o.pmat[mi] = orc.pmatrix(o.S, o.R, o.m["rates"], float(t), o._ev, o._vc, o._iv, o._pinv)

    for first, e in events[:60]:
        site, rate = (e // o.R, e % o.R) if per_rate else (e, 0)
        op = ops[first]
        mats = [int(op["child1_matrix_index"]), int(op["child2_matrix_index"])]
        t0 = [float(original[where[mi]]) for mi in mats]

        def f(which, g):
            # which: 0 / 1 = one of the op's two branches scaled by g, 2 = both
            for j, mi in enumerate(mats):
                setlen(mi, t0[j] * (g if which in (j, 2) else 1.0))
            o.scalers[:] = 0
            o.update_partials(ops[:first + 1])
            return _unscaled_block_max(o, ops, first, site, rate, per_rate) - target

        for which in (0, 1, 2):
            grid = (1.0, 0.5, 2.0, 0.25, 4.0, 0.1, 8.0, 0.02, 16.0, 0.002, 40.0)
            vals = [f(which, g) for g in grid]
            pair = None
            for a in range(len(grid)):
                for b in range(a + 1, len(grid)):
                    if vals[a] * vals[b] < 0 and pair is None:
                        pair = (grid[a], grid[b], vals[a])
            if pair is None:
                continue
            lo, hi, flo = pair
            for _ in range(200):
                mid = 0.5 * (lo + hi)
                if mid == lo or mid == hi:
                    break
                fm = f(which, mid)
                if fm * flo > 0:
                    lo, flo = mid, fm
                else:
                    hi = mid
            best = min((lo, hi), key=lambda g: abs(f(which, g)))
            d = f(which, best) / THR + side * 2.0 ** -40
            if abs(d) < 2.0 ** -36:
                return first, site, d
        for j, mi in enumerate(mats):
            setlen(mi, t0[j])
    pytest.skip("no branch length brackets the threshold for this seed")


@pytest.mark.parametrize("per_rate", [False, True], ids=["per-site", "per-rate"])
@pytest.mark.parametrize("side", [+1, -1], ids=["just-above", "just-below"])
def test_certificate_trips_and_the_list_runs_again(gpu, orc, monkeypatch, side, per_rate):
    monkeypatch.setenv("PLLHIP_FUSED", "2")      # the whole-list kernel at this size
    monkeypatch.delenv("PLLHIP_AA_TI_MFMA", raising=False)
    monkeypatch.delenv("PLLHIP_AA_EXACT", raising=False)
    attrs = ATTRIB_PATTERN_TIP | (ATTRIB_RATE_SCALERS if per_rate else 0)
    case = make_case(20, "caterpillar", 260, 96, seed=11)
    case["rates"], case["freqs"] = gpu.aa_model("lg")
    plan = case["plan"]
    p0 = build_partition(gpu, case, attrs)
    o = oracle_run(orc, gpu, p0, case, attrs)
    p0.destroy()
    k, site, d = _place_at_threshold(orc, o, case, per_rate, side)
    assert abs(d) < 2.0 ** -36, "the bisection did not get within the narrow window: %g" % d
    # the case as it stands now, on both sides
    p = build_partition(gpu, case, attrs)
    o = oracle_run(orc, gpu, p, case, attrs)
    o.update_partials()
    before = p.scaling_certificate()
    p.update_partials(plan.ops)
    for op in plan.ops:
        node, sc = int(op["parent_clv_index"]), int(op["parent_scaler_index"])
        assert (p.get_scaler(sc) == o.scalers[sc]).all(), "scaler counts of buffer %d differ from the reference's" % sc
        assert bits_equal(p.get_clv(node), o.clv[node]), "CLV %d: the list was run again from the tips, in the reference's order" % node
    after = p.scaling_certificate()
    assert after["raised"] > before["raised"] and after["rerun"] > before["rerun"], (before, after)
    assert after["uncertified"] == 0
    lnl = p.compute_edge_loglikelihood(*plan.root_edge, [0] * 4)
    ref = o.edge_loglikelihood(*plan.root_edge)
    assert abs(lnl - ref) <= 1e-11 * abs(ref)
    p.destroy()


def test_certificate_behind_a_result_call(gpu, orc, monkeypatch):
    """The flag of a list is looked at when the NEXT call's result has arrived (no stream wait in between): the log-
    likelihood call right behind the list must come back with the re-run's value."""
    monkeypatch.setenv("PLLHIP_FUSED", "2")
    monkeypatch.delenv("PLLHIP_AA_TI_MFMA", raising=False)
    attrs = ATTRIB_PATTERN_TIP
    case = make_case(20, "caterpillar", 260, 96, seed=11)
    case["rates"], case["freqs"] = gpu.aa_model("lg")
    plan = case["plan"]
    p0 = build_partition(gpu, case, attrs)
    o = oracle_run(orc, gpu, p0, case, attrs)
    p0.destroy()
    _place_at_threshold(orc, o, case, False, -1)
    p = build_partition(gpu, case, attrs)
    o = oracle_run(orc, gpu, p, case, attrs)
    o.update_partials()
    ref = o.edge_loglikelihood(*plan.root_edge)
    for _ in range(3):   # (the second and third call find the kept plan)
        p.update_partials(plan.ops)
        lnl = p.compute_edge_loglikelihood(*plan.root_edge, [0] * 4)
        assert abs(lnl - ref) <= 1e-11 * abs(ref)
        sc = int(plan.ops[-1]["parent_scaler_index"])
        assert (p.get_scaler(sc) == o.scalers[sc]).all()
    c = p.scaling_certificate()
    assert c["rerun"] == 3 and c["uncertified"] == 0, c
    p.destroy()


def test_certificate_on_a_rank_of_several(gpu, orc, monkeypatch):
    """One rank of a multi-process job (pll_amd_comm_init): an evaluation ends in an all-reduce that every rank enters
    once.  The flag is raised on ONE rank only, so that rank must run its list again BEFORE the evaluation -- never
    evaluate, notice, and evaluate (and all-reduce) a second time, which its peers would answer with their NEXT
    collective.  One rank here (a 1-GPU box): the collectives are counted."""
    import ctypes
    monkeypatch.setenv("PLLHIP_FUSED", "2")
    monkeypatch.delenv("PLLHIP_AA_TI_MFMA", raising=False)
    attrs = ATTRIB_PATTERN_TIP
    case = make_case(20, "caterpillar", 260, 96, seed=11)
    case["rates"], case["freqs"] = gpu.aa_model("lg")
    plan = case["plan"]
    p0 = build_partition(gpu, case, attrs)
    o = oracle_run(orc, gpu, p0, case, attrs)
    p0.destroy()
    _place_at_threshold(orc, o, case, False, -1)
    p = build_partition(gpu, case, attrs)
    o = oracle_run(orc, gpu, p, case, attrs)
    o.update_partials()
    ref = o.edge_loglikelihood(*plan.root_edge)
    uid = ctypes.create_string_buffer(128)
    assert gpu.lib.pll_amd_comm_unique_id(uid), gpu.errmsg()
    assert p.comm_reduces() == 0
    p.comm_init(0, 1, uid.raw)
    for k in range(3):
        p.update_partials(plan.ops)
        lnl = p.compute_edge_loglikelihood(*plan.root_edge, [0] * 4)
        assert abs(lnl - ref) <= 1e-11 * abs(ref)
        assert p.comm_reduces() == k + 1, "an evaluation entered %d collectives" % (p.comm_reduces() - k)
        sc = int(plan.ops[-1]["parent_scaler_index"])
        assert (p.get_scaler(sc) == o.scalers[sc]).all()
    c = p.scaling_certificate()
    assert c["rerun"] == 3 and c["uncertified"] == 0, c
    p.destroy()


@pytest.mark.parametrize("shape,tips", [("random", 120), ("caterpillar", 150)])
def test_default_path_counts_bitwise_clvs_to_rounding(gpu, orc, monkeypatch, shape, tips):
    """The default path on trees with many tip-inner ops: counts bit for bit, CLVs to 1e-13, lnL to 1e-12 -- and the
    very same partition with PLLHIP_AA_TI_MFMA=0: every CLV bit for bit, as until round 5."""
    monkeypatch.setenv("PLLHIP_FUSED", "2")
    attrs = ATTRIB_PATTERN_TIP
    case = make_case(20, shape, tips, 700, seed=3)
    case["rates"], case["freqs"] = gpu.aa_model("lg")
    plan = case["plan"]
    for flag in (None, "0"):
        if flag is None:
            monkeypatch.delenv("PLLHIP_AA_TI_MFMA", raising=False)
        else:
            monkeypatch.setenv("PLLHIP_AA_TI_MFMA", flag)
        p = build_partition(gpu, case, attrs)
        o = oracle_run(orc, gpu, p, case, attrs)
        o.update_partials()
        p.update_partials(plan.ops)
        worst = 0.0
        for op in plan.ops:
            node, sc = int(op["parent_clv_index"]), int(op["parent_scaler_index"])
            assert (p.get_scaler(sc) == o.scalers[sc]).all()
            if flag == "0":
                assert bits_equal(p.get_clv(node), o.clv[node])
            else:
                worst = max(worst, clv_err(p.get_clv(node), o.clv[node]))
        assert worst <= 1e-13, worst
        lnl = p.compute_edge_loglikelihood(*plan.root_edge, [0] * 4)
        ref = o.edge_loglikelihood(*plan.root_edge)
        assert abs(lnl - ref) <= 1e-12 * abs(ref)
        c = p.scaling_certificate()
        if flag == "0":
            assert c["lists"] == 0, c
        else:
            assert c["lists"] >= 1 and c["uncertified"] == 0, c
        p.destroy()


def test_partial_traversal_over_marked_clvs(gpu, orc, monkeypatch):
    """A one-op list (per-level kernel, reference order) over CLVs an earlier whole-list call left marked: the narrow
    window's test runs (lists + 1), nothing is raised, the counts are the reference's."""
    monkeypatch.setenv("PLLHIP_FUSED", "2")
    monkeypatch.delenv("PLLHIP_AA_TI_MFMA", raising=False)
    attrs = ATTRIB_PATTERN_TIP
    case = make_case(20, "caterpillar", 150, 300, seed=9)
    case["rates"], case["freqs"] = gpu.aa_model("lg")
    plan = case["plan"]
    p = build_partition(gpu, case, attrs)
    o = oracle_run(orc, gpu, p, case, attrs)
    o.update_partials()
    p.update_partials(plan.ops)
    a = p.scaling_certificate()
    last = plan.ops[-1:]
    p.update_partials(last)          # the top op again, alone: one launch of the per-level kernel
    o.update_partials(last)
    b = p.scaling_certificate()
    sc = int(last[0]["parent_scaler_index"])
    assert (p.get_scaler(sc) == o.scalers[sc]).all()
    assert clv_err(p.get_clv(int(last[0]["parent_clv_index"])), o.clv[int(last[0]["parent_clv_index"])]) <= 1e-13
    assert b["lists"] == a["lists"] + 1 and b["uncertified"] == 0, (a, b)
    p.destroy()


def test_certificate_on_a_sharded_partition(gpu, orc, monkeypatch):
    """One partition over two "devices" (ordinal 0 twice): the shard that holds the site at the threshold raises its
    flag, the group finds it when every shard's lnL is in, that shard runs its list again, and the evaluation is
    repeated -- the value, the counts and the CLVs the client sees are the reference's."""
    import ctypes as C
    monkeypatch.setenv("PLLHIP_FUSED", "2")
    monkeypatch.delenv("PLLHIP_AA_TI_MFMA", raising=False)
    attrs = ATTRIB_PATTERN_TIP
    case = make_case(20, "caterpillar", 260, 700, seed=11)
    case["rates"], case["freqs"] = gpu.aa_model("lg")
    plan = case["plan"]
    p0 = build_partition(gpu, case, attrs)
    o = oracle_run(orc, gpu, p0, case, attrs)
    p0.destroy()
    _place_at_threshold(orc, o, case, False, +1)
    arr = (C.c_int * 2)(0, 0)
    assert gpu.lib.pll_amd_set_devices(arr, 2) == 1
    try:
        p = build_partition(gpu, case, attrs)
    finally:
        gpu.lib.pll_amd_set_devices(None, 0)
    assert gpu.lib.pll_amd_shard_count(p.ptr) == 2
    o = oracle_run(orc, gpu, p, case, attrs)
    o.update_partials()
    ref = o.edge_loglikelihood(*plan.root_edge)
    p.update_partials(plan.ops)
    lnl = p.compute_edge_loglikelihood(*plan.root_edge, [0] * 4)     # (the flags are looked at behind this call)
    assert abs(lnl - ref) <= 1e-11 * abs(ref)
    c = p.scaling_certificate()
    assert c["raised"] >= 1 and c["rerun"] >= 1 and c["uncertified"] == 0, c
    for op in plan.ops[::9]:
        node, sc = int(op["parent_clv_index"]), int(op["parent_scaler_index"])
        assert (p.get_scaler(sc) == o.scalers[sc]).all()
        # (the shard that ran again holds the reference's bits; the other one its matrix-core values)
        assert clv_err(p.get_clv(node), o.clv[node]) <= 1e-12
    p.destroy()
