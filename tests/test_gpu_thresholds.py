"""Every dispatch threshold has the same bits on both sides.

The library picks kernels by partition size: the whole-list 4-state kernel from one tile per
SIMD on (16,384 sites at 4 rate categories on 256 CUs) and for lists of two ops or more,
non-temporal loads / stores once the CLVs exceed the 256 MB memory-side cache, the 20-state
lookup ops once their saving is twice the cost of their tables (a model in partials.hip).
Which side of a threshold a partition falls on must never show in a result: at sizes right
below, at and above each threshold the default choice, the forced-on and the forced-off variant
give identical CLVs, scale buffers and per-site lnL."""
import numpy as np
import pytest

from helpers import bits_equal
from libpll_amd import workload as W
from libpll_amd.pllapi import ATTRIB_PATTERN_TIP

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _reference_order_tip_inner(monkeypatch):
    """This file compares two PATHS of the library bit for bit (either side of a size threshold, table budgets).  On the default path the 20-state whole-list
    kernel runs tip-inner mat-vecs on the matrix cores (round 6: CLVs to 1e-15 per op, scaler counts bit for bit behind
    the scaling certificate -- tests/test_gpu_cert.py, tests/test_gpu_aa_whole_list.py), so a path that takes that
    kernel and one that does not agree to rounding only; pinned to the reference's order here."""
    monkeypatch.setenv("PLLHIP_AA_TI_MFMA", "0")


def observe(gpu, plan, seqs, states, ops=None, rate_cats=4):
    p = W.setup_partition(gpu, plan, seqs, states, rate_cats, ATTRIB_PATTERN_TIP)
    p.update_partials(plan.ops if ops is None else ops)
    lnl, ps = p.compute_edge_loglikelihood(*plan.root_edge, [0] * rate_cats, persite=True)
    top = plan.ops[-1]
    out = (lnl, ps, p.get_clv(int(top["parent_clv_index"])), p.get_scaler(int(top["parent_scaler_index"])))
    p.destroy()
    return out


def same(a, b):
    return a[0] == b[0] and bits_equal(a[1], b[1]) and bits_equal(a[2], b[2]) and (a[3] == b[3]).all()


@pytest.mark.parametrize("sites", [16383, 16384, 16385, 32768, 40000])
def test_whole_list_kernel_threshold(gpu, monkeypatch, sites):
    plan = W.balanced_tree(16, seed=3)
    seqs = W.random_alignment(16, sites, 4, seed=sites)
    res = {}
    for mode in ("default", "0", "2"):
        if mode == "default":
            monkeypatch.delenv("PLLHIP_FUSED", raising=False)
        else:
            monkeypatch.setenv("PLLHIP_FUSED", mode)
        res[mode] = observe(gpu, plan, seqs, 4)
    assert same(res["default"], res["0"]) and same(res["default"], res["2"])


@pytest.mark.parametrize("rate_cats,sites", [(1, 70_000), (2, 40_000), (8, 8_191), (8, 8_192), (8, 30_001)])
def test_whole_list_kernel_rate_categories(gpu, monkeypatch, rate_cats, sites):
    """The whole-list kernel's other tile shapes -- 64, 32 and (round 3) 8 sites per tile for 1, 2 and 8 rate
    categories -- on both sides of their size threshold."""
    plan = W.random_tree(24, seed=rate_cats)
    seqs = W.random_alignment(24, sites, 4, seed=sites)
    res = {}
    for mode in ("default", "0", "2"):
        if mode == "default":
            monkeypatch.delenv("PLLHIP_FUSED", raising=False)
        else:
            monkeypatch.setenv("PLLHIP_FUSED", mode)
        res[mode] = observe(gpu, plan, seqs, 4, rate_cats=rate_cats)
    assert same(res["default"], res["0"]) and same(res["default"], res["2"])


@pytest.mark.parametrize("nops", [1, 2, 3, 4, 6, 7, 8])
def test_short_list_threshold(gpu, monkeypatch, nops):
    """Single ops run per level also on large partitions, lists from two ops on in one launch (seven
    until round 3): the last `nops` ops of a traversal (a partial traversal towards the root -- every
    op reloads an operand an earlier call wrote) either way."""
    plan = W.caterpillar_tree(12, seed=5)
    seqs = W.random_alignment(12, 40_000, 4, seed=nops)
    res = {}
    for mode in ("default", "0", "2"):
        if mode == "default":
            monkeypatch.delenv("PLLHIP_FUSED", raising=False)
        else:
            monkeypatch.setenv("PLLHIP_FUSED", mode)
        p = W.setup_partition(gpu, plan, seqs, 4, 4, ATTRIB_PATTERN_TIP)
        p.update_partials(plan.ops)
        p.update_prob_matrices([0] * 4, [int(plan.ops[-nops]["child2_matrix_index"])], [0.33])
        p.update_partials(plan.ops[-nops:])
        lnl, ps = p.compute_edge_loglikelihood(*plan.root_edge, [0] * 4, persite=True)
        res[mode] = (lnl, ps, p.get_clv(int(plan.ops[-1]["parent_clv_index"])),
                     p.get_scaler(int(plan.ops[-1]["parent_scaler_index"])))
        p.destroy()
    assert same(res["default"], res["0"]) and same(res["default"], res["2"])


@pytest.mark.parametrize("states,sites", [(4, 20_000), (4, 300_000), (20, 4_000), (20, 60_000)])
def test_cache_policy_threshold(gpu, monkeypatch, states, sites):
    """PLLHIP_NT=0 / 1 / the default (non-temporal once the CLVs exceed 256 MB: 14 inner CLVs of
    16 taxa are 36 MB at 20,000 4-state sites and 538 MB at 300,000)."""
    plan = W.balanced_tree(16, seed=7)
    seqs = W.random_alignment(16, sites, states, seed=sites)
    res = {}
    for mode in ("default", "0", "1", "2"):  # (2: the whole-list kernel's counts non-temporal too)
        if mode == "default":
            monkeypatch.delenv("PLLHIP_NT", raising=False)
        else:
            monkeypatch.setenv("PLLHIP_NT", mode)
        res[mode] = observe(gpu, plan, seqs, states)
    assert same(res["default"], res["0"]) and same(res["default"], res["1"]) and same(res["default"], res["2"])


@pytest.mark.parametrize("taxa,sites", [(64, 20_000), (64, 26_000), (64, 40_000), (16, 90_000), (16, 120_000)])
def test_lookup_op_threshold(gpu, monkeypatch, taxa, sites):
    """20 states: lookup ops off / forced / the model's choice (64 taxa: on from ~26 k sites;
    16 taxa, four such ops: from ~103 k).  The tables are built by the kernels they replace, so
    not even the last bit moves."""
    monkeypatch.setenv("PLLHIP_AA_EXACT", "0")
    plan = W.balanced_tree(taxa, seed=11)
    seqs = W.random_alignment(taxa, sites, 20, seed=taxa + sites)
    res = {}
    for mode in ("default", "0", "2"):
        if mode == "default":
            monkeypatch.delenv("PLLHIP_AA_CHERRY", raising=False)
        else:
            monkeypatch.setenv("PLLHIP_AA_CHERRY", mode)
        res[mode] = observe(gpu, plan, seqs, 20)
    assert same(res["default"], res["0"]) and same(res["default"], res["2"])


@pytest.mark.parametrize("states,sites", [(4, 8_000), (4, 33_000), (4, 300_000), (20, 3_000), (20, 60_000), (5, 70_000)])
def test_final_sum_on_the_host_or_on_the_device(gpu, monkeypatch, states, sites):
    """Round 4: reducing kernels with more than 128 workgroups write their workgroup sums to host-mapped memory and
    the HOST adds them (in k_final_sum's order); below that the last workgroup finishes the sum in the launch.
    PLLHIP_HOSTSUM=0 (the one-workgroup k_final_sum launch again), PLLHIP_SPIN=0 (stream wait instead of polling)
    and PLLHIP_FUSE_REDUCE=0/1 must all give the same lnL and derivatives to the last bit."""
    if states in (4, 20):
        plan = W.balanced_tree(16, seed=2)
        seqs = W.random_alignment(16, sites, states, seed=sites)
    else:
        from helpers import odd_state_case, build_partition
        case = odd_state_case(states, tips=9, sites=sites, seed=4)
    res = {}
    for name, env in (("default", {}), ("device sum", {"PLLHIP_HOSTSUM": "0"}), ("stream wait", {"PLLHIP_SPIN": "0"}),
                      ("device sum + stream wait", {"PLLHIP_HOSTSUM": "0", "PLLHIP_SPIN": "0"}),
                      ("separate launch always", {"PLLHIP_FUSE_REDUCE": "0"})):
        for k in ("PLLHIP_HOSTSUM", "PLLHIP_SPIN", "PLLHIP_FUSE_REDUCE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        if states in (4, 20):
            p = W.setup_partition(gpu, plan, seqs, states, 4, ATTRIB_PATTERN_TIP)
            R = 4
        else:
            p = build_partition(gpu, case, ATTRIB_PATTERN_TIP)
            plan, R = case["plan"], case["rate_cats"]
        p.update_partials(plan.ops)
        e = plan.root_edge
        lnl, ps = p.compute_edge_loglikelihood(*e, [0] * R, persite=True)
        lnl2 = p.compute_edge_loglikelihood(*e, [0] * R)
        st = p.alloc_sumtable()
        p.update_sumtable(e[0], e[2], e[1], e[3], [0] * R, st)
        d = [p.compute_likelihood_derivatives(e[1], e[3], t, [0] * R, st) for t in (0.05, 0.4)]
        res[name] = (lnl, lnl2, d)
        p.destroy()
    for name, got in res.items():
        assert got[0] == got[1], name
        if "always" not in name:   # (the separate launch on a small grid adds in another order than the fused finish)
            assert got == res["default"], name
        else:
            assert abs(got[0] - res["default"][0]) <= 1e-13 * abs(got[0])


@pytest.mark.parametrize("budget_mb", ["0", "1", "2"])
def test_lookup_table_budget(gpu, monkeypatch, budget_mb):
    """20 states, whole-list kernel: the lookup tables and the tip-tip ops' pair tables share a pool with a budget
    (PLLHIP_AA_LOOKUP_MB).  Ops beyond it stay ordinary ops / fall back to the two tip tables: with no
    pool at all, with room for one table and by default the bits are the same."""
    monkeypatch.setenv("PLLHIP_AA_EXACT", "0")
    monkeypatch.setenv("PLLHIP_FUSED", "2")
    monkeypatch.setenv("PLLHIP_AA_CHERRY", "2")
    plan = W.random_tree(40, seed=6)
    seqs = W.random_alignment(40, 3_000, 20, seed=77)
    monkeypatch.delenv("PLLHIP_AA_LOOKUP_MB", raising=False)
    want = observe(gpu, plan, seqs, 20)
    monkeypatch.setenv("PLLHIP_AA_LOOKUP_MB", budget_mb)
    assert same(observe(gpu, plan, seqs, 20), want)
    monkeypatch.setenv("PLLHIP_AA_TT_PAIRS", "0")      # (tip-tip ops of the list over the two tip tables, as in round 3)
    assert same(observe(gpu, plan, seqs, 20), want)


def observe_all(gpu, plan, seqs, states, every=1):
    """every op's CLV and scale buffer, per-site lnL, and the same after a branch-length change + partial traversal"""
    p = W.setup_partition(gpu, plan, seqs, states, 4, ATTRIB_PATTERN_TIP)
    p.update_partials(plan.ops)
    lnl, ps = p.compute_edge_loglikelihood(*plan.root_edge, [0] * 4, persite=True)
    clvs = [p.get_clv(int(op["parent_clv_index"])) for op in plan.ops[::every]]
    scs = [p.get_scaler(int(op["parent_scaler_index"])) for op in plan.ops]
    p.update_partials(plan.ops)                 # (the kept plan)
    lnl2, ps2 = p.compute_edge_loglikelihood(*plan.root_edge, [0] * 4, persite=True)
    n = max(2, len(plan.ops) // 3)
    p.update_prob_matrices([0] * 4, [int(plan.ops[-n]["child1_matrix_index"])], [0.27])
    p.update_partials(plan.ops[-n:])
    lnl3, ps3 = p.compute_edge_loglikelihood(*plan.root_edge, [0] * 4, persite=True)
    p.destroy()
    return (lnl, ps, clvs, scs, lnl2, ps2, lnl3, ps3)


@pytest.mark.parametrize("states,shape,taxa,sites", [(4, "balanced", 64, 3_000), (4, "balanced", 64, 50_000),
                                                     (4, "random", 40, 62_500), (4, "balanced", 16, 200_000),
                                                     (4, "balanced", 8, 400_000), (4, "caterpillar", 30, 20_000),
                                                     (20, "balanced", 64, 3_000), (20, "random", 40, 25_000),
                                                     (20, "balanced", 16, 70_000), (20, "caterpillar", 30, 9_000)])
def test_segments_threshold(gpu, monkeypatch, states, shape, taxa, sites):
    """Round 5: the two sides of the root edge (any sets of ops that share no written buffer) as SEGMENTS of one
    whole-list launch -- (tile, segment) work items, handed out while the tiles alone do not fill the chip eight
    times over.  Never / two / up to eight / the size rule: every CLV, every scale buffer and the per-site lnL are
    the same bits, also through the kept plan and a partial traversal (one segment again)."""
    monkeypatch.delenv("PLLHIP_AA_EXACT", raising=False)
    monkeypatch.setenv("PLLHIP_FUSED", "2")
    plan = {"balanced": W.balanced_tree, "random": W.random_tree, "caterpillar": W.caterpillar_tree}[shape](taxa, seed=9)
    seqs = W.random_alignment(taxa, sites, states, seed=sites)
    res = {}
    for mode in ("0", "2", "8", "default"):
        if mode == "default":
            monkeypatch.delenv("PLLHIP_FUSED_SEGMENTS", raising=False)
        else:
            monkeypatch.setenv("PLLHIP_FUSED_SEGMENTS", mode)
        res[mode] = observe_all(gpu, plan, seqs, states, every=1 if sites <= 70_000 else 7)
    want = res["0"]
    for mode, got in res.items():
        assert got[0] == want[0] and got[4] == want[4] and got[6] == want[6], mode
        assert bits_equal(got[1], want[1]) and bits_equal(got[5], want[5]) and bits_equal(got[7], want[7]), mode
        assert got[0] == got[4] and bits_equal(got[1], got[5]), mode
        for a, b in zip(got[2], want[2]):
            assert bits_equal(a, b), mode
        for a, b in zip(got[3], want[3]):
            assert (a == b).all(), mode
