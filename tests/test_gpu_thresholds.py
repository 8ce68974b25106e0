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
