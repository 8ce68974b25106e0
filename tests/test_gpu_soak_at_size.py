"""The at-size soaks in the driver's GPU suite: the whole-list kernels against the per-level
launches at 100,000 sites x 200 taxa -- the size where a workgroup walks enough tiles for its waves to drift apart
(round 3's race between an inner-inner op and a run of barrier-free lookups showed only there) -- on random TREES over
one alignment: a new op list, the same list again (kept plan), a partial traversal after a branch-length change, seed
after seed; lnL, per-site lnL and the scale buffers of the last ops bit for bit, every 50th seed the CLVs of the last
three ops too.  The loops are tools/soak_aa_fused_at_size.py's (what the builder's logs under profiles/ ran for
thousands of seeds); here ~200 seeds per kernel, seeds that move with nothing, so a failure reproduces.

And one mid-size case under the numerics-relevant environment switches CROSSED (item 4c): PLLHIP_HOSTSUM x
PLLHIP_FUSED x PLLHIP_AA_EXACT -- every combination must give the per-site lnL and the scale buffers of the default
configuration bit for bit where the kernels are bit-exact (4 states: all; 20 states: within each PLLHIP_AA_EXACT
value -- the matrix cores' edge kernel differs from the vector one within 1e-11, tests/test_gpu_parity.py), and lnL
to 1e-12 (the summation trees differ)."""
import itertools
import os
import sys

import numpy as np
import pytest

from helpers import bits_equal, rel_err
from libpll_amd import workload as W
from libpll_amd.pllapi import ATTRIB_PATTERN_TIP

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("states,seeds,rate_scalers", [(20, 200, False), (20, 60, True), (4, 200, False), (4, 60, True)],
                         ids=["20-states", "20-states-per-rate-scalers", "4-states", "4-states-per-rate-scalers"])
def test_whole_list_against_per_level_at_size(gpu, monkeypatch, states, seeds, rate_scalers):
    """(20 states, round 6: the DEFAULT path -- tip-inner mat-vecs of the whole-list kernel on the matrix cores: scale
    buffers bit for bit, lnL / per-site lnL / CLVs to rounding, no uncertified scaling decision; per-rate scale buffers
    on the whole-list kernel.)"""
    monkeypatch.delenv("PLLHIP_AA_EXACT", raising=False)
    monkeypatch.delenv("PLLHIP_AA_TI_MFMA", raising=False)
    import soak_aa_fused_at_size as soak
    assert soak.run(first=31_000, count=seeds, sites=100_000, T=200, states=states, rate_scalers=rate_scalers,
                    quiet=True) == 0


@pytest.mark.parametrize("states", [4, 20])
def test_environment_switches_crossed(gpu, monkeypatch, states):
    sites, T, R = 40_000, 48, 4
    plan = W.random_tree(T, seed=7)
    rates, freqs = gpu.aa_model("lg") if states == 20 else (W.GTR_RATES, W.GTR_FREQS)
    seqs = W.simulated_alignment(plan, sites, rates, freqs, gpu.compute_gamma_cats(W.GAMMA_ALPHA, R), seed=7)
    fi = [0] * R
    got = {}
    for hostsum, fused, exact, ti in itertools.product(("1", "0"), ("0", "2"), ("0", "1") if states == 20 else ("0",),
                                                       ("1", "0") if states == 20 else ("1",)):
        if exact == "1" and ti == "0":
            continue   # (PLLHIP_AA_TI_MFMA has nothing to switch on the all-vector kernels)
        monkeypatch.setenv("PLLHIP_HOSTSUM", hostsum)
        monkeypatch.setenv("PLLHIP_FUSED", fused)
        monkeypatch.setenv("PLLHIP_AA_EXACT", exact)
        monkeypatch.setenv("PLLHIP_AA_TI_MFMA", ti)
        p = W.setup_partition(gpu, plan, seqs, states, R, ATTRIB_PATTERN_TIP)
        p.update_partials(plan.ops)
        lnl, ps = p.compute_edge_loglikelihood(*plan.root_edge, fi, persite=True)
        e = plan.root_edge
        st = p.alloc_sumtable()
        p.update_sumtable(e[0], e[2], e[1], e[3], fi, st)
        d = p.compute_likelihood_derivatives(e[1], e[3], 0.17, fi, st)
        scs = [p.get_scaler(int(op["parent_scaler_index"])) for op in plan.ops[-6:]]
        got[(hostsum, fused, exact, ti)] = (lnl, ps, np.array(d), scs)
        if states == 20:
            assert p.scaling_certificate()["uncertified"] == 0
        p.destroy()
    base = got[("1", "0", "0", "1")]
    for key, (lnl, ps, d, scs) in got.items():
        # the same arithmetic as the base: the matrix-core kernels, and -- 20 states -- not the whole-list kernel with
        # its tip-inner mat-vecs on the matrix cores (round 6; those agree among themselves)
        ti_on_matrix_cores = states == 20 and key[1] == "2" and key[3] == "1" and key[2] == "0"
        same_kernels = key[2] == "0" and not ti_on_matrix_cores
        if same_kernels:
            assert bits_equal(ps, base[1]), "per-site lnL under HOSTSUM=%s FUSED=%s AA_EXACT=%s AA_TI_MFMA=%s" % key
        else:
            assert rel_err(ps, base[1]) < 1e-11, key
        if ti_on_matrix_cores:
            assert bits_equal(ps, got[("1", "2", "0", "1")][1]), key
        assert abs(lnl - base[0]) <= 1e-12 * abs(base[0]), key
        assert rel_err(d, base[2]) < 1e-10, key
        for a, b in zip(scs, base[3]):
            assert (a == b).all(), "scale buffers under HOSTSUM=%s FUSED=%s AA_EXACT=%s AA_TI_MFMA=%s" % key


def test_repeat_identification_at_size(gpu, monkeypatch):
    """Site-repeat classes (repeats.hip) at 100 k - 800 k sites, where the sort changes from a merge sort to Onesweep,
    the keys from 32 to 64 bits and the prefix kernel starts to carry: 24 random trees / column pools, a partition with
    the attribute bitwise against one without it, through two subtree swaps and a replaced tip
    (tools/soak_repeats_at_size.py; profiles/r5_soak_repeats_at_size.log: 120 seeds)."""
    monkeypatch.setenv("PLLHIP_AA_EXACT", "0")   # (20 states: the matrix-core kernels follow row maps; the tool sets it too)
    import soak_repeats_at_size as soak
    assert soak.run(31_000, 24, log=lambda *a: None, amd=gpu) == 0
