"""CPU-only checks of the host side: the C-ABI library loads and exports every
declared symbol, host-side numerics (Gamma rates, eigen solver, character maps)
agree with the reference, the device expm1 restatement equals the C library's,
and the library refuses to run without a GPU instead of falling back."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from helpers import bits_equal
from libpll_amd import workload as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header, macro):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = "\n".join(l for l in text.splitlines() if not l.lstrip().startswith("#"))
    funcs = re.findall(macro + r"\s+[^;{]*?\b(\w+)\s*\(", text)
    data = re.findall(macro + r"\s+extern\s+[^;]*?\b(\w+)(?:\[[^\]]*\])*\s*;", text)
    return sorted(set(funcs)), sorted(set(data))


@pytest.mark.parametrize("header,macro", [("pll_amd.h", "PLL_EXPORT"), ("pllhip.h", "PLLHIP_EXPORT")])
def test_library_exports_every_declared_symbol(amd, header, macro):
    funcs, data = declared_symbols(header, macro)
    assert len(funcs) >= 25
    out = subprocess.run(["nm", "-D", "--defined-only", amd.path], capture_output=True, text=True,
                         check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    missing = [s for s in funcs + data if s not in exported]
    assert not missing, "declared in include/%s but not exported: %s" % (header, missing)


def test_drop_in_header_compiles_reference_style_client(tmp_path):
    """include/pll.h lets a client written against the reference's header name
    compile unchanged (compile only: running needs a GPU)."""
    src = tmp_path / "client.c"
    src.write_text('#include "pll.h"\n'
                   "int main(void) { pll_operation_t op; pll_partition_t * p =\n"
                   "  pll_partition_create(4, 2, 4, 6, 1, 5, 4, 2, PLL_ATTRIB_ARCH_AVX2);\n"
                   "  (void)op; if (!p) return pll_errno; pll_partition_destroy(p); return 0; }\n")
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o",
                    str(tmp_path / "client.o")], check=True)


def test_no_device_fails_loudly(amd):
    """No GPU => pll_partition_create returns NULL with PLL_ERROR_HIP_NODEVICE;
    there is no CPU fallback to fall into."""
    if amd.device_count() > 0:
        pytest.skip("a GPU is visible here")
    from libpll_amd.pllapi import PllError
    with pytest.raises(PllError):
        amd.partition_create(4, 2, 4, 10, 1, 5, 4, 2, 0)
    assert amd.errno() == 200


def test_character_maps(amd, ref):
    for name in ("nt", "aa", "bin"):
        assert (amd.map(name) == ref.map(name)).all()


AA_MODELS = ("dayhoff", "lg", "dcmut", "jtt", "mtrev", "wag", "rtrev", "cprev", "vt", "blosum62",
             "mtmam", "mtart", "mtzoa", "pmb", "hivb", "hivw", "jttdcmut", "flu", "stmtrev")


def test_aa_models_match_reference_data(amd, ref):
    import ctypes as C
    for name in AA_MODELS:
        ra, fa = amd.aa_model(name)
        rr, fr = ref.aa_model(name)
        assert bits_equal(ra, rr) and bits_equal(fa, fr)
        assert abs(fa.sum() - 1.0) < 1e-5
    for name in ("lg4m", "lg4x"):
        for kind, n in (("rates", 190), ("freqs", 20)):
            sym = "pll_aa_%s_%s" % (kind, name)
            a = np.ctypeslib.as_array(((C.c_double * n) * 4).in_dll(amd.lib, sym))
            r = np.ctypeslib.as_array(((C.c_double * n) * 4).in_dll(ref.lib, sym))
            assert bits_equal(a, r), sym


def test_gamma_categories_bit_exact(amd, ref):
    for alpha in (0.02, 0.05, 0.3, 0.7, 1.0, 2.5, 10.0, 99.0):
        for cats in (1, 2, 3, 4, 8, 16):
            for mode in (0, 1):
                assert bits_equal(amd.compute_gamma_cats(alpha, cats, mode),
                                  ref.compute_gamma_cats(alpha, cats, mode)), (alpha, cats, mode)
    from libpll_amd.pllapi import PllError
    with pytest.raises(PllError):
        amd.compute_gamma_cats(0.001, 4)
    assert amd.errno() == 113


def test_gamma_categories_known_values(amd):
    """Yang (1994) table values: mean-discretised Gamma, alpha = 0.5, 4 categories."""
    r = amd.compute_gamma_cats(0.5, 4)
    assert np.allclose(r, [0.03338775, 0.25191592, 0.82026848, 2.89442785], atol=2e-8)
    assert abs(r.mean() - 1.0) < 1e-9


@pytest.mark.parametrize("states", [4, 5, 7, 20])
def test_eigen_solver_bit_exact(amd, ref, states):
    rng = np.random.default_rng(states)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    for trial in range(6):
        params = rng.uniform(0.2, 5.0, states * (states - 1) // 2)
        freqs = rng.dirichlet(np.ones(states) * 5)
        if states == 20 and trial == 0:
            params, freqs = ref.aa_model("lg")
        p = ref.partition_create(2, 1, states, 4, 1, 1, 4, 0, 0)
        p.set_subst_params(0, params)
        p.set_frequencies(0, freqs)
        p.update_eigen(0)
        v, ev, inv = p.get_eigen(0)
        p.destroy()
        v2, ev2, inv2 = np.zeros(states), np.zeros((states, states)), np.zeros((states, states))
        assert amd.lib.pll_amd_eigen_decompose(states, dp(np.ascontiguousarray(params)),
                                               dp(np.ascontiguousarray(freqs)), dp(v2), dp(ev2),
                                               dp(inv2))
        assert bits_equal(v, v2) and bits_equal(ev, ev2) and bits_equal(inv, inv2)
        # and it IS an eigen system of Q: V^-1 diag(lambda) V == Q
        q = W.q_matrix(params / params[-1], freqs)
        assert np.allclose(inv2 @ np.diag(v2) @ ev2, q, atol=1e-12)


def test_device_expm1_restatement_equals_libm(tmp_path):
    """numerics.hpp's pll_expm1, compiled for the host, against glibc expm1 on
    4e6 arguments covering every branch (the P-matrix bit-exactness hinges on it)."""
    exe = tmp_path / "check_expm1"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-o", str(exe),
                    os.path.join(ROOT, "oracle", "check_expm1.cpp"),
                    "-I", os.path.join(ROOT, "libpll_amd", "csrc", "hip")], check=True)
    out = subprocess.run([str(exe), "4000000"], capture_output=True, text=True, check=True).stdout
    assert "mismatches 0" in out, out


def test_tree_plans():
    for T in (4, 8, 64):
        plan = W.balanced_tree(T)
        assert len(plan.ops) == T - 2 and plan.op_kinds() == (T // 2, 0, T // 2 - 2)
        assert len(set(plan.matrix_indices.tolist())) == 2 * T - 3
    plan = W.caterpillar_tree(10)
    assert plan.op_kinds() == (1, 7, 0) and plan.root_edge[2] == 9
    plan = W.random_tree(200, seed=1)
    assert len(plan.ops) == 198
    seen = set(range(200))
    for op in plan.ops:   # children are computed before their parent
        assert int(op["child1_clv_index"]) in seen and int(op["child2_clv_index"]) in seen
        seen.add(int(op["parent_clv_index"]))


def test_shard_bounds():
    for sites, n in ((1_000_000, 8), (8_000_000, 8), (1000, 2), (1300, 3), (255, 1)):
        b = W.shard_bounds(sites, n)
        assert b[0] == 0 and b[-1] == sites and len(b) == n + 1
        assert all(b[i] < b[i + 1] for i in range(n))
        assert all(x % 256 == 0 for x in b[1:-1])
    # a rank without sites would leave the others waiting in the first collective: refused,
    # identically on every rank
    for sites, n in ((1000, 3), (255, 2), (1, 4), (512, 3)):
        with pytest.raises(ValueError):
            W.shard_bounds(sites, n)


def test_bench_cpu_baseline_leg(ref):
    """bench.py's cpu_baseline machinery (the reference library on sliced copies of
    the workload, one process per core) runs here without a GPU and reports a
    plausible rate; its core count follows the cgroup quota."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cores = bench.usable_cores()
    assert 1 <= cores <= len(os.sched_getaffinity(0))
    from libpll_amd import workload as W
    from libpll_amd.pllapi import ATTRIB_PATTERN_TIP, ATTRIB_ARCH_AVX2
    plan = W.balanced_tree(16, seed=42)
    seqs = W.random_alignment(16, 4000, 4, seed=1)
    ref_path = os.path.join(root, "oracle", "_ref", "libpll_ref.so")
    rate = bench.cpu_all_cores(ref_path, plan, seqs, 4, 4, ATTRIB_PATTERN_TIP | ATTRIB_ARCH_AVX2, 2, 3)
    assert rate is not None and 1.0 < rate < 1e5      # M site-updates/s on two cores
    assert set(bench.BYTES_PER_SITE["ii"]) == {4, 20}


# ---------------------------------------------------------------- the op-list planner (host logic)

def _plan_dry(amd, ops, tips, clv_buffers, scale_buffers, pattern_tip, nslots):
    """(rc, order, operands copied back from HBM) from pllhip_fused_plan_dry."""
    ops = np.ascontiguousarray(ops)
    n = len(ops)
    order = (C.c_uint * n)()
    slots = (C.c_int * (6 * n))()
    hbm = C.c_uint()
    rc = amd.lib.pllhip_fused_plan_dry(C.c_uint(tips), C.c_uint(clv_buffers), C.c_uint(scale_buffers),
                                       C.c_int(pattern_tip), ops.ctypes.data_as(C.c_void_p), C.c_uint(n),
                                       C.c_uint(nslots), order, C.byref(hbm), slots)
    if rc == 0:
        _simulate_slots(ops, tips if pattern_tip else 0, list(order), np.array(slots).reshape(n, 6), nslots)
    return rc, list(order), hbm.value


def _simulate_slots(ops, pattern_tips, order, slots, nslots):
    """Walk a plan the way the kernel does and check that every inner operand is found where
    the plan says: in the slot its producer (or the copy from HBM issued at the top of the op
    before) left it in, with nothing having overwritten it in between."""
    content = [None] * nslots            # what each slot holds: ("v", list op) or ("hbm", clv index)
    written = {}                         # clv index -> list op that wrote it last (in plan order)

    def inner_operands(i):
        op = ops[i]
        c = [int(op["child1_clv_index"]), int(op["child2_clv_index"])]
        inner = [x for x in c if x >= pattern_tips]
        if len(inner) == 2:
            return {0: c[0], 1: c[1]}     # left, right
        if len(inner) == 1:
            return {1: inner[0]}          # a tip-inner op presents its inner child on the right
        return {}

    def expect(clv):
        return ("v", written[clv]) if clv in written else ("hbm", clv)

    def copy_in(pos):
        i = order[pos]
        for side, clv in inner_operands(i).items():
            if slots[pos][5] & (1 << side):
                s = int(slots[pos][side])
                assert 0 <= s < nslots
                content[s] = ("arriving", expect(clv), pos)

    copy_in(0)
    for pos, i in enumerate(order):
        if pos + 1 < len(order):
            before = list(content)
            copy_in(pos + 1)              # top of the op: the next op's copies are issued
            for s in range(nslots):
                if content[s] != before[s]:
                    # the slot must not be read or written by the op that runs meanwhile
                    assert s not in (int(slots[pos][0]), int(slots[pos][1]), int(slots[pos][2])), (pos, s)
        for side, clv in inner_operands(i).items():
            s = int(slots[pos][side])
            assert 0 <= s < nslots, (pos, side)
            if slots[pos][5] & (1 << side) and clv in written:
                # copied back from HBM at the top of the op before: its producer's stores must
                # have left by then
                assert order.index(written[clv]) + 2 < pos
            want = expect(clv)
            got = content[s]
            if got is not None and got[0] == "arriving":
                assert got[2] <= pos      # (a value copied back for an earlier reader stays for later ones)
                got = got[1]
            assert got == want, (pos, side, s, got, want)
        ps = int(slots[pos][2])
        if ps >= 0:
            content[ps] = ("v", i)
        written[int(ops[i]["parent_clv_index"])] = i


def _order_respects_hazards(ops, order):
    """The planner may re-order a list in any way that keeps every read-after-write,
    write-after-read and write-after-write pair on CLV and scale-buffer indices in list order."""
    pos = {i: p for p, i in enumerate(order)}
    if sorted(order) != list(range(len(ops))):
        return False
    for kind in ("clv", "sc"):
        last_w, readers = {}, {}
        for i, op in enumerate(ops):
            if kind == "clv":
                reads = [int(op["child1_clv_index"]), int(op["child2_clv_index"])]
                write = int(op["parent_clv_index"])
            else:
                reads = [int(x) for x in (op["child1_scaler_index"], op["child2_scaler_index"]) if x >= 0]
                write = int(op["parent_scaler_index"])
            for r in reads:
                if r in last_w and not pos[last_w[r]] < pos[i]:
                    return False
                readers.setdefault(r, []).append(i)
            if write >= 0:
                if write in last_w and last_w[write] != i and not pos[last_w[write]] < pos[i]:
                    return False
                for r in readers.get(write, []):
                    if r != i and not pos[r] < pos[i]:
                        return False
                last_w[write] = i
                readers[write] = []
    return True


def test_fused_planner_on_tree_shapes(amd):
    """Depth-first, heavier subtree first bounds the live values by the tree's Strahler
    number: a balanced 64-taxon list needs no operand from HBM with 5 slots, a balanced
    128-taxon list (BASELINE config 4) none with 6 -- what the 12-wave configuration of
    the whole-list kernel has -- and a random 200-taxon list (config 5's shape) a handful."""
    for plan, nslots, most in ((W.balanced_tree(64), 5, 0), (W.balanced_tree(128), 6, 0),
                               (W.balanced_tree(128), 5, 4), (W.random_tree(200, seed=42), 6, 8),
                               (W.caterpillar_tree(300), 5, 0)):
        rc, order, hbm = _plan_dry(amd, plan.ops, plan.tips, plan.clv_buffers, plan.scale_buffers, 1, nslots)
        assert rc == 0
        assert _order_respects_hazards(plan.ops, order)
        assert hbm <= most, (plan.shape, plan.tips, nslots, hbm)
    # tips as CLVs: every tip operand comes from HBM, nothing else does with 7 slots
    plan = W.balanced_tree(64)
    rc, order, hbm = _plan_dry(amd, plan.ops, plan.tips, plan.clv_buffers, plan.scale_buffers, 0, 7)
    assert rc == 0 and hbm == 64 and _order_respects_hazards(plan.ops, order)


def test_fused_planner_on_random_op_sequences(amd):
    """Arbitrary lists with heavy CLV / scale-buffer reuse: whatever the planner accepts
    (rc 0) is a hazard-respecting permutation; a list it declines (rc 1: e.g. counts that
    were not written together with their CLV) goes to the per-level launches."""
    from helpers import random_op_sequence
    taken = 0
    for seed in range(40):
        rng = np.random.default_rng(seed)
        tips, inner, scalers = 12, 10, 10
        ops = random_op_sequence(rng, tips, inner, scalers, 2 * tips - 3, 60 + seed)
        for pattern_tip in (0, 1):
            for nslots in (5, 6, 7):
                rc, order, _ = _plan_dry(amd, ops, tips, inner, scalers, pattern_tip, nslots)
                assert rc in (0, 1)
                if rc == 0:
                    taken += 1
                    assert _order_respects_hazards(ops, order), (seed, pattern_tip, nslots)
    assert taken > 40
    # indices out of range are refused, not read
    bad = W.balanced_tree(8).ops.copy()
    bad[0]["child1_clv_index"] = 1000
    assert _plan_dry(amd, bad, 8, 6, 6, 1, 5)[0] == -1


def _segments_dry(amd, ops, tips, clv_buffers, scale_buffers, pattern_tip, max_segments=8):
    ops = np.ascontiguousarray(ops)
    n = len(ops)
    seg = (C.c_uint * n)()
    amd.lib.pllhip_fused_segments_dry.restype = C.c_uint
    ns = amd.lib.pllhip_fused_segments_dry(C.c_uint(tips), C.c_uint(clv_buffers), C.c_uint(scale_buffers),
                                           C.c_int(pattern_tip), ops.ctypes.data_as(C.c_void_p), C.c_uint(n),
                                           C.c_uint(max_segments), seg)
    return ns, list(seg)


def _segments_are_independent(ops, seg):
    """No buffer written by an op of one segment is touched by an op of another."""
    writes, touches = {}, {}
    for i, op in enumerate(ops):
        w = [("clv", int(op["parent_clv_index"]))] + ([("sc", int(op["parent_scaler_index"]))] if op["parent_scaler_index"] >= 0 else [])
        t = w + [("clv", int(op["child1_clv_index"])), ("clv", int(op["child2_clv_index"]))] + \
            [("sc", int(x)) for x in (op["child1_scaler_index"], op["child2_scaler_index"]) if x >= 0]
        for b in w:
            writes.setdefault(b, set()).add(seg[i])
        for b in t:
            touches.setdefault(b, set()).add(seg[i])
    return all(len(touches[b]) == 1 for b in writes)


def test_fused_segments(amd):
    """pllhip_fused_segments (round 5; host logic of the whole-list kernels' (tile, segment) work items): a full
    traversal of an unrooted tree is the two sides of its root edge -- two segments of 31 ops for BASELINE config 2's
    64 taxa, the longer side first; a partial traversal (a path to the root) is one; random lists with heavy buffer
    reuse split only into sets that share no written buffer; every segment has two ops at least."""
    plan = W.balanced_tree(64)
    ns, seg = _segments_dry(amd, plan.ops, plan.tips, plan.clv_buffers, plan.scale_buffers, 1)
    assert ns == 2 and sorted(seg.count(k) for k in range(ns)) == [31, 31]
    assert _segments_are_independent(plan.ops, seg)
    # tips as CLVs (nobody writes a tip): the same split
    ns0, seg0 = _segments_dry(amd, plan.ops, plan.tips, plan.clv_buffers, plan.scale_buffers, 0)
    assert ns0 == 2 and seg0 == seg
    # the sides of a random tree are unequal: segment 0 is the longer one
    plan = W.random_tree(200, seed=42)
    ns, seg = _segments_dry(amd, plan.ops, plan.tips, plan.clv_buffers, plan.scale_buffers, 1)
    assert ns == 2 and seg.count(0) >= seg.count(1) >= 2 and _segments_are_independent(plan.ops, seg)
    # max_segments = 1 and lists that do not split
    assert _segments_dry(amd, plan.ops, plan.tips, plan.clv_buffers, plan.scale_buffers, 1, max_segments=1)[0] == 1
    cat = W.caterpillar_tree(40)
    ns, seg = _segments_dry(amd, cat.ops, cat.tips, cat.clv_buffers, cat.scale_buffers, 1)
    assert (ns == 1 and set(seg) == {0}) or min(seg.count(k) for k in range(ns)) >= 2
    view = W.UnrootedView(W.balanced_tree(64))
    ops_r, _ = view.traversal(view.root)
    part = view.partial(ops_r, [view.edges()[5]], view.root)
    assert _segments_dry(amd, part, 64, 62, 62, 1)[0] == 1
    # random lists: whatever the split, it is one into independent sets of >= 2 ops; many components are dealt
    # to at most eight segments
    from helpers import random_op_sequence
    split = 0
    for seed in range(60):
        rng = np.random.default_rng(seed)
        tips, inner, scalers = 12, 14, 14
        ops = random_op_sequence(rng, tips, inner, scalers, 2 * tips - 3, 30 + seed)
        for pattern_tip in (0, 1):
            ns, seg = _segments_dry(amd, ops, tips, inner, scalers, pattern_tip)
            assert 1 <= ns <= 8 and set(seg) == set(range(ns))
            assert _segments_are_independent(ops, seg), seed
            if ns > 1:
                split += 1
                assert min(seg.count(k) for k in range(ns)) >= 2
    # sixteen cherries: sixteen components, eight segments of two
    cher = np.zeros(16, dtype=W.balanced_tree(8).ops.dtype)
    for i in range(16):
        cher[i]["parent_clv_index"] = 32 + i
        cher[i]["parent_scaler_index"] = i
        cher[i]["child1_clv_index"], cher[i]["child2_clv_index"] = 2 * i, 2 * i + 1
        cher[i]["child1_scaler_index"] = cher[i]["child2_scaler_index"] = -1
        cher[i]["child1_matrix_index"], cher[i]["child2_matrix_index"] = 2 * i, 2 * i + 1
    ns, seg = _segments_dry(amd, cher, 32, 16, 16, 1)
    assert ns == 8 and all(seg.count(k) == 2 for k in range(8))
    # more components than segments: three independent ladders of 5, 4 and 4 ops over two segments are
    # loads of 5 and 8 -- segment 0 must still be the longest (the kernels hand its tiles out first)
    def ladder(first_tip, first_inner, n):
        out = np.zeros(n, dtype=cher.dtype)
        for i in range(n):
            out[i]["parent_clv_index"] = first_inner + i
            out[i]["parent_scaler_index"] = first_inner + i - 32
            out[i]["child1_clv_index"] = first_tip if i == 0 else first_inner + i - 1
            out[i]["child2_clv_index"] = first_tip + 1 + i
            out[i]["child1_scaler_index"] = -1 if i == 0 else first_inner + i - 1 - 32
            out[i]["child2_scaler_index"] = -1
            out[i]["child1_matrix_index"], out[i]["child2_matrix_index"] = 2 * (first_inner + i - 32), 2 * (first_inner + i - 32) + 1
        return out
    three = np.concatenate([ladder(0, 32, 5), ladder(8, 37, 4), ladder(16, 41, 4)])
    ns, seg = _segments_dry(amd, three, 32, 16, 16, 1, max_segments=2)
    assert ns == 2 and _segments_are_independent(three, seg)
    assert [seg.count(0), seg.count(1)] == [8, 5], "segment 0 is the longest"
    for seed in range(40):
        rng = np.random.default_rng(1000 + seed)
        ops = random_op_sequence(rng, 12, 14, 14, 21, 40 + seed)
        for max_segments in (2, 3, 8):
            ns, seg = _segments_dry(amd, ops, 12, 14, 14, 1, max_segments=max_segments)
            loads = [seg.count(k) for k in range(ns)]
            assert loads == sorted(loads, reverse=True), (seed, loads)


@pytest.mark.parametrize("rate_cats", [1, 2, 4, 8])
def test_tip_character_batches_of_the_whole_list_kernel(amd, rate_cats):
    """pllhip_fused_char_batches (host logic of partials_fused.hip): a wave fetches the characters of its tile's
    tip rows 64 lanes x 16 bytes at a time, in the order the list uses them.  For random lists: every tip row gets
    lanes of its own inside its batch, both rows of an op lie in the same batch, batches never go backwards, an
    op without tips takes no lanes, and the number of batches is what the rows need (no batch but the last may
    leave more than one row's lanes unused)."""
    rng = np.random.default_rng(rate_cats)
    tile_sites = 2 * (64 // (2 * rate_cats))
    lpr = max(1, tile_sites // 16)       # lanes per row
    rpb = 64 // lpr                      # rows per batch
    for count in (1, 2, 7, 62, 126, 198, 1000):
        tips = rng.integers(0, 4, size=count).astype(np.uint32)
        if count == 62:  # a balanced 64-taxon list as planned: 32 tip-tip ops among 30 inner ones
            tips = np.array([3 if i % 2 == 0 and i < 64 else 0 for i in range(62)], dtype=np.uint32)
        chars = (C.c_uint * count)()
        batch = (C.c_uint * count)()
        nb = amd.lib.pllhip_fused_char_batches_dry(tips.ctypes.data_as(C.POINTER(C.c_uint)), C.c_uint(count),
                                                   C.c_uint(rate_cats), chars, batch)
        taken = {}                        # batch -> set of lanes
        rows_in = {}
        last = 0
        for i in range(count):
            ch, b = int(chars[i]), int(batch[i])
            assert b >= last and b < nb
            last = b
            assert bool(ch & (1 << 16)) == bool(tips[i] & 1) and bool(ch & (1 << 17)) == bool(tips[i] & 2)
            for has, lane0 in ((tips[i] & 1, ch & 0xff), (tips[i] & 2, (ch >> 8) & 0xff)):
                if not has:
                    continue
                lanes = set(range(lane0, lane0 + lpr))
                assert lane0 % lpr == 0 and lane0 + lpr <= 64
                assert not (lanes & taken.setdefault(b, set())), "two rows share a lane"
                taken[b] |= lanes
                rows_in[b] = rows_in.get(b, 0) + 1
        assert nb == last + 1
        for b in range(nb - 1):
            assert rows_in.get(b, 0) >= rpb - 1, "a batch was closed with room for a whole op left"


def test_whole_list_kernel_keeps_out_of_the_slot_registers():
    """partials_aa_fused.hip keeps its values in the accumulation registers a0..a109 behind the
    compiler's back (inline assembly).  The generated code is checked: no instruction outside that
    assembly may touch them (without -mllvm -amdgpu-mfma-vgpr-form the compiler parks the matrix cores'
    accumulators there: seen in round 3)."""
    import subprocess
    import sys
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "check_agprs.py")], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "9 kernels checked, 0 instructions" in out.stdout  # (three scaling modes x three cache policies)


def test_device_selection_per_thread_with_a_process_wide_default(amd):
    """pll_amd_set_device: a thread that has set a device uses its own; a thread that has not uses the process-wide
    default, which belongs to the thread that selected a device first -- here the main thread (round 3 made the
    setting thread-local only: a client that selects its device once on the main thread and creates partitions from
    worker threads silently fell back to the environment there; round 4 let ANY thread move the default:
    last writer wins, partitions of threads that had set nothing landed where a sibling had just pointed)."""
    import ctypes
    import threading
    lib = amd.lib
    lib.pll_amd_get_device.restype = ctypes.c_int
    seen = {}

    def worker(name, own):
        if own is not None:
            lib.pll_amd_set_device(own)
        seen[name] = lib.pll_amd_get_device()

    lib.pll_amd_set_device(3)                       # main thread: also the process-wide default
    t = threading.Thread(target=worker, args=("inherits", None))
    t.start(); t.join()
    assert seen["inherits"] == 3
    t = threading.Thread(target=worker, args=("own", 1))
    t.start(); t.join()
    assert seen["own"] == 1
    assert lib.pll_amd_get_device() == 3           # the main thread keeps its own value ...
    t = threading.Thread(target=worker, args=("later", None))
    t.start(); t.join()
    assert seen["later"] == 3                       # ... and so does the default: a worker's choice is its own
    lib.pll_amd_set_device(2)                       # the owner of the default moves it
    t = threading.Thread(target=worker, args=("after the owner's change", None))
    t.start(); t.join()
    assert seen["after the owner's change"] == 2
    # (round 6) the owner is named by its thread id and can give the ownership up: pll_amd_set_device(-1);
    # the next thread that selects a device becomes the owner -- here a worker, whose choice then IS the default
    lib.pll_amd_set_device(-1)
    t = threading.Thread(target=worker, args=("new owner", 1))
    t.start(); t.join()
    assert seen["new owner"] == 1
    assert lib.pll_amd_get_device() == 1            # the main thread has no device of its own now: the new default
    t = threading.Thread(target=worker, args=("follows the new owner", None))
    t.start(); t.join()
    assert seen["follows the new owner"] == 1
    lib.pll_amd_set_device(2)                       # not the owner any more: its own choice only
    t = threading.Thread(target=worker, args=("still the worker's default", None))
    t.start(); t.join()
    assert seen["still the worker's default"] == 1 and lib.pll_amd_get_device() == 2
    lib.pll_amd_set_device(0)


def test_bench_multi_gpu_launcher_branch(monkeypatch):
    """bench.py --gpus N (N > 1, nobody launched the ranks): the branch the driver's scaling run relies on, on CPU
    (VERDICT r4 item 7b).  The ranks are started as a FRESH child -- python -m torch.distributed.run, one node, N
    processes, rendezvous on 127.0.0.1, dmabuf IPC -- with this process's own arguments, before anything here has
    initialised a GPU and without this process replacing itself (no os.exec*); the child's exit code is the run's.
    Without devices the branch refuses with a message instead of launching."""
    import subprocess
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    cmd, env = bench.launcher_command(8, ["--gpus", "8", "--steps", "3"], 29517)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29517"
    assert cmd[-5] == os.path.join(root, "bench.py") and cmd[-4:] == ["--gpus", "8", "--steps", "3"]
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "OMP_NUM_THREADS" in env
    src = open(os.path.join(root, "bench.py")).read()
    assert "os.exec" not in src.replace("os.exec*", "") and "execv" not in src
    # the branch itself: eight devices "visible", the child's exit code handed on, nothing else run in this process
    seen = {}

    def fake_run(c, env=None, **kw):
        seen["cmd"], seen["env"] = c, env
        return subprocess.CompletedProcess(c, 7)

    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "2", "--warmup", "1"])
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    assert seen["cmd"][-6:] == ["--gpus", "8", "--steps", "2", "--warmup", "1"] and seen["env"]["MASTER_ADDR"] == "127.0.0.1"
    # fewer devices than ranks: refused before anything is launched
    seen.clear()
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "device(s) visible" in str(e.value.code) and not seen
    # a launched rank whose WORLD_SIZE disagrees with --gpus is refused too
    monkeypatch.setenv("WORLD_SIZE", "4")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE" in str(e.value.code)


def test_bench_alignment_blocks():
    """workload.global_alignment (bench.py's alignment since round 4): a range is a slice of the whole, whatever
    the block boundaries; `distinct` makes the blocks repeat; reference_lnl's chunks add up."""
    from libpll_amd import workload as W
    plan = W.balanced_tree(8, seed=42)
    cat = np.array([0.1, 0.5, 1.0, 2.4])
    whole = W.global_alignment(plan, 0, 2300, W.GTR_RATES, W.GTR_FREQS, cat, seed=9, block=500)
    assert len(whole) == 8 and all(len(s) == 2300 for s in whole)
    for lo, hi in ((0, 500), (499, 501), (250, 2300), (1000, 1000), (2299, 2300)):
        part = W.global_alignment(plan, lo, hi, W.GTR_RATES, W.GTR_FREQS, cat, seed=9, block=500)
        assert all(w[lo:hi] == p for w, p in zip(whole, part))
    rep = W.global_alignment(plan, 0, 2000, W.GTR_RATES, W.GTR_FREQS, cat, seed=9, block=500, distinct=2)
    assert rep[0][:500] == rep[0][1000:1500] and rep[0][:500] == whole[0][:500] and rep[0][500:1000] == whole[0][500:1000]
    rnd = W.global_alignment(plan, 100, 900, W.GTR_RATES, W.GTR_FREQS, cat, seed=9, block=500, kind="random")
    assert len(rnd[0]) == 800
    with pytest.raises(ValueError):
        W.global_alignment(plan, 10, 5, W.GTR_RATES, W.GTR_FREQS, cat)


# ---------------------------------------------------------------- small partitions: which path (host logic)

def _small_choice(amd, states, sites, plan, pattern_tip=1, ops=None):
    ops = np.ascontiguousarray(plan.ops if ops is None else ops)
    whole, level = C.c_double(), C.c_double()
    amd.lib.pllhip_small_partition_estimate.restype = C.c_int
    rc = amd.lib.pllhip_small_partition_estimate(C.c_uint(states), C.c_uint(sites), C.c_uint(plan.tips),
                                                 C.c_uint(plan.clv_buffers), C.c_int(pattern_tip),
                                                 ops.ctypes.data_as(C.c_void_p), C.c_uint(len(ops)),
                                                 C.byref(whole), C.byref(level))
    return rc, whole.value, level.value


def test_small_partition_path_choice(amd):
    """Below 16,384 sites the library chooses between the whole-list kernel and the per-level launches from the op
    list (partials.hip, whole_list_pays_when_small).  The choices measured on the device
    (profiles/r5_small_partitions_ab.txt, round 4: r4_small_partitions_ab.txt -- the default matches the faster path
    in every row) are pinned here."""
    from libpll_amd import workload as W
    bal64, rnd200, rnd64, bal128 = (W.balanced_tree(64, seed=42), W.random_tree(200, seed=42),
                                    W.random_tree(64, seed=42), W.balanced_tree(128, seed=42))
    ladder = W.caterpillar_tree(100, seed=42)
    # 20 states (round 5: the whole-list kernel walks the two sides of the root edge side by side -- segments --, which
    # moved every crossover down; profiles/r5_small_partitions_ab.txt: the default matches the faster path in every row)
    assert _small_choice(amd, 20, 2000, bal64)[0] == 1      # 108 against 116 us per step (round 4: 167 against 120)
    assert _small_choice(amd, 20, 6000, bal64)[0] == 1      # 119 against 186
    assert _small_choice(amd, 20, 12000, bal64)[0] == 1     # 175 against 286
    assert _small_choice(amd, 20, 2000, rnd200)[0] == 1     # 519 against 541
    assert _small_choice(amd, 20, 6000, rnd200)[0] == 1     # 518 against 771
    assert _small_choice(amd, 20, 3000, rnd64)[0] == 1      # 191 against 277
    assert _small_choice(amd, 20, 3000, bal128)[0] == 1     # 193-200 against 220 (round 4: 321 against 224)
    assert _small_choice(amd, 20, 10000, bal128)[0] == 1    # 330 against 456
    assert _small_choice(amd, 20, 3000, ladder)[0] == 1     # 402 against 1877: a launch per op on the per-level path
    # 4 states
    assert _small_choice(amd, 4, 2000, bal64)[0] == 0       # 45 against 48 us: six launches still win here
    assert _small_choice(amd, 4, 6000, bal64)[0] == 1       # 52-53 against 56
    assert _small_choice(amd, 4, 12000, bal64)[0] == 1      # 56 against 66 (round 4: per level, 67 against 76)
    assert _small_choice(amd, 4, 6000, rnd200)[0] == 1      # 166 against 195
    assert _small_choice(amd, 4, 3000, rnd64)[0] == 1       # 70 against 105
    assert _small_choice(amd, 4, 3000, bal128)[0] == 0      # 72 against 76
    assert _small_choice(amd, 4, 10000, bal128)[0] == 1     # 81 against 95
    assert _small_choice(amd, 4, 3000, ladder)[0] == 1      # 107 against 663
    # a partial traversal -- the path from a changed branch to the root: one op per level -- takes the whole list, from
    # three ops on (4 states: one launch since round 5, 49 against 55-58 us)
    for n in (3, 7, 15):
        chain = ladder.ops[-n:]
        assert _small_choice(amd, 20, 2000, ladder, ops=chain)[0] == 1
        assert _small_choice(amd, 4, 2000, ladder, ops=chain)[0] == 1
    # estimates are positive and finite; a bad index is reported, not dereferenced
    rc, whole, level = _small_choice(amd, 20, 8000, rnd200)
    assert rc == 1 and 0 < whole < level < 1e6
    bad = bal64.ops.copy()
    bad["child1_clv_index"][3] = 10 ** 6
    assert _small_choice(amd, 4, 2000, bal64, ops=bad)[0] == -1


def test_developer_switches_are_gated(amd, monkeypatch):
    """Environment switches (round 5): the ones a client may set are a short table in the library (ctx.hip) and in
    INTEGRATION.md section 6; every other PLLHIP_* variable the sources read is a developer's knob, honoured only
    under PLLHIP_DEVELOPER=1 -- a stray variable cannot move a production run off the tested configuration."""
    lib = amd.lib
    lib.pllhip_env_is_user_switch.argtypes = [C.c_char_p]
    lib.pllhip_env_is_honoured.argtypes = [C.c_char_p]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    read = set()
    src = os.path.join(root, "libpll_amd", "csrc", "hip")
    for name in os.listdir(src):
        text = open(os.path.join(src, name)).read()
        read |= set(re.findall(r'pllhip_env\("([A-Z_0-9]+)"', text))
        # nothing in the device layer reads a PLLHIP_ variable past the gate
        assert not [m for m in re.findall(r'[^_]getenv\("(PLLHIP_[A-Z_0-9]+)"', text) if m != "PLLHIP_DEVELOPER"], name
    assert len(read) > 30
    user = {n for n in read if lib.pllhip_env_is_user_switch(n.encode())}
    assert user == {"PLLHIP_AA_EXACT", "PLLHIP_AA_TI_MFMA", "PLLHIP_FUSED", "PLLHIP_HOSTSUM", "PLLHIP_FUSE_REDUCE",
                    "PLLHIP_SPIN", "PLLHIP_SHARD_THREADS", "PLLHIP_SHARD_POLL", "PLLHIP_SHARD_PIN", "PLLHIP_PLACEMENT_TRIES", "PLLHIP_FUSED_DEBUG",
                    "PLLHIP_RCCL_DEBUG"}
    # the first table holds the client's switches, the second every developer's one
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    sec = doc[doc.index("## 6. Environment switches"):]
    client, developer = sec.split("**Developer's switches")
    in_client = set(re.findall(r"`(PLLHIP_[A-Z_0-9]+)", "\n".join(l for l in client.splitlines() if l.startswith("| `"))))
    in_developer = set(re.findall(r"`(PLLHIP_[A-Z_0-9]+)", "\n".join(l for l in developer.splitlines() if l.startswith("| `"))))
    assert in_client - {"PLLHIP_DEVELOPER"} == user
    assert read - user <= in_developer, sorted(read - user - in_developer)
    assert not (in_developer & user)
    # the gate itself
    dev = sorted(read - user)[0]
    monkeypatch.setenv(dev, "1")
    monkeypatch.setenv("PLLHIP_SPIN", "0")
    # (PLLHIP_DEVELOPER is read once per process -- several developer's switches are read per launch --
    # and again by pllhip_env_reload)
    try:
        monkeypatch.setenv("PLLHIP_DEVELOPER", "1")
        lib.pllhip_env_reload()
        assert lib.pllhip_env_is_honoured(dev.encode()) == 1 and lib.pllhip_env_is_honoured(b"PLLHIP_SPIN") == 1
        monkeypatch.delenv("PLLHIP_DEVELOPER")
        assert lib.pllhip_env_is_honoured(dev.encode()) == 1          # not read again yet
        lib.pllhip_env_reload()
        assert lib.pllhip_env_is_honoured(dev.encode()) == 0 and lib.pllhip_env_is_honoured(b"PLLHIP_SPIN") == 1
        monkeypatch.setenv("PLLHIP_DEVELOPER", "0")
        lib.pllhip_env_reload()
        assert lib.pllhip_env_is_honoured(dev.encode()) == 0
        monkeypatch.delenv(dev)
        monkeypatch.setenv("PLLHIP_DEVELOPER", "1")
        lib.pllhip_env_reload()
        assert lib.pllhip_env_is_honoured(dev.encode()) == 0
    finally:
        monkeypatch.undo()
        lib.pllhip_env_reload()
