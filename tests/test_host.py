"""CPU-only checks of the host side: the C-ABI library loads and exports every
declared symbol, host-side numerics (Gamma rates, eigen solver, character maps)
agree with the reference, the device expm1 restatement equals the C library's,
and the library refuses to run without a GPU instead of falling back."""
import ctypes as C
import os
import re
import subprocess

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
    for sites, n in ((1_000_000, 8), (8_000_000, 8), (1000, 3), (255, 2), (1, 4)):
        b = W.shard_bounds(sites, n)
        assert b[0] == 0 and b[-1] == sites and len(b) == n + 1
        assert all(b[i] <= b[i + 1] for i in range(n))
        assert all(x % 256 == 0 for x in b[1:-1] if x != sites)


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
