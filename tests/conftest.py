"""pytest configuration.

Markers
  gpu      needs a real MI355X (selected with ``-m gpu`` on the GPU box); these are
           the parity tests proper and call the product through its C-ABI.
  (none)   CPU-only: oracle vs golden vectors / reference build, host logic,
           symbol exports, gloo multi-process sharding.

The product library and the CPU checkers are built on demand (make) when their
shared objects are missing, so a fresh checkout can run ``pytest`` directly.
"""
import os
import subprocess
import sys

import pytest

# tests set tile shapes, grid caps and the like to reach code paths at small sizes: developer's switches, which the
# library honours only under this one (tests/test_host.py::test_developer_switches_are_gated)
os.environ.setdefault("PLLHIP_DEVELOPER", "1")
# Host mirrors: the suite drives the product like a client that KNOWS it (pll_amd_sync_* before it reads
# partition->clv[i]) and reads hundreds of small partitions' CLVs through those calls; the mirrors the library keeps
# current by itself for partitions below 64 MB of CLVs (round 6) would copy every CLV back a second time.  Off for
# the suite; tests/test_gpu_api.py::test_small_partitions_keep_their_mirrors_current and the reference's own client
# programs (tests/test_reference_programs.py) run with the default.
os.environ.setdefault("PLL_AMD_AUTO_MIRROR_MB", "0")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

LIB = os.environ.get("PLL_AMD_LIB") or os.path.join(ROOT, "libpll_amd", "libpll_amd.so")
ORACLE = os.environ.get("PLL_ORACLE_LIB") or os.path.join(ROOT, "oracle", "liboracle.so")
REF = os.path.join(ROOT, "oracle", "_ref", "libpll_ref.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X GPU (run with -m gpu)")


def _make(target, cwd=ROOT):
    subprocess.run(["make", "-j4", target], cwd=cwd, check=True, stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def amd():
    """The product library.  Missing library = hard failure, never a skip."""
    if not os.path.exists(LIB):
        _make("lib")
    import libpll_amd
    return libpll_amd.load()


@pytest.fixture(scope="session")
def gpu(amd):
    """The product library with a visible device; fails loudly without one."""
    n = amd.device_count()
    assert n > 0, "no HIP device visible: the gpu-marked tests must run on a GPU box"
    return amd


@pytest.fixture(scope="session")
def orc():
    if not os.path.exists(ORACLE):
        _make("oracle", os.path.join(ROOT, "oracle"))
    from oracle_api import Oracle
    return Oracle(ORACLE)


@pytest.fixture(scope="session")
def ref():
    """The genuine reference, built in place from /root/reference by
    oracle/Makefile (dev container) or shipped prebuilt (GPU box)."""
    if not os.path.exists(REF) and os.path.exists("/root/reference/src/pll.h"):
        _make("ref", os.path.join(ROOT, "oracle"))
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/libpll_ref.so not available")
    from libpll_amd.pllapi import PllLibrary
    return PllLibrary(REF)


@pytest.fixture(scope="session")
def ref_tree():
    """The reference's tree traversal / op-list builders (oracle/_ref/libpll_ref_tree.so)."""
    import ctypes
    path = os.path.join(ROOT, "oracle", "_ref", "libpll_ref_tree.so")
    if not os.path.exists(path) and os.path.exists("/root/reference/src/pll.h"):
        _make("ref", os.path.join(ROOT, "oracle"))
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libpll_ref_tree.so not available")

    class _L:
        lib = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
    return _L


@pytest.fixture(params=["exact", "mfma", "mfma-reference-order"])
def aa_mode(request, monkeypatch):
    """20-state kernels: bit-exact vector kernels (PLLHIP_AA_EXACT=1), the default matrix-core kernels (round 6: the
    whole-list kernel's tip-inner mat-vecs on the matrix cores too -- CLVs to rounding below such an op, scaler counts
    bit for bit behind the scaling certificate), or the matrix-core kernels with those mat-vecs in the reference's
    order (PLLHIP_AA_TI_MFMA=0: every CLV bit for bit).  Read when a partition is created."""
    monkeypatch.setenv("PLLHIP_AA_EXACT", "1" if request.param == "exact" else "0")
    monkeypatch.setenv("PLLHIP_AA_TI_MFMA", "1" if request.param == "mfma" else "0")
    # the table-lookup ops (used from 32 k sites on) also for the small test partitions
    monkeypatch.setenv("PLLHIP_AA_CHERRY", "2")
    return request.param


@pytest.fixture(params=["fused", "fused-8-waves", "levels"])
def dna_path(request, monkeypatch):
    """4-state CLV updates: the whole op list in one site-blocked launch (the default from
    ~16 k sites on; forced here, PLLHIP_FUSED=2) or one launch per dependency level
    (PLLHIP_FUSED=0); the whole-list kernel in its 12-wave (six LDS slots per wave) and 8-wave
    (seven slots) configurations.  All must give the same bits."""
    monkeypatch.setenv("PLLHIP_FUSED", "0" if request.param == "levels" else "2")
    monkeypatch.setenv("PLLHIP_FUSED_WGS", "2" if request.param == "fused-8-waves" else "3")
    return request.param

