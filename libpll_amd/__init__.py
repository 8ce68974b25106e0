"""libpll_amd -- MI355X-native Felsenstein-pruning hot path behind libpll's C API.

The product is the shared library ``libpll_amd/libpll_amd.so`` (C host code +
HIP kernels, see include/pll_amd.h and include/pllhip.h).  This package only
holds the ctypes binding used by the tests and bench.py, and the synthetic
workload generator.  There is no Python or CPU compute path: if the library is
missing, or no GPU is visible, creating a partition raises.
"""
import os

from .pllapi import PllLibrary, PllError  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# (PLL_AMD_LIB: another build of the same library, e.g. `make asan`'s sanitizer build)
LIB_PATH = os.environ.get("PLL_AMD_LIB") or os.path.join(_HERE, "libpll_amd.so")
_lib = None


def load():
    """The product library (loaded once)."""
    global _lib
    if _lib is None:
        _lib = PllLibrary(LIB_PATH)
    return _lib
