"""ctypes binding of the pll.h-shaped C API.

The same binding drives two different shared libraries with the same ABI:

  * libpll_amd/libpll_amd.so -- the product (HIP kernels, include/pll_amd.h);
  * oracle/_ref/libpll_ref.so -- the reference built in place by oracle/Makefile
    (only tests/ and bench.py's cpu_baseline leg load that one).

Nothing here computes: every method is one C call, with the reference's names
(pll.h:530-664) minus the ``pll_`` prefix, so the parity tests read like the
reference's own test programs.
"""
import ctypes as C
import os

import numpy as np

SCALE_BUFFER_NONE = -1
ATTRIB_ARCH_CPU = 0
ERROR_PARAM_INVALID = 113   # pll.h:159
ATTRIB_ARCH_SSE = 1 << 0
ATTRIB_ARCH_AVX = 1 << 1
ATTRIB_ARCH_AVX2 = 1 << 2
ATTRIB_PATTERN_TIP = 1 << 4
ATTRIB_AB_LEWIS = 1 << 5
ATTRIB_AB_FELSENSTEIN = 2 << 5
ATTRIB_AB_STAMATAKIS = 3 << 5
ATTRIB_AB_FLAG = 1 << 8
ATTRIB_RATE_SCALERS = 1 << 9
ATTRIB_SITE_REPEATS = 1 << 10   # not in libpll 0.3.2: own extension, see host/repeats.c
GAMMA_RATES_MEAN = 0
GAMMA_RATES_MEDIAN = 1


class PartitionStruct(C.Structure):
    """pll_partition_t, pll.h:202-244 / include/pll_amd.h."""
    _fields_ = [
        ("tips", C.c_uint), ("clv_buffers", C.c_uint), ("states", C.c_uint),
        ("sites", C.c_uint), ("pattern_weight_sum", C.c_uint),
        ("rate_matrices", C.c_uint), ("prob_matrices", C.c_uint),
        ("rate_cats", C.c_uint), ("scale_buffers", C.c_uint), ("attributes", C.c_uint),
        ("alignment", C.c_size_t), ("states_padded", C.c_uint),
        ("clv", C.POINTER(C.POINTER(C.c_double))),
        ("pmatrix", C.POINTER(C.POINTER(C.c_double))),
        ("rates", C.POINTER(C.c_double)), ("rate_weights", C.POINTER(C.c_double)),
        ("subst_params", C.POINTER(C.POINTER(C.c_double))),
        ("scale_buffer", C.POINTER(C.POINTER(C.c_uint))),
        ("frequencies", C.POINTER(C.POINTER(C.c_double))),
        ("prop_invar", C.POINTER(C.c_double)), ("invariant", C.POINTER(C.c_int)),
        ("pattern_weights", C.POINTER(C.c_uint)),
        ("eigen_decomp_valid", C.POINTER(C.c_int)),
        ("eigenvecs", C.POINTER(C.POINTER(C.c_double))),
        ("inv_eigenvecs", C.POINTER(C.POINTER(C.c_double))),
        ("eigenvals", C.POINTER(C.POINTER(C.c_double))),
        ("maxstates", C.c_uint),
        ("tipchars", C.POINTER(C.POINTER(C.c_ubyte))),
        ("charmap", C.POINTER(C.c_ubyte)), ("ttlookup", C.POINTER(C.c_double)),
        ("tipmap", C.POINTER(C.c_uint)), ("asc_bias_alloc", C.c_int),
    ]


class Operation(C.Structure):
    """pll_operation_t, pll.h:249-259."""
    _fields_ = [
        ("parent_clv_index", C.c_uint), ("parent_scaler_index", C.c_int),
        ("child1_clv_index", C.c_uint), ("child1_matrix_index", C.c_uint),
        ("child1_scaler_index", C.c_int),
        ("child2_clv_index", C.c_uint), ("child2_matrix_index", C.c_uint),
        ("child2_scaler_index", C.c_int),
    ]


OPS_DTYPE = np.dtype([
    ("parent_clv_index", "<u4"), ("parent_scaler_index", "<i4"),
    ("child1_clv_index", "<u4"), ("child1_matrix_index", "<u4"),
    ("child1_scaler_index", "<i4"),
    ("child2_clv_index", "<u4"), ("child2_matrix_index", "<u4"),
    ("child2_scaler_index", "<i4")])

_PP = C.POINTER(PartitionStruct)
_dp = C.POINTER(C.c_double)
_up = C.POINTER(C.c_uint)


def _d(a):
    return a.ctypes.data_as(_dp)


def _u(a):
    return a.ctypes.data_as(_up)


class PllError(RuntimeError):
    pass


class PllLibrary:
    """One loaded shared library exporting the pll_* API."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise PllError("shared library %s is missing -- build it first "
                           "(python -c 'import __graft_entry__ as g; g.build()')" % path)
        self.path = path
        self.lib = lib = C.CDLL(path, mode=C.RTLD_LOCAL)
        self.is_amd = hasattr(lib, "pll_amd_device_count")
        f = lib.pll_partition_create
        f.restype = _PP
        f.argtypes = [C.c_uint] * 9
        lib.pll_partition_destroy.restype = None
        lib.pll_partition_destroy.argtypes = [_PP]
        lib.pll_set_tip_states.argtypes = [_PP, C.c_uint, _up, C.c_char_p]
        lib.pll_set_tip_clv.argtypes = [_PP, C.c_uint, _dp, C.c_int]
        lib.pll_set_pattern_weights.restype = None
        lib.pll_set_pattern_weights.argtypes = [_PP, _up]
        for name in ("pll_set_subst_params", "pll_set_frequencies"):
            g = getattr(lib, name)
            g.restype = None
            g.argtypes = [_PP, C.c_uint, _dp]
        for name in ("pll_set_category_rates", "pll_set_category_weights"):
            g = getattr(lib, name)
            g.restype = None
            g.argtypes = [_PP, _dp]
        lib.pll_update_eigen.argtypes = [_PP, C.c_uint]
        lib.pll_update_prob_matrices.argtypes = [_PP, _up, _up, _dp, C.c_uint]
        lib.pll_update_invariant_sites.argtypes = [_PP]
        lib.pll_update_invariant_sites_proportion.argtypes = [_PP, C.c_uint, C.c_double]
        lib.pll_update_partials.restype = None
        lib.pll_update_partials.argtypes = [_PP, C.c_void_p, C.c_uint]
        lib.pll_compute_edge_loglikelihood.restype = C.c_double
        lib.pll_compute_edge_loglikelihood.argtypes = [_PP, C.c_uint, C.c_int, C.c_uint, C.c_int,
                                                       C.c_uint, _up, _dp]
        lib.pll_compute_root_loglikelihood.restype = C.c_double
        lib.pll_compute_root_loglikelihood.argtypes = [_PP, C.c_uint, C.c_int, _up, _dp]
        lib.pll_update_sumtable.argtypes = [_PP, C.c_uint, C.c_uint, C.c_int, C.c_int, _up, _dp]
        lib.pll_compute_likelihood_derivatives.argtypes = [_PP, C.c_int, C.c_int, C.c_double, _up,
                                                           _dp, _dp, _dp]
        lib.pll_compute_gamma_cats.argtypes = [C.c_double, C.c_uint, _dp, C.c_int]
        lib.pll_aligned_alloc.restype = C.c_void_p
        lib.pll_aligned_alloc.argtypes = [C.c_size_t, C.c_size_t]
        lib.pll_aligned_free.restype = None
        lib.pll_aligned_free.argtypes = [C.c_void_p]
        if self.is_amd:
            lib.pll_amd_sync_clv.argtypes = [_PP, C.c_uint]
            lib.pll_amd_sync_scaler.argtypes = [_PP, C.c_uint]
            lib.pll_amd_sync_pmatrix.argtypes = [_PP, C.c_uint]
            lib.pll_amd_sync_sumtable.argtypes = [_PP, _dp]
            lib.pll_amd_forget_sumtable.argtypes = [_PP, _dp]
            lib.pll_amd_wait.argtypes = [_PP]
            lib.pll_amd_set_devices.argtypes = [C.POINTER(C.c_int), C.c_uint]
            lib.pll_amd_shard_count.argtypes = [_PP]
            lib.pll_amd_shard_count.restype = C.c_uint
            lib.pll_amd_timer_start.argtypes = [_PP]
            lib.pll_amd_timer_stop_ms.argtypes = [_PP, C.POINTER(C.c_float)]
            lib.pll_amd_timer_shard_ms.argtypes = [_PP, C.POINTER(C.c_float), C.c_uint]
            lib.pll_amd_timer_shard_ms.restype = C.c_uint
            lib.pll_amd_comm_unique_id.argtypes = [C.c_void_p]
            lib.pll_amd_comm_init.argtypes = [_PP, C.c_int, C.c_int, C.c_void_p]
            if hasattr(lib, "pll_amd_arena_fill_bandwidth"):
                lib.pll_amd_arena_fill_bandwidth.argtypes = [_PP, C.POINTER(C.c_double)]
                lib.pll_amd_placement_info.argtypes = [_PP, C.POINTER(C.c_double), C.c_uint, C.POINTER(C.c_int)]
            if hasattr(lib, "pll_amd_comm_reduces"):
                lib.pll_amd_comm_reduces.argtypes = [_PP]
                lib.pll_amd_comm_reduces.restype = C.c_ulonglong
            lib.pll_amd_profile_enable.argtypes = [_PP, C.c_int]
            lib.pll_amd_profile_read.argtypes = [_PP, _up, _dp]
            if hasattr(lib, "pll_amd_scaling_certificate"):
                lib.pll_amd_scaling_certificate.argtypes = [_PP, C.POINTER(C.c_ulonglong)]
            if hasattr(lib, "pll_amd_write_ceiling"):   # (older builds under PLL_AMD_LIB: tools/list_time.py)
                lib.pll_amd_write_ceiling.argtypes = [_PP, C.c_void_p, C.c_uint, C.c_uint, C.POINTER(C.c_float),
                                                      C.POINTER(C.c_double)]
                lib.pll_amd_list_kinds.argtypes = [_PP, _up]
            lib.pll_amd_eigen_decompose.argtypes = [C.c_uint, _dp, _dp, _dp, _dp, _dp]

    # -- library-level helpers -------------------------------------------------
    def errno(self):
        return C.c_int.in_dll(self.lib, "pll_errno").value

    def clear_error(self):
        C.c_int.in_dll(self.lib, "pll_errno").value = 0

    def errmsg(self):
        return C.string_at(C.addressof((C.c_char * 200).in_dll(self.lib, "pll_errmsg"))).decode()

    def map(self, name):
        """pll_map_nt / pll_map_aa / pll_map_bin as a uint32[256] array."""
        return np.ctypeslib.as_array((C.c_uint * 256).in_dll(self.lib, "pll_map_" + name)).copy()

    def aa_model(self, name):
        r = np.ctypeslib.as_array((C.c_double * 190).in_dll(self.lib, "pll_aa_rates_" + name)).copy()
        f = np.ctypeslib.as_array((C.c_double * 20).in_dll(self.lib, "pll_aa_freqs_" + name)).copy()
        return r, f

    def compute_gamma_cats(self, alpha, cats, mode=GAMMA_RATES_MEAN):
        out = np.zeros(cats)
        if not self.lib.pll_compute_gamma_cats(alpha, cats, _d(out), mode):
            raise PllError(self.errmsg())
        return out

    def device_count(self):
        return self.lib.pll_amd_device_count() if self.is_amd else 0

    def partition_create(self, tips, clv_buffers, states, sites, rate_matrices, prob_matrices,
                         rate_cats, scale_buffers, attributes):
        p = self.lib.pll_partition_create(tips, clv_buffers, states, sites, rate_matrices,
                                          prob_matrices, rate_cats, scale_buffers, attributes)
        if not p:
            raise PllError("pll_partition_create failed (pll_errno=%d): %s"
                           % (self.errno(), self.errmsg()))
        return Partition(self, p)


class Partition:
    """A pll_partition_t* plus the calls that take it as first argument."""

    def __init__(self, owner, ptr):
        self.o = owner
        self.lib = owner.lib
        self.ptr = ptr
        self.s = ptr.contents
        self._keep = []

    def destroy(self):
        if self.ptr:
            self.lib.pll_partition_destroy(self.ptr)
            self.ptr = None

    def _check(self, ok, what):
        if not ok:
            raise PllError("%s failed (pll_errno=%d): %s" % (what, self.o.errno(), self.o.errmsg()))

    @property
    def span(self):
        return self.s.rate_cats * self.s.states_padded

    @property
    def sites_total(self):
        """alignment sites + the `states` ascertainment sites, if allocated (pll.c:492-495)"""
        return self.s.sites + (self.s.states if self.s.asc_bias_alloc else 0)

    @property
    def scaler_len(self):
        per = self.s.rate_cats if (self.s.attributes & ATTRIB_RATE_SCALERS) else 1
        return self.sites_total * per

    # -- setters -----------------------------------------------------------------
    def set_tip_states(self, tip, cmap, seq):
        cmap = np.ascontiguousarray(cmap, dtype=np.uint32)
        if isinstance(seq, str):
            seq = seq.encode()
        self._check(self.lib.pll_set_tip_states(self.ptr, tip, _u(cmap), seq), "pll_set_tip_states")

    def set_tip_clv(self, tip, clv, padding=0):
        clv = np.ascontiguousarray(clv, dtype=np.float64)
        self._check(self.lib.pll_set_tip_clv(self.ptr, tip, _d(clv), padding), "pll_set_tip_clv")

    def set_pattern_weights(self, w):
        w = np.ascontiguousarray(w, dtype=np.uint32)
        self.lib.pll_set_pattern_weights(self.ptr, _u(w))

    def repeats_classes(self, clv_index):
        """rows the CLV is stored in under PLL_ATTRIB_SITE_REPEATS (0 = one per site)"""
        return int(self.lib.pll_amd_repeats_classes(self.ptr, clv_index))

    def set_asc_bias_type(self, asc_type):
        self._check(self.lib.pll_set_asc_bias_type(self.ptr, asc_type), "pll_set_asc_bias_type")

    def set_asc_state_weights(self, w):
        w = np.ascontiguousarray(w, dtype=np.uint32)
        assert len(w) == self.s.states
        self.lib.pll_set_asc_state_weights(self.ptr, _u(w))

    def set_subst_params(self, idx, params):
        a = np.ascontiguousarray(params, dtype=np.float64)
        self.lib.pll_set_subst_params(self.ptr, idx, _d(a))

    def set_frequencies(self, idx, freqs):
        a = np.ascontiguousarray(freqs, dtype=np.float64)
        self.lib.pll_set_frequencies(self.ptr, idx, _d(a))

    def set_category_rates(self, rates):
        a = np.ascontiguousarray(rates, dtype=np.float64)
        self.lib.pll_set_category_rates(self.ptr, _d(a))

    def set_category_weights(self, w):
        a = np.ascontiguousarray(w, dtype=np.float64)
        self.lib.pll_set_category_weights(self.ptr, _d(a))

    def update_eigen(self, idx):
        self._check(self.lib.pll_update_eigen(self.ptr, idx), "pll_update_eigen")

    def update_invariant_sites(self):
        self._check(self.lib.pll_update_invariant_sites(self.ptr), "pll_update_invariant_sites")

    def update_invariant_sites_proportion(self, idx, pinv):
        self._check(self.lib.pll_update_invariant_sites_proportion(self.ptr, idx, pinv),
                    "pll_update_invariant_sites_proportion")

    # -- the hot path ---------------------------------------------------------------
    def update_prob_matrices(self, params_indices, matrix_indices, branch_lengths):
        pi = np.ascontiguousarray(params_indices, dtype=np.uint32)
        mi = np.ascontiguousarray(matrix_indices, dtype=np.uint32)
        bl = np.ascontiguousarray(branch_lengths, dtype=np.float64)
        assert len(pi) == self.s.rate_cats and len(mi) == len(bl)
        self._check(self.lib.pll_update_prob_matrices(self.ptr, _u(pi), _u(mi), _d(bl), len(mi)),
                    "pll_update_prob_matrices")

    def update_partials(self, ops):
        ops = np.ascontiguousarray(ops, dtype=OPS_DTYPE)
        self.lib.pll_update_partials(self.ptr, ops.ctypes.data, len(ops))

    def compute_edge_loglikelihood(self, pclv, pscaler, cclv, cscaler, matrix, freqs_indices,
                                   persite=False):
        fi = np.ascontiguousarray(freqs_indices, dtype=np.uint32)
        ps = np.zeros(self.s.sites) if persite else None
        v = self.lib.pll_compute_edge_loglikelihood(self.ptr, pclv, pscaler, cclv, cscaler, matrix,
                                                    _u(fi), _d(ps) if persite else None)
        return (v, ps) if persite else v

    def compute_root_loglikelihood(self, clv, scaler, freqs_indices, persite=False):
        fi = np.ascontiguousarray(freqs_indices, dtype=np.uint32)
        ps = np.zeros(self.s.sites) if persite else None
        v = self.lib.pll_compute_root_loglikelihood(self.ptr, clv, scaler, _u(fi),
                                                    _d(ps) if persite else None)
        return (v, ps) if persite else v

    def alloc_sumtable(self):
        """A caller-owned, aligned host sumtable like test/src/scaling.c:215-218 allocates."""
        n = self.sites_total * self.span
        raw = self.lib.pll_aligned_alloc(n * 8, 32)
        arr = np.ctypeslib.as_array(C.cast(raw, _dp), shape=(n,))
        self._keep.append(raw)
        return arr

    def update_sumtable(self, pclv, cclv, pscaler, cscaler, params_indices, sumtable):
        pi = np.ascontiguousarray(params_indices, dtype=np.uint32)
        self._check(self.lib.pll_update_sumtable(self.ptr, pclv, cclv, pscaler, cscaler, _u(pi),
                                                 _d(sumtable)), "pll_update_sumtable")

    def compute_likelihood_derivatives(self, pscaler, cscaler, t, params_indices, sumtable):
        pi = np.ascontiguousarray(params_indices, dtype=np.uint32)
        d = C.c_double()
        dd = C.c_double()
        self._check(self.lib.pll_compute_likelihood_derivatives(
            self.ptr, pscaler, cscaler, t, _u(pi), _d(sumtable), C.byref(d), C.byref(dd)),
            "pll_compute_likelihood_derivatives")
        return d.value, dd.value

    # -- reading results back ----------------------------------------------------------
    def get_clv(self, idx):
        if self.o.is_amd:
            self._check(self.lib.pll_amd_sync_clv(self.ptr, idx), "pll_amd_sync_clv")
        n = self.sites_total * self.span
        return np.ctypeslib.as_array(self.s.clv[idx], shape=(n,)).copy().reshape(
            self.sites_total, self.s.rate_cats, self.s.states_padded)[:, :, :self.s.states]

    def get_scaler(self, idx):
        if self.o.is_amd:
            self._check(self.lib.pll_amd_sync_scaler(self.ptr, idx), "pll_amd_sync_scaler")
        return np.ctypeslib.as_array(self.s.scale_buffer[idx], shape=(self.scaler_len,)).copy()

    def get_pmatrix(self, idx):
        if self.o.is_amd:
            self._check(self.lib.pll_amd_sync_pmatrix(self.ptr, idx), "pll_amd_sync_pmatrix")
        S, SP, R = self.s.states, self.s.states_padded, self.s.rate_cats
        return np.ctypeslib.as_array(self.s.pmatrix[idx], shape=(R * S * SP,)).copy().reshape(
            R, S, SP)[:, :, :S]

    def get_sumtable(self, sumtable):
        if self.o.is_amd:
            self._check(self.lib.pll_amd_sync_sumtable(self.ptr, _d(sumtable)),
                        "pll_amd_sync_sumtable")
        return np.array(sumtable).reshape(self.sites_total, self.s.rate_cats,
                                          self.s.states_padded)[:, :, :self.s.states]

    def get_eigen(self, idx):
        S, SP = self.s.states, self.s.states_padded
        vals = np.ctypeslib.as_array(self.s.eigenvals[idx], shape=(SP,)).copy()[:S]
        vecs = np.ctypeslib.as_array(self.s.eigenvecs[idx], shape=(S * SP,)).copy().reshape(S, SP)[:, :S]
        inv = np.ctypeslib.as_array(self.s.inv_eigenvecs[idx], shape=(S * SP,)).copy().reshape(S, SP)[:, :S]
        return vals, vecs, inv

    # -- libpll_amd additions ------------------------------------------------------------
    def wait(self):
        if self.o.is_amd:
            self._check(self.lib.pll_amd_wait(self.ptr), "pll_amd_wait")

    def timer_start(self):
        self._check(self.lib.pll_amd_timer_start(self.ptr), "pll_amd_timer_start")

    def timer_stop_ms(self):
        ms = C.c_float()
        self._check(self.lib.pll_amd_timer_stop_ms(self.ptr, C.byref(ms)), "pll_amd_timer_stop_ms")
        return ms.value

    def shard_ms(self):
        """what the last timer_stop_ms measured on each shard's own stream (one entry if unsharded)"""
        buf = (C.c_float * 64)()
        n = self.lib.pll_amd_timer_shard_ms(self.ptr, buf, 64)
        return [float(buf[i]) for i in range(min(n, 64))]

    def profile_enable(self, on=True):
        self._check(self.lib.pll_amd_profile_enable(self.ptr, 1 if on else 0), "pll_amd_profile_enable")

    def profile_read(self):
        """{kind: (launches, total_ms)} since the last read."""
        n = np.zeros(7, dtype=np.uint32)
        ms = np.zeros(7)
        self._check(self.lib.pll_amd_profile_read(self.ptr, _u(n), _d(ms)), "pll_amd_profile_read")
        names = ("partials_ii", "partials_ti", "partials_tt", "lnl", "sumtable", "derivatives",
                 "pmatrix")
        return {k: (int(n[i]), float(ms[i])) for i, k in enumerate(names)}

    def scaling_certificate(self):
        """{lists, raised, rerun, uncertified} of the 20-state scaling certificate (pll_amd.h)."""
        buf = (C.c_ulonglong * 4)()
        self._check(self.lib.pll_amd_scaling_certificate(self.ptr, buf), "pll_amd_scaling_certificate")
        return dict(zip(("lists", "raised", "rerun", "uncertified"), (int(v) for v in buf)))

    def write_ceiling(self, ops, reps):
        """(ms per pass, bytes per pass) of nothing but the stores of `ops` -- OVERWRITES their CLVs (pll_amd.h)."""
        ops = np.ascontiguousarray(ops, dtype=OPS_DTYPE)
        ms, nbytes = C.c_float(), C.c_double()
        self._check(self.lib.pll_amd_write_ceiling(self.ptr, C.c_void_p(ops.ctypes.data), len(ops), reps,
                                                   C.byref(ms), C.byref(nbytes)), "pll_amd_write_ceiling")
        return ms.value, nbytes.value

    def list_kinds(self):
        """what the 20-state whole-list kernel made of the last list it planned (pll_amd.h)"""
        v = np.zeros(8, dtype=np.uint32)
        self._check(self.lib.pll_amd_list_kinds(self.ptr, _u(v)), "pll_amd_list_kinds")
        return dict(zip(("ops", "tip_tip_ahead", "tip_tip_in_list", "lookups", "inner_inner_matrix_cores",
                         "tip_inner_matrix_cores", "tip_inner_vector_unit", "reloads"), (int(x) for x in v)))

    def placement(self):
        """where the CLV arena lies: {"tried": n, "kept": i, "GBs": [write rate of each place tried]} (pll_amd.h)"""
        g = (C.c_double * 32)()
        kept = C.c_int(0)
        n = self.lib.pll_amd_placement_info(self.ptr, g, 32, C.byref(kept))
        return {"tried": int(n), "kept": int(kept.value), "GBs": [round(g[i], 1) for i in range(min(n, 32))]}

    def arena_fill_bandwidth(self):
        """GB/s of one timed zeroing pass over the CLV arena -- OVERWRITES every CLV (pll_amd.h)."""
        g = C.c_double()
        self._check(self.lib.pll_amd_arena_fill_bandwidth(self.ptr, C.byref(g)), "pll_amd_arena_fill_bandwidth")
        return g.value

    def comm_reduces(self):
        """collectives entered so far (every rank of a job must count alike)"""
        return int(self.lib.pll_amd_comm_reduces(self.ptr))

    def comm_init(self, rank, nranks, unique_id):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._check(self.lib.pll_amd_comm_init(self.ptr, rank, nranks, buf), "pll_amd_comm_init")
