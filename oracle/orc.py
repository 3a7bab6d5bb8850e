"""ctypes wrapper of oracle/liborc.so -- TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Parity status of the oracle: see oracle/hpf_oracle.h
("parity unpinned" end to end; component pins listed there).
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
LIB_PATH = _DIR / "liborc.so"
REFPART = _DIR / "_ref" / "refpart"

STATE_NAMES = [
    "THETA_SHAPE", "THETA_RATE", "THETA_E", "THETA_ELOG",
    "BETA_SHAPE", "BETA_RATE", "BETA_E", "BETA_ELOG",
    "XI_SHAPE", "XI_RATE", "XI_E", "XI_ELOG",
    "ETA_SHAPE", "ETA_RATE", "ETA_E", "ETA_ELOG",
    "UBIAS_SHAPE", "UBIAS_RATE", "UBIAS_E", "UBIAS_ELOG",
    "IBIAS_SHAPE", "IBIAS_RATE", "IBIAS_E", "IBIAS_ELOG",
]
STATE = {n: i for i, n in enumerate(STATE_NAMES)}


class RunArgs(C.Structure):
    _fields_ = [
        ("datadir", C.c_char_p), ("outdir", C.c_char_p),
        ("n", C.c_uint32), ("m", C.c_uint32), ("k", C.c_uint32),
        ("hier", C.c_int), ("bias", C.c_int), ("binary", C.c_int),
        ("rating_threshold", C.c_uint32), ("rfreq", C.c_uint32),
        ("max_iterations", C.c_uint32), ("seed", C.c_double), ("logl", C.c_int), ("novb", C.c_int),
    ]


_lib = None


def build():
    subprocess.run(["make", "-C", str(_DIR), "liborc.so"], check=True, capture_output=True)


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        build()
    L = C.CDLL(str(LIB_PATH))
    vp, dp = C.c_void_p, C.POINTER(C.c_double)
    u32p, i32p = C.POINTER(C.c_uint32), C.POINTER(C.c_int32)
    L.orc_rng_seed.argtypes = [vp, C.c_ulong]
    L.orc_rng_u32.argtypes = [vp]
    L.orc_rng_u32.restype = C.c_uint32
    L.orc_rng_uniform.argtypes = [vp]
    L.orc_rng_uniform.restype = C.c_double
    L.orc_rng_uniform_int.argtypes = [vp, C.c_ulong]
    L.orc_rng_uniform_int.restype = C.c_ulong
    L.orc_psi.argtypes = [C.c_double]
    L.orc_psi.restype = C.c_double
    L.orc_logsum.argtypes = [dp, C.c_uint32]
    L.orc_logsum.restype = C.c_double
    L.orc_sum_strided.argtypes = [dp, C.c_uint32, C.c_size_t]
    L.orc_sum_strided.restype = C.c_double
    L.orc_lognormalize.argtypes = [dp, C.c_uint32]
    L.orc_ratings_new.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_uint32]
    L.orc_ratings_new.restype = vp
    L.orc_ratings_free.argtypes = [vp]
    L.orc_ratings_read_train.argtypes = [vp, C.c_char_p]
    L.orc_ratings_read_heldout.argtypes = [vp, C.c_char_p, C.c_int]
    for f in ("n", "m"):
        getattr(L, f"orc_ratings_{f}").argtypes = [vp]
        getattr(L, f"orc_ratings_{f}").restype = C.c_uint32
    L.orc_ratings_nnz.argtypes = [vp]
    L.orc_ratings_nnz.restype = C.c_uint64
    L.orc_ratings_rowptr.argtypes = [vp]
    L.orc_ratings_rowptr.restype = C.POINTER(C.c_int64)
    L.orc_ratings_col.argtypes = [vp]
    L.orc_ratings_col.restype = u32p
    L.orc_ratings_val.argtypes = [vp]
    L.orc_ratings_val.restype = C.POINTER(C.c_uint8)
    L.orc_ratings_seq2user.argtypes = [vp]
    L.orc_ratings_seq2user.restype = u32p
    L.orc_ratings_seq2item.argtypes = [vp]
    L.orc_ratings_seq2item.restype = u32p
    L.orc_ratings_heldout_count.argtypes = [vp, C.c_int]
    L.orc_ratings_heldout_count.restype = C.c_uint64
    for f, t in (("u", u32p), ("i", u32p), ("y", i32p)):
        getattr(L, f"orc_ratings_heldout_{f}").argtypes = [vp, C.c_int]
        getattr(L, f"orc_ratings_heldout_{f}").restype = t
    L.orc_ratings_write_marginals.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.orc_model_new.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int]
    L.orc_model_new.restype = vp
    L.orc_model_free.argtypes = [vp]
    L.orc_model_set_novb.argtypes = [vp, C.c_int]
    L.orc_model_set_csr.argtypes = [vp, C.POINTER(C.c_int64), u32p, C.POINTER(C.c_uint8)]
    L.orc_model_initialize.argtypes = [vp, C.c_double]
    L.orc_model_iterate.argtypes = [vp, C.c_int]
    L.orc_model_heldout_sum.argtypes = [vp, u32p, u32p, i32p, C.c_uint64]
    L.orc_model_heldout_sum.restype = C.c_double
    L.orc_model_iterate_all_cores.argtypes = [vp]
    L.orc_omp_threads.restype = C.c_int
    L.orc_model_elbo.argtypes = [vp]
    L.orc_model_elbo.restype = C.c_double
    L.orc_model_state.argtypes = [vp, C.c_int, C.POINTER(dp)]
    L.orc_model_state.restype = C.c_size_t
    L.orc_model_set_state.argtypes = [vp, C.c_int, dp, C.c_size_t]
    L.orc_model_rng.argtypes = [vp]
    L.orc_model_rng.restype = vp
    L.orc_run.argtypes = [C.POINTER(RunArgs)]
    L.orc_save_matrix.argtypes = [C.c_char_p, dp, C.c_uint32, C.c_uint32, u32p, C.c_uint32]
    L.orc_save_vector.argtypes = [C.c_char_p, dp, C.c_uint32, u32p, C.c_uint32]
    _lib = L
    return L


class Rng:
    def __init__(self, seed=0):
        self._buf = C.create_string_buffer(624 * 4 + 16)
        lib().orc_rng_seed(self._buf, int(seed))

    def u32(self):
        return lib().orc_rng_u32(self._buf)

    def uniform(self):
        return lib().orc_rng_uniform(self._buf)

    def uniform_int(self, n):
        return lib().orc_rng_uniform_int(self._buf, int(n))


def omp_threads():
    return lib().orc_omp_threads()


def psi(x):
    L = lib()
    return np.array([L.orc_psi(float(v)) for v in np.atleast_1d(x)])


def lognormalize(x):
    a = np.array(x, dtype=np.float64)
    lib().orc_lognormalize(a.ctypes.data_as(C.POINTER(C.c_double)), a.size)
    return a


def seq_sum(x, stride=1, n=None):
    a = np.ascontiguousarray(x, dtype=np.float64)
    n = a.size // stride if n is None else int(n)
    assert (n - 1) * stride < a.size or n == 0
    return lib().orc_sum_strided(a.ctypes.data_as(C.POINTER(C.c_double)), n, stride)


def sweep_steps(mode, snext_in, ev, u, v=0.0):
    """the model's own sweep steps (gp_set_prior_rate / gp_update_rate_next / gp_array_* / gp_swap of hpf_oracle.c) on
    caller arrays -> (scurr, rcurr, snext, rnext); mode as in orc_test_sweep_steps"""
    L = lib()
    snext_in = np.ascontiguousarray(snext_in, np.float64)
    rows, k = (snext_in.shape[0], 1) if snext_in.ndim == 1 else snext_in.shape
    ev = np.ascontiguousarray(ev, np.float64); u = np.ascontiguousarray(u, np.float64)
    nr = k if mode == 1 else rows * k
    scurr, snext = np.empty(rows * k), np.empty(rows * k)
    rcurr, rnext = np.empty(nr), np.empty(nr)
    dp = C.POINTER(C.c_double)
    L.orc_test_sweep_steps.restype = None
    L.orc_test_sweep_steps.argtypes = [C.c_int, C.c_uint32, C.c_uint32, dp, dp, dp, C.c_double, dp, dp, dp, dp]
    L.orc_test_sweep_steps(mode, rows, k, snext_in.ctypes.data_as(dp), ev.ctypes.data_as(dp), u.ctypes.data_as(dp), float(v),
                           scurr.ctypes.data_as(dp), rcurr.ctypes.data_as(dp), snext.ctypes.data_as(dp), rnext.ctypes.data_as(dp))
    return scurr, rcurr, snext, rnext


def logsum(x):
    a = np.ascontiguousarray(x, dtype=np.float64)
    return lib().orc_logsum(a.ctypes.data_as(C.POINTER(C.c_double)), a.size)


class Ratings:
    def __init__(self, cap_n, cap_m, binary=False, rating_threshold=1):
        self.L = lib()
        self._r = C.c_void_p(self.L.orc_ratings_new(cap_n, cap_m, int(binary), rating_threshold))

    def read_train(self, path):
        return self.L.orc_ratings_read_train(self._r, str(path).encode())

    def read_heldout(self, path, which):
        return self.L.orc_ratings_read_heldout(self._r, str(path).encode(), which)

    @property
    def n(self):
        return self.L.orc_ratings_n(self._r)

    @property
    def m(self):
        return self.L.orc_ratings_m(self._r)

    @property
    def nnz(self):
        return self.L.orc_ratings_nnz(self._r)

    def csr(self):
        n, nnz = self.n, self.nnz
        rp = np.ctypeslib.as_array(self.L.orc_ratings_rowptr(self._r), shape=(n + 1,)).copy()
        if nnz:
            col = np.ctypeslib.as_array(self.L.orc_ratings_col(self._r), shape=(nnz,)).copy()
            val = np.ctypeslib.as_array(self.L.orc_ratings_val(self._r), shape=(nnz,)).copy()
        else:
            col = np.zeros(0, np.uint32)
            val = np.zeros(0, np.uint8)
        return rp, col, val

    def seq2user(self):
        return np.ctypeslib.as_array(self.L.orc_ratings_seq2user(self._r), shape=(max(self.n, 1),))[: self.n].copy()

    def seq2item(self):
        return np.ctypeslib.as_array(self.L.orc_ratings_seq2item(self._r), shape=(max(self.m, 1),))[: self.m].copy()

    def heldout(self, which):
        c = self.L.orc_ratings_heldout_count(self._r, which)
        if c == 0:
            return np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.int32)
        u = np.ctypeslib.as_array(self.L.orc_ratings_heldout_u(self._r, which), shape=(c,)).copy()
        i = np.ctypeslib.as_array(self.L.orc_ratings_heldout_i(self._r, which), shape=(c,)).copy()
        y = np.ctypeslib.as_array(self.L.orc_ratings_heldout_y(self._r, which), shape=(c,)).copy()
        return u, i, y

    def write_marginals(self, byusers, byitems):
        return self.L.orc_ratings_write_marginals(self._r, str(byusers).encode(), str(byitems).encode())

    def __del__(self):
        try:
            self.L.orc_ratings_free(self._r)
        except Exception:
            pass


class Model:
    """CPU restatement of HGAPRec (vb_hier / vb / vb_bias)."""

    def __init__(self, n, m, K, hier=True, bias=False, binary=False, novb=False):
        self.L = lib()
        self.n, self.m, self.K = n, m, K
        self.hier, self.bias, self.binary = hier, bias, binary
        self._m = C.c_void_p(self.L.orc_model_new(n, m, K, int(hier), int(bias), int(binary)))
        self.L.orc_model_set_novb(self._m, int(bool(novb)))       # Env::vb = !novb; vb_bias() alone reads it
        self._keep = None

    def set_csr(self, rowptr, col, val=None):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        col = np.ascontiguousarray(col, dtype=np.uint32)
        v = None if val is None else np.ascontiguousarray(val, dtype=np.uint8)
        self._keep = (rowptr, col, v)
        self.L.orc_model_set_csr(
            self._m, rowptr.ctypes.data_as(C.POINTER(C.c_int64)),
            col.ctypes.data_as(C.POINTER(C.c_uint32)),
            None if v is None else v.ctypes.data_as(C.POINTER(C.c_uint8)))

    def initialize(self, seed=0.0):
        self.L.orc_model_initialize(self._m, float(seed))

    def iterate(self, n=1):
        self.L.orc_model_iterate(self._m, int(n))

    def heldout_sum(self, u, i, y):
        u = np.ascontiguousarray(u, dtype=np.uint32)
        i = np.ascontiguousarray(i, dtype=np.uint32)
        y = np.ascontiguousarray(y, dtype=np.int32)
        return self.L.orc_model_heldout_sum(
            self._m, u.ctypes.data_as(C.POINTER(C.c_uint32)),
            i.ctypes.data_as(C.POINTER(C.c_uint32)),
            y.ctypes.data_as(C.POINTER(C.c_int32)), u.size)

    def iterate_all_cores(self):
        self.L.orc_model_iterate_all_cores(self._m)

    def elbo(self):
        return self.L.orc_model_elbo(self._m)

    def _shape(self, which):
        obj, kind = STATE[which] // 4, STATE[which] % 4
        if obj == 0:
            s = (self.n, self.K)
        elif obj == 1:
            s = (self.m, self.K)
        elif obj in (2, 4):
            s = (self.n,)
        else:
            s = (self.m,)
        if obj <= 1 and kind == 1 and not self.hier:
            s = (self.K,)
        return s

    def state(self, which) -> np.ndarray:
        p = C.POINTER(C.c_double)()
        cnt = self.L.orc_model_state(self._m, STATE[which], C.byref(p))
        if cnt == 0:
            raise KeyError(which)
        return np.ctypeslib.as_array(p, shape=(cnt,)).copy().reshape(self._shape(which))

    def set_state(self, which, arr):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        rc = self.L.orc_model_set_state(self._m, STATE[which],
                                        a.ctypes.data_as(C.POINTER(C.c_double)), a.size)
        if rc:
            raise ValueError(which)

    def rng_u32(self):
        return self.L.orc_rng_u32(self.L.orc_model_rng(self._m))

    def __del__(self):
        try:
            self.L.orc_model_free(self._m)
        except Exception:
            pass


def run(datadir, outdir, n, m, k, hier=True, bias=False, binary=False,
        rating_threshold=1, rfreq=10, max_iterations=1000, seed=0.0, logl=False, novb=False):
    a = RunArgs(str(datadir).encode(), str(outdir).encode(), n, m, k, int(hier), int(bias),
                int(binary), rating_threshold, rfreq, max_iterations, float(seed), int(logl), int(bool(novb)))
    return lib().orc_run(C.byref(a))


def save_matrix(path, a, seq2id=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    ids = None if seq2id is None else np.ascontiguousarray(seq2id, dtype=np.uint32)
    return lib().orc_save_matrix(
        str(path).encode(), a.ctypes.data_as(C.POINTER(C.c_double)), a.shape[0], a.shape[1],
        None if ids is None else ids.ctypes.data_as(C.POINTER(C.c_uint32)),
        0 if ids is None else ids.size)


def save_vector(path, a, seq2id=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    ids = None if seq2id is None else np.ascontiguousarray(seq2id, dtype=np.uint32)
    return lib().orc_save_vector(
        str(path).encode(), a.ctypes.data_as(C.POINTER(C.c_double)), a.shape[0],
        None if ids is None else ids.ctypes.data_as(C.POINTER(C.c_uint32)),
        0 if ids is None else ids.size)
