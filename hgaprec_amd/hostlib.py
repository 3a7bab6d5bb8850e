"""ctypes binding of libhgaprec_host.so -- the C++ host side of the reference
interface (CLI/Env naming, TSV reader -> CSR, MT19937 start state, TSV
writers, stop rule).  No HIP involved; used by the CPU tests and by Python
callers that want the reference's file semantics without the CLI."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
# HGAPREC_HOST_LIB: load another build of the same library (the ASan/UBSan one,
# tests/test_host_sanitizers.py)
LIB_PATH = Path(os.environ.get("HGAPREC_HOST_LIB") or _PKG / "libhgaprec_host.so")
_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(f"{LIB_PATH} not found: run __graft_entry__.build()")
    L = C.CDLL(str(LIB_PATH))
    vp, dp, u32p = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint32)
    L.hg_prefix.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    L.hg_open_output.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_char_p, C.c_size_t]
    L.hg_ratings_new.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_uint32]
    L.hg_ratings_new.restype = vp
    L.hg_ratings_free.argtypes = [vp]
    L.hg_ratings_read_train.argtypes = [vp, C.c_char_p]
    L.hg_ratings_read_heldout.argtypes = [vp, C.c_char_p, C.c_int]
    L.hg_ratings_n.argtypes = [vp]; L.hg_ratings_n.restype = C.c_uint32
    L.hg_ratings_m.argtypes = [vp]; L.hg_ratings_m.restype = C.c_uint32
    L.hg_ratings_nnz.argtypes = [vp]; L.hg_ratings_nnz.restype = C.c_uint64
    L.hg_ratings_rowptr.argtypes = [vp]; L.hg_ratings_rowptr.restype = C.POINTER(C.c_int64)
    L.hg_ratings_col.argtypes = [vp]; L.hg_ratings_col.restype = u32p
    L.hg_ratings_val.argtypes = [vp]; L.hg_ratings_val.restype = C.POINTER(C.c_uint8)
    L.hg_ratings_seq2user.argtypes = [vp]; L.hg_ratings_seq2user.restype = u32p
    L.hg_ratings_seq2item.argtypes = [vp]; L.hg_ratings_seq2item.restype = u32p
    L.hg_ratings_heldout_count.argtypes = [vp, C.c_int]; L.hg_ratings_heldout_count.restype = C.c_uint64
    L.hg_ratings_heldout_u.argtypes = [vp, C.c_int]; L.hg_ratings_heldout_u.restype = u32p
    L.hg_ratings_heldout_i.argtypes = [vp, C.c_int]; L.hg_ratings_heldout_i.restype = u32p
    L.hg_ratings_heldout_y.argtypes = [vp, C.c_int]; L.hg_ratings_heldout_y.restype = C.POINTER(C.c_int32)
    L.hg_ratings_write_marginals.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.hg_ratings_save_cache.argtypes = [vp, C.c_char_p]
    L.hg_ratings_load_cache.argtypes = [vp, C.c_char_p]
    L.hg_ratings_test_users.argtypes = [vp, C.c_char_p, u32p, C.c_uint32]
    L.hg_mt_u32.argtypes = [C.c_double, C.c_uint32, u32p]
    L.hg_digamma.argtypes = [C.c_double]; L.hg_digamma.restype = C.c_double
    L.hg_state_new.argtypes = [C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int]
    L.hg_state_new.restype = vp
    L.hg_state_new_range.argtypes = [C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_uint32,
                                     C.POINTER(C.c_uint32)]
    L.hg_state_new_range.restype = vp
    L.hg_state_free.argtypes = [vp]
    L.hg_state_get.argtypes = [vp, C.c_int, C.POINTER(dp)]; L.hg_state_get.restype = C.c_size_t
    L.hg_save_matrix.argtypes = [C.c_char_p, dp, C.c_uint32, C.c_uint32, u32p, C.c_uint32]
    L.hg_save_vector.argtypes = [C.c_char_p, dp, C.c_uint32, u32p, C.c_uint32]
    L.hg_format_fixed8.argtypes = [dp, C.c_size_t, C.c_char_p]
    L.hg_format_fixed8.restype = C.c_size_t
    L.hg_partition_users.argtypes = [C.POINTER(C.c_int64), C.c_uint32, C.c_int, u32p]
    L.hg_partition_users.restype = None
    L.hg_stop_rule.argtypes = [u32p, dp, C.c_uint32, C.POINTER(C.c_int)]
    _lib = L
    return L


def _argv(args):
    arr = (C.c_char_p * len(args))(*[str(a).encode() for a in args])
    return len(args), arr


def prefix(args) -> str:
    """output-directory name for a CLI argument list (Env::prefix)"""
    n, arr = _argv(args)
    out, bad = C.create_string_buffer(4096), C.create_string_buffer(256)
    rc = lib().hg_prefix(n, arr, out, 4096, bad, 256)
    if rc:
        raise ValueError(f"unknown option {bad.value.decode()}")
    return out.value.decode()


def open_output(args) -> str:
    n, arr = _argv(args)
    out = C.create_string_buffer(4096)
    rc = lib().hg_open_output(n, arr, out, 4096)
    if rc:
        raise RuntimeError(f"open_output failed ({rc})")
    return out.value.decode()


class Ratings:
    def __init__(self, cap_n, cap_m, binary=False, rating_threshold=1):
        self.L = lib()
        self._r = C.c_void_p(self.L.hg_ratings_new(cap_n, cap_m, int(binary), rating_threshold))

    def read_train(self, path):
        return self.L.hg_ratings_read_train(self._r, str(path).encode())

    def read_heldout(self, path, which):
        return self.L.hg_ratings_read_heldout(self._r, str(path).encode(), which)

    n = property(lambda s: s.L.hg_ratings_n(s._r))
    m = property(lambda s: s.L.hg_ratings_m(s._r))
    nnz = property(lambda s: s.L.hg_ratings_nnz(s._r))

    def csr(self):
        n, nnz = self.n, self.nnz
        rp = np.ctypeslib.as_array(self.L.hg_ratings_rowptr(self._r), shape=(n + 1,)).copy()
        if nnz == 0:
            return rp, np.zeros(0, np.uint32), np.zeros(0, np.uint8)
        col = np.ctypeslib.as_array(self.L.hg_ratings_col(self._r), shape=(nnz,)).copy()
        val = np.ctypeslib.as_array(self.L.hg_ratings_val(self._r), shape=(nnz,)).copy()
        return rp, col, val

    def seq2user(self):
        return np.ctypeslib.as_array(self.L.hg_ratings_seq2user(self._r), shape=(max(self.n, 1),))[: self.n].copy()

    def seq2item(self):
        return np.ctypeslib.as_array(self.L.hg_ratings_seq2item(self._r), shape=(max(self.m, 1),))[: self.m].copy()

    def heldout(self, which):
        c = self.L.hg_ratings_heldout_count(self._r, which)
        if c == 0:
            return np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.int32)
        return (np.ctypeslib.as_array(self.L.hg_ratings_heldout_u(self._r, which), shape=(c,)).copy(),
                np.ctypeslib.as_array(self.L.hg_ratings_heldout_i(self._r, which), shape=(c,)).copy(),
                np.ctypeslib.as_array(self.L.hg_ratings_heldout_y(self._r, which), shape=(c,)).copy())

    def write_marginals(self, byusers, byitems):
        return self.L.hg_ratings_write_marginals(self._r, str(byusers).encode(), str(byitems).encode())

    def save_cache(self, data_dir):
        """-cache: binary image of the parsed dataset in <data_dir>/hgaprec.cache.bin"""
        return self.L.hg_ratings_save_cache(self._r, str(data_dir).encode())

    def load_cache(self, data_dir):
        """0 = loaded; 1 = absent, stale (TSV size/mtime), other parameters or damaged"""
        return self.L.hg_ratings_load_cache(self._r, str(data_dir).encode())

    def test_users(self, path):
        out = np.empty(max(self.n, 1), np.uint32)
        c = self.L.hg_ratings_test_users(self._r, str(path).encode(), out.ctypes.data_as(C.POINTER(C.c_uint32)), out.size)
        return None if c < 0 else out[:c].copy()

    def __del__(self):
        try:
            self.L.hg_ratings_free(self._r)
        except Exception:
            pass


def mt_u32(seed, count):
    out = np.empty(count, np.uint32)
    lib().hg_mt_u32(float(seed), count, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def digamma(x):
    L = lib()
    return np.array([L.hg_digamma(float(v)) for v in np.atleast_1d(x)])


def initial_state(seed, n, m, k, hier, bias, rows=None) -> dict:
    """HGAPRec::initialize on the host -> {state name: array} (hpf_set_state layout).
    rows = (lo, hi): the user-side arrays of that range only, as one rank of several keeps them;
    the dict then also holds "_word_after", the generator's next 32-bit word."""
    from .capi import STATE_NAMES
    L = lib()
    out = {}
    if rows is None:
        s = C.c_void_p(L.hg_state_new(float(seed), n, m, k, int(hier), int(bias)))
    else:
        w = C.c_uint32(0)
        s = C.c_void_p(L.hg_state_new_range(float(seed), n, m, k, int(hier), int(bias), int(rows[0]), int(rows[1]), C.byref(w)))
        out["_word_after"] = int(w.value)
        n = int(rows[1]) - int(rows[0])
    try:
        for idx, name in enumerate(STATE_NAMES):
            p = C.POINTER(C.c_double)()
            cnt = L.hg_state_get(s, idx, C.byref(p))
            if cnt:
                a = np.ctypeslib.as_array(p, shape=(cnt,)).copy()
                obj, kind = idx // 4, idx % 4
                if obj <= 1 and not (kind == 1 and not hier):
                    a = a.reshape((n if obj == 0 else m), k)
                out[name] = a
    finally:
        L.hg_state_free(s)
    return out


def save_matrix(path, a, ids=None):
    a = np.ascontiguousarray(a, np.float64)
    i = None if ids is None else np.ascontiguousarray(ids, np.uint32)
    return lib().hg_save_matrix(str(path).encode(), a.ctypes.data_as(C.POINTER(C.c_double)), a.shape[0],
                                a.shape[1], None if i is None else i.ctypes.data_as(C.POINTER(C.c_uint32)),
                                0 if i is None else i.size)


def save_vector(path, a, ids=None):
    a = np.ascontiguousarray(a, np.float64)
    i = None if ids is None else np.ascontiguousarray(ids, np.uint32)
    return lib().hg_save_vector(str(path).encode(), a.ctypes.data_as(C.POINTER(C.c_double)), a.shape[0],
                                None if i is None else i.ctypes.data_as(C.POINTER(C.c_uint32)),
                                0 if i is None else i.size)


def format_fixed8(values):
    """the writers' "%.8f" (exact, printf-compatible) for an array -> list of str"""
    a = np.ascontiguousarray(values, np.float64)
    buf = C.create_string_buffer(a.size * 420 + 16)
    n = lib().hg_format_fixed8(a.ctypes.data_as(C.POINTER(C.c_double)), a.size, buf)
    return buf.raw[:n].decode().split("\n")[:-1]


def partition_users(rowptr, world):
    """the C++ host's user sharding (same rule as hgaprec_amd.dist.partition_users)"""
    rp = np.ascontiguousarray(rowptr, np.int64)
    out = np.empty(2 * world, np.uint32)
    lib().hg_partition_users(rp.ctypes.data_as(C.POINTER(C.c_int64)), rp.size - 1, world,
                             out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return [(int(out[2 * r]), int(out[2 * r + 1])) for r in range(world)]


def stop_rule(iters, series):
    it = np.ascontiguousarray(iters, np.uint32)
    a = np.ascontiguousarray(series, np.float64)
    why = (C.c_int * it.size)()
    at = lib().hg_stop_rule(it.ctypes.data_as(C.POINTER(C.c_uint32)), a.ctypes.data_as(C.POINTER(C.c_double)),
                            it.size, why)
    return at, list(why)
