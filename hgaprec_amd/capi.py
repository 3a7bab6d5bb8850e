"""ctypes binding of include/hpf.h (libhpf_hip.so).

This is the reference-side binding a Python caller would use; the C++ host
(`hgaprec` CLI, hgaprec_amd/csrc/host/) links the same library directly.
There is no CPU fallback: if the HIP library is missing or no gfx950 device is
visible, loading / `Hpf(...)` raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "libhpf_hip.so"

HPF_OK = 0
STATE_NAMES = [
    "THETA_SHAPE", "THETA_RATE", "THETA_E", "THETA_ELOG",
    "BETA_SHAPE", "BETA_RATE", "BETA_E", "BETA_ELOG",
    "XI_SHAPE", "XI_RATE", "XI_E", "XI_ELOG",
    "ETA_SHAPE", "ETA_RATE", "ETA_E", "ETA_ELOG",
    "UBIAS_SHAPE", "UBIAS_RATE", "UBIAS_E", "UBIAS_ELOG",
    "IBIAS_SHAPE", "IBIAS_RATE", "IBIAS_E", "IBIAS_ELOG",
]
STATE = {n: i for i, n in enumerate(STATE_NAMES)}

# every symbol include/hpf.h declares (tests check the library exports them all)
EXPORTS = [
    "hpf_abi_version", "hpf_strerror", "hpf_last_error", "hpf_create", "hpf_destroy",
    "hpf_upload_csr", "hpf_set_state", "hpf_get_state", "hpf_iterate",
    "hpf_iterate_local", "hpf_iterate_local_items", "hpf_iterate_local_users",
    "hpf_iterate_local_phi", "hpf_iterate_local_sweep", "hpf_exchange_buffer", "hpf_bind_exchange_buffer",
    "hpf_iterate_global", "hpf_heldout_ll", "hpf_synchronize", "hpf_gather_only", "hpf_last_timing",
    "hpf_mean_timing", "hpf_elbo", "hpf_scores", "hpf_rank_topn", "hpf_item_ranks",
    "hpf_comm_unique_id", "hpf_comm_init", "hpf_allreduce_items_begin", "hpf_allreduce_exchange", "hpf_exchange_read", "hpf_exchange_write",
    "hpf_algorithmic_bytes",
    "hpf_snapshot_size", "hpf_snapshot_save", "hpf_snapshot_load",
    "hpf_get_work_info", "hpf_upload_csr_device", "hpf_get_csc", "hpf_set_state_device", "hpf_get_state_device",
    "hpf_iteration_times", "hpf_debug_poke_index", "hpf_start_sums", "hpf_host_alloc", "hpf_host_free",
    "hpf_heldout_bind", "hpf_heldout_ll_bound",
]



class _PinnedBlock:
    """owner of one hpf_host_alloc block (freed with the last array that views it)"""
    def __init__(self, lib, nbytes):
        self.lib, self.ptr = lib, C.c_void_p()
        rc = lib.hpf_host_alloc(C.byref(self.ptr), nbytes)
        if rc != 0:
            raise HpfError(f"hpf_host_alloc({nbytes}): {lib.hpf_strerror(rc).decode()}")

    def __del__(self):
        if getattr(self, "ptr", None) and self.ptr.value:
            self.lib.hpf_host_free(self.ptr)
            self.ptr = C.c_void_p()


def pinned_empty(shape, dtype=np.float64) -> np.ndarray:
    """an uninitialised numpy array in page-locked host memory (include/hpf.h, hpf_host_alloc): hpf_get_state /
    hpf_set_state to and from it move at the rate of the DMA, with no staging copy"""
    lib = load_library()
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) if np.ndim(shape) else int(shape)
    blk = _PinnedBlock(lib, max(1, n * dt.itemsize))
    buf = (C.c_char * max(1, n * dt.itemsize)).from_address(blk.ptr.value)
    buf._hpf_block = blk                    # the ctypes buffer is the array's base: keeps the block alive
    return np.frombuffer(buf, dtype=dt, count=n).reshape(shape)

class HpfConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("n_users", C.c_uint32), ("n_items", C.c_uint32),
        ("K", C.c_uint32), ("hier", C.c_uint32), ("bias", C.c_uint32),
        ("binary", C.c_uint32), ("n_users_total", C.c_uint32), ("device", C.c_int32),
        ("n_ranks", C.c_uint32), ("rank", C.c_uint32), ("w_storage", C.c_uint32),
        ("stream", C.c_void_p), ("s_prior", C.c_double), ("r_prior", C.c_double),
        ("novb", C.c_uint32), ("tiling", C.c_uint32),
    ]


class HpfTiming(C.Structure):
    _fields_ = [
        ("phi_user_ms", C.c_float), ("combine_user_ms", C.c_float),
        ("phi_item_ms", C.c_float), ("combine_item_ms", C.c_float),
        ("sweep_user_ms", C.c_float), ("sweep_item_ms", C.c_float),
        ("iteration_ms", C.c_float), ("iterations", C.c_uint32),
        ("exchange_wait_ms", C.c_float),
    ]


class HpfWorkInfo(C.Structure):
    _fields_ = [
        ("nnz", C.c_uint64),
        ("user_segments", C.c_uint32), ("user_long_rows", C.c_uint32), ("user_huge_rows", C.c_uint32),
        ("item_segments", C.c_uint32), ("item_long_rows", C.c_uint32), ("item_huge_rows", C.c_uint32),
        ("phi_G", C.c_uint32), ("phi_R", C.c_uint32), ("phi_V", C.c_uint32),
        ("sweep_G", C.c_uint32), ("sweep_R", C.c_uint32), ("ld", C.c_uint32),
        ("graph_replay", C.c_uint32), ("w_layout", C.c_uint32), ("tiles_user", C.c_uint32), ("tiles_item", C.c_uint32),
        ("tile_rows_user", C.c_uint32), ("tile_rows_item", C.c_uint32),
        ("heavy_min_nnz_user", C.c_uint64), ("heavy_min_nnz_item", C.c_uint64),
        ("w_fallbacks", C.c_uint32), ("notes", C.c_uint32),
        ("start_sums_pending", C.c_uint32), ("tile_chunk_user", C.c_uint32), ("tile_chunk_item", C.c_uint32),
        ("reserved0", C.c_uint32),
    ]


class HpfError(RuntimeError):
    pass


_lib = None


def load_library(path: os.PathLike | None = None) -> C.CDLL:
    """dlopen libhpf_hip.so and declare the prototypes.  Fails loudly."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path) if path else LIB_PATH
    # A/B measurements only (tools/, profiles/): HPF_EXPERIMENTAL=1 HPF_LIB=<file name or path> loads another build of the
    # library -- e.g. last round's, kept beside this one -- in place of libhpf_hip.so
    if path is None and os.environ.get("HPF_EXPERIMENTAL") == "1" and os.environ.get("HPF_LIB"):
        alt = Path(os.environ["HPF_LIB"])
        p = alt if alt.is_absolute() else _PKG / alt
    # One HIP runtime per process.  libhpf_hip.so is linked against the system ROCm
    # (libamdhip64.so.7); a torch wheel brings its own copy of the HIP and HSA runtimes.
    # If the system copy is mapped first and torch is imported later, the process ends up
    # with TWO HSA runtimes and torch finds "No HIP GPUs"; the other way round the library's
    # dependency resolves to the copy torch already mapped (same SONAME).  So in a process
    # that may use torch at all, torch goes first.  (C/C++ callers are not concerned.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not p.exists():
        raise HpfError(
            f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(str(p))
    vp, u32p, dp = C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_double)
    lib.hpf_abi_version.restype = C.c_int
    lib.hpf_strerror.restype = C.c_char_p
    lib.hpf_strerror.argtypes = [C.c_int]
    lib.hpf_last_error.restype = C.c_char_p
    lib.hpf_last_error.argtypes = [vp]
    lib.hpf_create.argtypes = [C.POINTER(HpfConfig), C.POINTER(vp)]
    lib.hpf_destroy.argtypes = [vp]
    lib.hpf_destroy.restype = None
    lib.hpf_upload_csr.argtypes = [vp, C.POINTER(C.c_int64), u32p, C.POINTER(C.c_uint8)]
    lib.hpf_snapshot_size.argtypes = [vp, C.POINTER(C.c_size_t)]
    lib.hpf_snapshot_save.argtypes = [vp, vp, C.c_size_t]
    lib.hpf_snapshot_load.argtypes = [vp, vp, C.c_size_t]
    lib.hpf_get_work_info.argtypes = [vp, C.POINTER(HpfWorkInfo)]
    lib.hpf_upload_csr_device.argtypes = [vp, vp, vp, vp]
    lib.hpf_get_csc.argtypes = [vp, C.POINTER(C.c_int64), u32p, C.POINTER(C.c_uint8)]
    lib.hpf_set_state_device.argtypes = [vp, C.c_int, vp, C.c_size_t]
    lib.hpf_get_state_device.argtypes = [vp, C.c_int, vp, C.c_size_t]
    lib.hpf_set_state.argtypes = [vp, C.c_int, dp, C.c_size_t]
    lib.hpf_get_state.argtypes = [vp, C.c_int, dp, C.c_size_t]
    lib.hpf_iterate.argtypes = [vp, C.c_int]
    lib.hpf_iterate_local.argtypes = [vp]
    lib.hpf_iterate_global.argtypes = [vp]
    lib.hpf_start_sums.argtypes = [vp]
    lib.hpf_iterate_local_phi.argtypes = [vp]
    lib.hpf_iterate_local_items.argtypes = [vp]
    lib.hpf_iterate_local_users.argtypes = [vp]
    lib.hpf_iterate_local_sweep.argtypes = [vp]
    lib.hpf_exchange_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    lib.hpf_bind_exchange_buffer.argtypes = [vp, vp, C.c_size_t]
    lib.hpf_heldout_ll.argtypes = [vp, u32p, u32p, C.POINTER(C.c_int32), C.c_size_t, dp,
                                   C.POINTER(C.c_uint64)]
    if hasattr(lib, "hpf_heldout_bind"):          # (an older build loaded through HPF_LIB for an A/B run lacks the v8 calls)
        lib.hpf_heldout_bind.argtypes = [vp, C.c_int, u32p, u32p, C.POINTER(C.c_int32), C.c_size_t]
        lib.hpf_heldout_ll_bound.argtypes = [vp, C.c_int, dp, C.POINTER(C.c_uint64)]
    lib.hpf_elbo.argtypes = [vp, dp]
    u64p = C.POINTER(C.c_uint64)
    lib.hpf_scores.argtypes = [vp, u32p, C.c_uint32, dp]
    lib.hpf_rank_topn.argtypes = [vp, u32p, C.c_uint32, u64p, u32p, C.c_uint32, u32p, dp]
    lib.hpf_item_ranks.argtypes = [vp, u32p, C.c_uint32, u64p, u32p, u32p, u32p, C.c_uint32, u32p, dp]
    lib.hpf_comm_unique_id.argtypes = [vp]
    lib.hpf_comm_init.argtypes = [vp, vp]
    lib.hpf_allreduce_exchange.argtypes = [vp]
    lib.hpf_allreduce_items_begin.argtypes = [vp]
    lib.hpf_exchange_read.argtypes = [vp, dp, C.c_size_t]
    lib.hpf_exchange_write.argtypes = [vp, dp, C.c_size_t]
    lib.hpf_synchronize.argtypes = [vp]
    lib.hpf_host_alloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    lib.hpf_host_free.argtypes = [C.c_void_p]
    lib.hpf_gather_only.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_float)]
    lib.hpf_last_timing.argtypes = [vp, C.POINTER(HpfTiming)]
    lib.hpf_mean_timing.argtypes = [vp, C.c_uint32, C.POINTER(HpfTiming)]
    lib.hpf_algorithmic_bytes.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                          C.POINTER(C.c_uint64)]
    lib.hpf_iteration_times.argtypes = [vp, C.c_uint32, C.POINTER(C.c_float), u32p]
    lib.hpf_debug_poke_index.argtypes = [vp, C.c_int, C.c_uint64, C.c_uint32, u32p, u32p]
    if path is None:
        _lib = lib
    return lib


def _ptr(a: np.ndarray, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


class Hpf:
    """One device-side model (one rank's shard).  Thin, 1:1 over the C-ABI."""

    def __init__(self, n_users, n_items, K, hier=True, bias=False, binary=False,
                 device=0, stream=None, n_ranks=1, rank=0, n_users_total=0,
                 s_prior=0.3, r_prior=0.3, w_storage=0, novb=False, tiling=0):
        self.lib = load_library()
        cfg = HpfConfig()
        cfg.struct_size = C.sizeof(HpfConfig)
        cfg.n_users, cfg.n_items, cfg.K = int(n_users), int(n_items), int(K)
        cfg.hier, cfg.bias, cfg.binary = int(bool(hier)), int(bool(bias)), int(bool(binary))
        cfg.n_users_total = int(n_users_total)
        cfg.device, cfg.n_ranks, cfg.rank = int(device), int(n_ranks), int(rank)
        cfg.stream = C.c_void_p(stream) if stream else None
        cfg.s_prior, cfg.r_prior = float(s_prior), float(r_prior)
        cfg.w_storage = int(w_storage)
        cfg.novb = int(bool(novb))
        cfg.tiling = int(tiling)             # 0: the library decides (tiled phi pass where it pays), 1: never
        self.n_users, self.n_items, self.K = int(n_users), int(n_items), int(K)
        self.hier, self.bias, self.binary = bool(hier), bool(bias), bool(binary)
        self.n_ranks = int(n_ranks)
        self.device = int(device)            # the HIP ordinal every device pointer handed in must live on
        self._h = C.c_void_p()
        rc = self.lib.hpf_create(C.byref(cfg), C.byref(self._h))
        if rc != HPF_OK:
            self._h = C.c_void_p()
            raise HpfError(f"hpf_create failed: {self.lib.hpf_strerror(rc).decode()} ({rc})")

    # -- plumbing
    def _check(self, rc):
        if rc != HPF_OK:
            msg = self.lib.hpf_last_error(self._h).decode()
            raise HpfError(f"{self.lib.hpf_strerror(rc).decode()} ({rc}): {msg}")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.hpf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _on_my_device(self, t, what):
        """a raw data_ptr() of another GPU's memory would be a cross-device access (or a
        fault) inside the library instead of an error here"""
        if not t.is_cuda or t.device.index != self.device:
            raise ValueError(f"{what}: tensor on {t.device}, this handle runs on cuda:{self.device}")

    # -- data
    def upload_csr(self, rowptr, col, val=None):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        col = np.ascontiguousarray(col, dtype=np.uint32)
        if rowptr.shape[0] != self.n_users + 1:
            raise ValueError("rowptr must have n_users + 1 entries")
        vp = None
        if val is not None:
            val = np.ascontiguousarray(val, dtype=np.uint8)
            vp = _ptr(val, C.c_uint8)
        self._check(self.lib.hpf_upload_csr(self._h, _ptr(rowptr, C.c_int64),
                                            _ptr(col, C.c_uint32), vp))

    def upload_csr_device(self, rowptr, col, val=None):
        """CSR already resident in HBM: torch tensors on this handle's device
        (int64 rowptr[n+1], int32/uint32-as-int32 col[nnz], uint8 val[nnz] or None).
        Torch is plumbing here: only data_ptr() crosses the C-ABI."""
        import torch
        if rowptr.dtype != torch.int64 or rowptr.numel() != self.n_users + 1 or not rowptr.is_cuda:
            raise ValueError("rowptr: int64 device tensor with n_users + 1 entries")
        u32 = getattr(torch, "uint32", None)              # absent before torch 2.3
        if col.dtype not in tuple(d for d in (torch.int32, u32) if d is not None) or not col.is_cuda:
            raise ValueError("col: 32-bit device tensor")
        if val is not None and (val.dtype != torch.uint8 or not val.is_cuda or val.numel() != col.numel()):
            raise ValueError("val: uint8 device tensor as long as col")
        for t, what in ((rowptr, "rowptr"), (col, "col"), (val, "val")):
            if t is not None:
                self._on_my_device(t, what)
        rowptr, col = rowptr.contiguous(), col.contiguous()
        val = None if val is None else val.contiguous()
        torch.cuda.synchronize(rowptr.device)             # the producer's work is complete
        if int(rowptr[-1]) != col.numel() or int(rowptr[0]) != 0:     # the library reads rowptr[n] entries of col
            raise ValueError(f"rowptr runs from {int(rowptr[0])} to {int(rowptr[-1])} but col has {col.numel()} entries")
        self._check(self.lib.hpf_upload_csr_device(
            self._h, C.c_void_p(rowptr.data_ptr()), C.c_void_p(col.data_ptr() if col.numel() else None),
            C.c_void_p(val.data_ptr()) if val is not None and val.numel() else None))

    def get_csc(self, nnz: int | None = None, with_vals=True):
        """host copies of the item-major view: colptr[m+1], users[nnz], vals[nnz] | None"""
        have = self.work_info()["nnz"]
        if nnz is not None and int(nnz) != have:
            raise ValueError(f"the uploaded matrix has {have} nonzeros, not {nnz}")
        nnz = have
        colptr = np.empty(self.n_items + 1, np.int64)
        users = np.empty(nnz, np.uint32)
        vals = np.empty(nnz, np.uint8) if with_vals else None
        self._check(self.lib.hpf_get_csc(self._h, _ptr(colptr, C.c_int64), _ptr(users, C.c_uint32),
                                         _ptr(vals, C.c_uint8) if with_vals else None))
        return colptr, users, vals

    def set_state_device(self, which: str, t):
        """like set_state for a float64 torch tensor on this handle's device"""
        import torch
        if t.dtype != torch.float64 or not t.is_cuda or tuple(t.shape) != self.state_shape(which):
            raise ValueError(f"{which}: float64 device tensor of shape {self.state_shape(which)}")
        self._on_my_device(t, which)
        t = t.contiguous()
        torch.cuda.synchronize(t.device)
        self._check(self.lib.hpf_set_state_device(self._h, STATE[which], C.c_void_p(t.data_ptr()), t.numel()))

    def get_state_device(self, which: str, device=None):
        import torch
        dev = torch.device("cuda", self.device) if device is None else torch.device(device)
        if dev.type != "cuda" or (dev.index is not None and dev.index != self.device):
            raise ValueError(f"{which}: asked for {dev}, this handle runs on cuda:{self.device}")
        out = torch.empty(self.state_shape(which), dtype=torch.float64, device=torch.device("cuda", self.device))
        torch.cuda.synchronize(out.device)       # the block may still be in use by work queued on torch's stream
        self._check(self.lib.hpf_get_state_device(self._h, STATE[which], C.c_void_p(out.data_ptr()), out.numel()))
        return out

    def state_shape(self, which: str):
        obj = STATE[which] // 4
        kind = STATE[which] % 4
        if obj == 0:
            shp = (self.n_users, self.K)
        elif obj == 1:
            shp = (self.n_items, self.K)
        elif obj in (2, 4):
            shp = (self.n_users,)
        else:
            shp = (self.n_items,)
        if obj <= 1 and kind == 1 and not self.hier:
            shp = (self.K,)
        return shp

    def set_state(self, which: str, arr):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        if a.shape != self.state_shape(which):
            raise ValueError(f"{which}: expected shape {self.state_shape(which)}, got {a.shape}")
        self._check(self.lib.hpf_set_state(self._h, STATE[which], _ptr(a, C.c_double), a.size))

    def get_state(self, which: str, out: np.ndarray | None = None) -> np.ndarray:
        """out: a C-contiguous float64 array of the state's shape to fill (e.g. from pinned_empty: the DMA then writes
        it directly) instead of a fresh one"""
        if out is None:
            out = np.empty(self.state_shape(which), dtype=np.float64)
        elif out.dtype != np.float64 or not out.flags.c_contiguous or out.shape != self.state_shape(which):
            raise ValueError(f"{which}: out must be a C-contiguous float64 array of shape {self.state_shape(which)}")
        self._check(self.lib.hpf_get_state(self._h, STATE[which], _ptr(out, C.c_double), out.size))
        return out

    # -- compute
    def iterate(self, n_iters=1):
        self._check(self.lib.hpf_iterate(self._h, int(n_iters)))

    def iterate_local(self):
        self._check(self.lib.hpf_iterate_local(self._h))

    def iterate_local_items(self):
        self._check(self.lib.hpf_iterate_local_items(self._h))

    def iterate_local_users(self):
        self._check(self.lib.hpf_iterate_local_users(self._h))

    def iterate_local_phi(self):
        self._check(self.lib.hpf_iterate_local_phi(self._h))

    def iterate_local_sweep(self):
        self._check(self.lib.hpf_iterate_local_sweep(self._h))

    def iterate_global(self):
        self._check(self.lib.hpf_iterate_global(self._h))

    def start_sums(self):
        """-novb on several ranks: leave this rank's sum_u E[theta] of the start state in the tail of the
        exchange buffer (the caller all-reduces its last ld doubles unless hpf_comm_init was done)"""
        self._check(self.lib.hpf_start_sums(self._h))

    def exchange_buffer(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self.lib.hpf_exchange_buffer(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def bind_exchange_buffer(self, dev_ptr: int, count: int):
        """dev_ptr: device memory the caller owns (e.g. a torch tensor's data_ptr()).  The
        library clears it on ITS stream, so no other stream may still be using that memory:
        a torch caching-allocator block can be one that kernels queued earlier on torch's
        stream are still working in (torch only orders reuse on the same stream) -- found the
        hard way in tools/emulate_shards.py, where the clear landed in the temporaries of a
        generator that was still running.  If torch is loaded, quiesce it first."""
        import sys
        t = sys.modules.get("torch")
        if t is not None and t.cuda.is_initialized():
            t.cuda.synchronize()
        self._check(self.lib.hpf_bind_exchange_buffer(self._h, C.c_void_p(dev_ptr), count))

    def exchange_read(self) -> np.ndarray:
        out = np.empty(self.exchange_count(), dtype=np.float64)
        self._check(self.lib.hpf_exchange_read(self._h, _ptr(out, C.c_double), out.size))
        return out

    def exchange_write(self, arr):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        self._check(self.lib.hpf_exchange_write(self._h, _ptr(a, C.c_double), a.size))

    @staticmethod
    def _rccl_of_this_process():
        """The library dlopen()s librccl.so.1 on first use.  In a Python process that
        also uses torch, torch's own copy (same SONAME, possibly another version) must
        be the one that is mapped: whichever loads first is what BOTH get.  Importing
        torch first makes that torch's -- one RCCL per process, deterministically."""
        try:
            import torch  # noqa: F401
        except ImportError:
            pass

    @staticmethod
    def comm_unique_id() -> bytes:
        Hpf._rccl_of_this_process()
        buf = C.create_string_buffer(128)
        rc = load_library().hpf_comm_unique_id(buf)
        if rc != HPF_OK:
            raise HpfError(f"hpf_comm_unique_id failed ({rc}): is librccl.so available?")
        return buf.raw

    def comm_init(self, unique_id: bytes):
        self._rccl_of_this_process()
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._check(self.lib.hpf_comm_init(self._h, buf))

    def allreduce_items_begin(self):
        self._check(self.lib.hpf_allreduce_items_begin(self._h))

    def allreduce_exchange(self):
        self._check(self.lib.hpf_allreduce_exchange(self._h))

    def exchange_count(self):
        return self.exchange_buffer()[1]

    def heldout_ll(self, u, i, y):
        u = np.ascontiguousarray(u, dtype=np.uint32)
        i = np.ascontiguousarray(i, dtype=np.uint32)
        y = np.ascontiguousarray(y, dtype=np.int32)
        s, c = C.c_double(), C.c_uint64()
        self._check(self.lib.hpf_heldout_ll(self._h, _ptr(u, C.c_uint32), _ptr(i, C.c_uint32),
                                            _ptr(y, C.c_int32), u.size, C.byref(s), C.byref(c)))
        return s.value, c.value

    def heldout_bind(self, slot, u, i, y) -> None:
        """validate and upload a held-out set once (hpf_heldout_bind); heldout_ll_bound(slot) then evaluates it"""
        u = np.ascontiguousarray(u, dtype=np.uint32)
        i = np.ascontiguousarray(i, dtype=np.uint32)
        y = np.ascontiguousarray(y, dtype=np.int32)
        if not (u.size == i.size == y.size):
            raise ValueError("u, i, y differ in length")
        self._check(self.lib.hpf_heldout_bind(self._h, int(slot), _ptr(u, C.c_uint32), _ptr(i, C.c_uint32),
                                              _ptr(y, C.c_int32), u.size))

    def heldout_ll_bound(self, slot):
        s, c = C.c_double(), C.c_uint64()
        self._check(self.lib.hpf_heldout_ll_bound(self._h, int(slot), C.byref(s), C.byref(c)))
        return s.value, c.value

    def elbo(self) -> float:
        v = C.c_double()
        self._check(self.lib.hpf_elbo(self._h, C.byref(v)))
        return v.value

    def scores(self, users) -> np.ndarray:
        users = np.ascontiguousarray(users, dtype=np.uint32)
        out = np.empty((users.size, self.n_items), dtype=np.float64)
        self._check(self.lib.hpf_scores(self._h, _ptr(users, C.c_uint32), users.size, _ptr(out, C.c_double)))
        return out

    @staticmethod
    def _mask(mask_ptr, mask_items):
        if mask_ptr is None:
            return None, None, None, None
        mp = np.ascontiguousarray(mask_ptr, dtype=np.uint64)
        mi = np.ascontiguousarray(mask_items, dtype=np.uint32)
        return mp, mi, _ptr(mp, C.c_uint64), _ptr(mi, C.c_uint32)

    def rank_topn(self, users, topn=100, mask_ptr=None, mask_items=None):
        users = np.ascontiguousarray(users, dtype=np.uint32)
        mp, mi, pmp, pmi = self._mask(mask_ptr, mask_items)
        items = np.empty((users.size, topn), dtype=np.uint32)
        sc = np.empty((users.size, topn), dtype=np.float64)
        self._check(self.lib.hpf_rank_topn(self._h, _ptr(users, C.c_uint32), users.size, pmp, pmi, topn,
                                           _ptr(items, C.c_uint32), _ptr(sc, C.c_double)))
        return items, sc

    def item_ranks(self, users, q_sel, q_item, mask_ptr=None, mask_items=None):
        users = np.ascontiguousarray(users, dtype=np.uint32)
        q_sel = np.ascontiguousarray(q_sel, dtype=np.uint32)
        q_item = np.ascontiguousarray(q_item, dtype=np.uint32)
        mp, mi, pmp, pmi = self._mask(mask_ptr, mask_items)
        rank = np.empty(q_sel.size, dtype=np.uint32)
        sc = np.empty(q_sel.size, dtype=np.float64)
        self._check(self.lib.hpf_item_ranks(self._h, _ptr(users, C.c_uint32), users.size, pmp, pmi,
                                            _ptr(q_sel, C.c_uint32), _ptr(q_item, C.c_uint32), q_sel.size,
                                            _ptr(rank, C.c_uint32), _ptr(sc, C.c_double)))
        return rank, sc

    def synchronize(self):
        self._check(self.lib.hpf_synchronize(self._h))

    def last_timing(self) -> dict:
        t = HpfTiming()
        self._check(self.lib.hpf_last_timing(self._h, C.byref(t)))
        return {f: getattr(t, f) for f, _ in HpfTiming._fields_}

    def mean_timing(self, n_last: int) -> dict:
        t = HpfTiming()
        self._check(self.lib.hpf_mean_timing(self._h, int(n_last), C.byref(t)))
        return {f: getattr(t, f) for f, _ in HpfTiming._fields_}

    def iteration_times(self, n_last: int) -> np.ndarray:
        """iteration_ms of each of the last n_last iterations (hipEvents), oldest first"""
        out = np.zeros(max(int(n_last), 1), np.float32)
        got = C.c_uint32(0)
        self._check(self.lib.hpf_iteration_times(self._h, int(n_last), _ptr(out, C.c_float), C.byref(got)))
        return out[: got.value].astype(np.float64)

    def debug_poke_index(self, side: int, pos: int, value: int):
        """TEST HOOK: overwrite one entry of the index stream of a phi pass -> (old value, owner row)"""
        old, own = C.c_uint32(0), C.c_uint32(0)
        self._check(self.lib.hpf_debug_poke_index(self._h, int(side), int(pos), int(value), C.byref(old), C.byref(own)))
        return old.value, own.value

    def snapshot(self) -> np.ndarray:
        """the loop's device state as one opaque blob (uint8 array)"""
        n = C.c_size_t()
        self._check(self.lib.hpf_snapshot_size(self._h, C.byref(n)))
        buf = np.empty(n.value, np.uint8)
        self._check(self.lib.hpf_snapshot_save(self._h, C.c_void_p(buf.ctypes.data), n.value))
        return buf

    def restore(self, blob: np.ndarray):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        self._check(self.lib.hpf_snapshot_load(self._h, C.c_void_p(blob.ctypes.data), blob.size))

    def gather_only_ms(self, side: int, reps: int = 3) -> float:
        """mean time of a phi pass with the arithmetic taken out (0: user-major, 1: item-major)"""
        ms = C.c_float(0.0)
        self._check(self.lib.hpf_gather_only(self._h, int(side), int(reps), C.byref(ms)))
        return float(ms.value)

    def work_info(self) -> dict:
        w = HpfWorkInfo()
        self._check(self.lib.hpf_get_work_info(self._h, C.byref(w)))
        return {f: getattr(w, f) for f, _ in HpfWorkInfo._fields_}

    def algorithmic_bytes(self) -> dict:
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self.lib.hpf_algorithmic_bytes(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"phi_user": a.value, "phi_item": b.value, "rows": c.value}
