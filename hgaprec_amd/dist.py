"""Multi-GPU plumbing: user sharding and the one all-reduce per iteration.

The path shards by users: each rank owns a contiguous range of
user seq ids (boundaries chosen on the nnz prefix sum, not on user counts),
their nonzeros and their theta / xi / user-bias state; beta / eta / item-bias
state is replicated.  Per iteration every rank runs hpf_iterate_local, the
exchange buffer [m x ld item shape sums | ld column sums of E[theta]] is
sum-all-reduced (RCCL over xGMI through torch.distributed, backend "nccl";
"gloo" on CPU in the tests), then every rank runs the identical
hpf_iterate_global.  One process per GPU.
"""
from __future__ import annotations

import numpy as np


def partition_users(rowptr: np.ndarray, world: int):
    """-> [(a, b)] * world: contiguous user ranges with ~nnz/world nonzeros each.
    Every range is non-empty when n >= world."""
    n = rowptr.shape[0] - 1
    nnz = int(rowptr[-1])
    cuts = [0]
    for r in range(1, world):
        target = nnz * r // world
        c = int(np.searchsorted(rowptr, target, side="left"))
        c = max(c, cuts[-1] + 1)                 # at least one user per rank
        c = min(c, n - (world - r))              # leave one for each later rank
        cuts.append(max(c, cuts[-1]))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def shard_csr(rowptr, col, val, a: int, b: int):
    """CSR of users [a, b) with local row indices"""
    lo, hi = int(rowptr[a]), int(rowptr[b])
    rp = (rowptr[a:b + 1] - rowptr[a]).astype(np.int64)
    return rp, col[lo:hi], (None if val is None else val[lo:hi])


def shard_heldout(u, i, y, a: int, b: int):
    """held-out pairs whose user falls in [a, b), with local user indices"""
    sel = (u >= a) & (u < b)
    return (u[sel] - a).astype(np.uint32), i[sel], y[sel]


USER_STATES = ("THETA_SHAPE", "THETA_RATE", "THETA_E", "THETA_ELOG", "XI_SHAPE", "XI_RATE", "XI_E",
               "XI_ELOG", "UBIAS_SHAPE", "UBIAS_E", "UBIAS_ELOG")
ITEM_STATES = ("BETA_SHAPE", "BETA_RATE", "BETA_E", "BETA_ELOG", "ETA_SHAPE", "ETA_RATE", "ETA_E",
               "ETA_ELOG", "IBIAS_SHAPE", "IBIAS_E", "IBIAS_ELOG")


def scatter_state(engine, state: dict, a: int, b: int, hier: bool):
    """hand a full start state (hostlib.initial_state layout) to one shard"""
    for w in USER_STATES:
        if w in state:
            if w == "THETA_RATE" and not hier:
                engine.set_state(w, state[w])            # K-vector, shared
            else:
                engine.set_state(w, state[w][a:b])
    for w in ITEM_STATES:
        if w in state:
            engine.set_state(w, state[w])


class Exchange:
    """Owns the tensor the engine uses as its exchange buffer and reduces it.
    `engine` needs exchange_count() and bind_exchange_buffer(ptr, count)
    (hgaprec_amd.capi.Hpf) -- or, for host-side engines, set_exchange_array."""

    def __init__(self, engine, device=None, group=None):
        import torch
        self.torch = torch
        self.group = group
        cnt = engine.exchange_count()
        self.buf = torch.zeros(cnt, dtype=torch.float64, device=device or "cpu")
        if self.buf.is_cuda:
            engine.bind_exchange_buffer(self.buf.data_ptr(), cnt)
        else:
            engine.set_exchange_array(self.buf.numpy())

    def allreduce(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=self.group)

    def allreduce_tail(self, ld: int):
        """the last ld doubles alone: sum_u E[theta_u,:] (hpf_start_sums)"""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(self.buf[-ld:], op=dist.ReduceOp.SUM, group=self.group)


def start_sums(engine, exchange: Exchange | None):
    """-bias -novb without -hier on several ranks (vb_bias()'s else-branch, hgaprec.cc:1276-1297): the first
    item rate is built from sum_u E[theta] of the START state, summed over the ranks once.  The engine leaves
    its part in the tail of the exchange buffer (hpf_start_sums); the tail alone is reduced.  Call after the
    start state has been handed over and before the first iterate(); a no-op for every other mode."""
    wi = engine.work_info()
    pending = bool(wi.get("start_sums_pending", 1))     # 0 after a snapshot load: the tail came with it, already reduced
    engine.start_sums()
    if exchange is not None and pending:
        exchange.allreduce_tail(wi["ld"])


def iterate(engine, exchange: Exchange | None, n_iters: int = 1):
    """n_iters CAVI iterations of one rank of a sharded run (bench.py overlaps the
    big all-reduce with the user-major pass and the user sweep through
    iterate_local_items / iterate_local_users)"""
    for _ in range(n_iters):
        engine.iterate_local()
        if exchange is not None:
            exchange.allreduce()
        engine.iterate_global()


def global_heldout_mean(engine, u, i, y, device=None, group=None):
    """mean held-out log-likelihood over all ranks' local pairs"""
    import torch
    import torch.distributed as dist
    s, c = engine.heldout_ll(u, i, y)
    t = torch.tensor([s, float(c)], dtype=torch.float64, device=device or "cpu")
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, group=group)
    return float(t[0] / t[1]), int(t[1])
