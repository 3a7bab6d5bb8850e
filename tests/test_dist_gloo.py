"""CPU, world_size 2, gloo: the N>1 protocol of hgaprec_amd/dist.py -- user
partition by nnz, exchange-buffer layout [m x ld item sums | ld colsums],
one sum-all-reduce between iterate_local and iterate_global, prior added once
after the reduce -- must reproduce the single-process oracle.

The engine here is a small numpy test double with the C-ABI's call sequence
(the HIP engine needs a GPU; tests/test_gpu_parity.py::test_two_logical_ranks
runs the same protocol on the device).  The double is itself checked against
the oracle at world_size 1."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from scipy.special import digamma

from hgaprec_amd import dist as hd
from tests.util import make_problem, rel_err


class NumpyEngine:
    """-hier (+ optional -bias) CAVI sweep in numpy with the hpf_* call sequence"""

    def __init__(self, n, m, K, bias, n_total, rowptr, col, val):
        self.n, self.m, self.K, self.bias, self.n_total = n, m, K, bias, n_total
        self.C = K + (2 if bias else 0)
        self.ld = (self.C + 1) & ~1
        self.rowptr, self.col, self.val = rowptr, col, val
        self.st = {}
        self.x = np.zeros(m * self.ld + self.ld)

    def exchange_count(self):
        return self.x.size

    def set_exchange_array(self, arr):
        self.x = arr

    def set_state(self, w, a):
        self.st[w] = np.array(a, np.float64)

    def get_state(self, w):
        return self.st[w]

    def _elog(self, side, r):
        K = self.K
        out = np.zeros(self.C)
        if side == "u":
            out[:K] = self.st["THETA_ELOG"][r]
            if self.bias:
                out[K] = self.st["UBIAS_ELOG"][r]
        else:
            out[:K] = self.st["BETA_ELOG"][r]
            if self.bias:
                out[K + 1] = self.st["IBIAS_ELOG"][r]
        return out

    def iterate_local(self):
        K, C, ld, m = self.K, self.C, self.ld, self.m
        St = np.zeros((self.n, C))
        Sb = np.zeros((m, ld))
        for u in range(self.n):
            lu = self._elog("u", u)
            for j in range(self.rowptr[u], self.rowptr[u + 1]):
                i, y = int(self.col[j]), int(self.val[j])
                x = lu + self._elog("i", i)
                e = np.exp(x - x.max())
                phi = e / e.sum() * (y if y > 1 else 1)
                St[u] += phi
                Sb[i, :C] += phi
        # user sweep (steps B, D-user, E)
        c = self.st["BETA_E"].sum(0)
        shape = 0.3 + St[:, :K]
        rate = self.st["XI_E"][:, None] + c[None, :]
        self.st["THETA_SHAPE"], self.st["THETA_RATE"] = shape, rate
        self.st["THETA_E"] = shape / rate
        self.st["THETA_ELOG"] = digamma(shape) - np.log(rate)
        if self.bias:
            sh, rt = 0.3 + St[:, K], 0.3 + m
            self.st["UBIAS_SHAPE"], self.st["UBIAS_E"] = sh, sh / rt
            self.st["UBIAS_ELOG"] = digamma(sh) - np.log(rt)
        rx = 0.3 + self.st["THETA_E"].sum(1)
        self.st["XI_RATE"], self.st["XI_E"] = rx, (0.3 + K * 0.3) / rx
        # exchange payload: item sums WITHOUT the prior, then sum_u E[theta]
        self.x[: m * ld] = Sb.ravel()
        self.x[m * ld:] = 0.0
        self.x[m * ld: m * ld + K] = self.st["THETA_E"].sum(0)

    def iterate_global(self):
        K, ld, m = self.K, self.ld, self.m
        Sb = self.x[: m * ld].reshape(m, ld)
        d = self.x[m * ld: m * ld + K]
        shape = 0.3 + Sb[:, :K]
        rate = self.st["ETA_E"][:, None] + d[None, :]
        self.st["BETA_SHAPE"], self.st["BETA_RATE"] = shape, rate
        self.st["BETA_E"] = shape / rate
        self.st["BETA_ELOG"] = digamma(shape) - np.log(rate)
        if self.bias:
            sh, rt = 0.3 + Sb[:, K + 1], 0.3 + self.n_total
            self.st["IBIAS_SHAPE"], self.st["IBIAS_E"] = sh, sh / rt
            self.st["IBIAS_ELOG"] = digamma(sh) - np.log(rt)
        re = 0.3 + self.st["BETA_E"].sum(1)
        self.st["ETA_RATE"], self.st["ETA_E"] = re, (0.3 + K * 0.3) / re


class NumpyEngineFlatNovb:
    """vb_bias() with -novb (no -hier; hgaprec.cc:1276-1297) in numpy with the hpf_* call sequence: K-vector rates,
    BOTH built from the previous iteration's expectations -- the item rate from sum_u E[theta] of before the user
    update, which on several ranks is what hpf_start_sums + one all-reduce of the tail provide for the first iteration"""

    def __init__(self, n, m, K, n_total, rowptr, col, val):
        self.n, self.m, self.K, self.n_total = n, m, K, n_total
        self.C = K + 2
        self.ld = self.C
        self.rowptr, self.col, self.val = rowptr, col, val
        self.st = {}
        self.x = np.zeros(m * self.ld + self.ld)

    def exchange_count(self):
        return self.x.size

    def set_exchange_array(self, arr):
        self.x = arr

    def work_info(self):
        return {"ld": self.ld}

    def set_state(self, w, a):
        self.st[w] = np.array(a, np.float64)

    def get_state(self, w):
        return self.st[w]

    def start_sums(self):
        self.x[self.m * self.ld:] = 0.0
        self.x[self.m * self.ld: self.m * self.ld + self.K] = self.st["THETA_E"].sum(0)

    def iterate_local(self):
        K, C, ld, m = self.K, self.C, self.ld, self.m
        self.prev_tail = self.x[m * ld: m * ld + K].copy()          # the REDUCED sum_u E[theta] of the previous iteration
        St = np.zeros((self.n, C))
        Sb = np.zeros((m, ld))
        for u in range(self.n):
            lu = np.concatenate([self.st["THETA_ELOG"][u], [self.st["UBIAS_ELOG"][u], 0.0]])
            for j in range(self.rowptr[u], self.rowptr[u + 1]):
                i, y = int(self.col[j]), int(self.val[j])
                li = np.concatenate([self.st["BETA_ELOG"][i], [0.0, self.st["IBIAS_ELOG"][i]]])
                x = lu + li
                e = np.exp(x - x.max())
                phi = e / e.sum() * (y if y > 1 else 1)
                St[u] += phi
                Sb[i, :C] += phi
        shape, rate = 0.3 + St[:, :K], 0.3 + self.st["BETA_E"].sum(0)
        self.st["THETA_SHAPE"], self.st["THETA_RATE"] = shape, rate
        self.st["THETA_E"] = shape / rate[None, :]
        self.st["THETA_ELOG"] = digamma(shape) - np.log(rate)[None, :]
        sh, rt = 0.3 + St[:, K], 0.3 + m
        self.st["UBIAS_SHAPE"], self.st["UBIAS_E"], self.st["UBIAS_ELOG"] = sh, sh / rt, digamma(sh) - np.log(rt)
        self.x[: m * ld] = Sb.ravel()
        self.x[m * ld:] = 0.0
        self.x[m * ld: m * ld + K] = self.st["THETA_E"].sum(0)

    def iterate_global(self):
        K, ld, m = self.K, self.ld, self.m
        Sb = self.x[: m * ld].reshape(m, ld)
        shape, rate = 0.3 + Sb[:, :K], 0.3 + self.prev_tail          # -novb: the OLD sum_u E[theta]
        self.st["BETA_SHAPE"], self.st["BETA_RATE"] = shape, rate
        self.st["BETA_E"] = shape / rate[None, :]
        self.st["BETA_ELOG"] = digamma(shape) - np.log(rate)[None, :]
        sh, rt = 0.3 + Sb[:, K + 1], 0.3 + self.n_total
        self.st["IBIAS_SHAPE"], self.st["IBIAS_E"], self.st["IBIAS_ELOG"] = sh, sh / rt, digamma(sh) - np.log(rt)


N, M, K, NNZ, SEED, ITERS = 120, 80, 6, 1500, 13, 3


def _run_rank_novb(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import orc
    rowptr, col, val = make_problem(N, M, NNZ, SEED)
    Mo = orc.Model(N, M, K, False, True, False, novb=True)
    Mo.set_csr(rowptr, col, val)
    Mo.initialize(5)
    init = {w: Mo.state(w) for w in ("THETA_SHAPE", "THETA_E", "THETA_ELOG", "BETA_SHAPE", "BETA_E", "BETA_ELOG",
                                      "UBIAS_SHAPE", "UBIAS_E", "UBIAS_ELOG", "IBIAS_SHAPE", "IBIAS_E", "IBIAS_ELOG")}
    Mo.iterate(ITERS)
    a, b = hd.partition_users(rowptr, world)[rank]
    rp, c, v = hd.shard_csr(rowptr, col, val, a, b)
    eng = NumpyEngineFlatNovb(b - a, M, K, N, rp, c, v)
    hd.scatter_state(eng, init, a, b, hier=False)
    ex = hd.Exchange(eng)
    hd.start_sums(eng, ex)                         # the start state's sum_u E[theta], over all ranks, before the first iteration
    hd.iterate(eng, ex, ITERS)
    errs = {"THETA_E": rel_err(eng.get_state("THETA_E"), Mo.state("THETA_E")[a:b]),
            "BETA_E": rel_err(eng.get_state("BETA_E"), Mo.state("BETA_E")),
            "BETA_RATE": rel_err(eng.get_state("BETA_RATE"), Mo.state("BETA_RATE")),
            "UBIAS_E": rel_err(eng.get_state("UBIAS_E"), Mo.state("UBIAS_E")[a:b]),
            "IBIAS_E": rel_err(eng.get_state("IBIAS_E"), Mo.state("IBIAS_E"))}
    q.put((rank, errs, (a, b)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2])
def test_novb_start_sums_protocol_over_gloo(world):
    """dist.start_sums: -bias -novb across ranks needs the START state's sum_u E[theta] reduced once (round 4)"""
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    if world == 1:
        _run_rank_novb(0, 1, port, q)
        res = [q.get()]
    else:
        procs = [ctx.Process(target=_run_rank_novb, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = [q.get() for _ in procs]
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
    for _, errs, _ in res:
        assert max(errs.values()) < 1e-10, errs


def _oracle_states(bias):
    from oracle import orc
    rowptr, col, val = make_problem(N, M, NNZ, SEED)
    Mo = orc.Model(N, M, K, True, bias, False)
    Mo.set_csr(rowptr, col, val)
    Mo.initialize(5)
    names = list(hd.USER_STATES) + list(hd.ITEM_STATES)
    init = {}
    for w in names:
        try:
            init[w] = Mo.state(w)
        except KeyError:
            pass
    Mo.iterate(ITERS)
    final = {w: Mo.state(w) for w in ("THETA_E", "BETA_E", "XI_E", "ETA_E")}
    if bias:
        final["UBIAS_E"], final["IBIAS_E"] = Mo.state("UBIAS_E"), Mo.state("IBIAS_E")
    return (rowptr, col, val), init, final


def _run_rank(rank, world, bias, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    (rowptr, col, val), init, final = _oracle_states(bias)
    a, b = hd.partition_users(rowptr, world)[rank]
    rp, c, v = hd.shard_csr(rowptr, col, val, a, b)
    eng = NumpyEngine(b - a, M, K, bias, N, rp, c, v)
    hd.scatter_state(eng, init, a, b, hier=True)
    ex = hd.Exchange(eng)
    hd.iterate(eng, ex, ITERS)
    errs = {
        "THETA_E": rel_err(eng.get_state("THETA_E"), final["THETA_E"][a:b]),
        "XI_E": rel_err(eng.get_state("XI_E"), final["XI_E"][a:b]),
        "BETA_E": rel_err(eng.get_state("BETA_E"), final["BETA_E"]),
        "ETA_E": rel_err(eng.get_state("ETA_E"), final["ETA_E"]),
    }
    if bias:
        errs["UBIAS_E"] = rel_err(eng.get_state("UBIAS_E"), final["UBIAS_E"][a:b])
        errs["IBIAS_E"] = rel_err(eng.get_state("IBIAS_E"), final["IBIAS_E"])
    q.put((rank, errs, (a, b)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("bias", [False, True])
def test_world1_double_matches_oracle(bias):
    q = mp.get_context("spawn").SimpleQueue()
    _run_rank(0, 1, bias, _free_port(), q)
    _, errs, _ = q.get()
    assert max(errs.values()) < 1e-10, errs


@pytest.mark.parametrize("world,bias", [(2, False), (2, True), (4, True)])
def test_gloo_ranks_match_oracle(world, bias):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_run_rank, args=(r, world, bias, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get() for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    ranges = sorted(r[2] for r in res)
    assert ranges[0][0] == 0 and ranges[-1][1] == N
    assert all(ranges[k][1] == ranges[k + 1][0] for k in range(world - 1))      # the user ranges tile [0, N)
    for _, errs, _ in res:
        assert max(errs.values()) < 1e-10, errs


def test_partition_balances_nnz_not_users():
    rowptr, _, _ = make_problem(2000, 300, 40000, 3, alpha_u=0.9)
    for world in (2, 4, 8):
        parts = hd.partition_users(rowptr, world)
        assert parts[0][0] == 0 and parts[-1][1] == 2000
        assert all(a < b for a, b in parts)
        assert all(parts[r][1] == parts[r + 1][0] for r in range(world - 1))
        loads = np.array([rowptr[b] - rowptr[a] for a, b in parts], float)
        assert loads.max() / loads.mean() < 1.25


def test_partition_degenerate():
    rowptr = np.array([0, 5, 5, 5, 5], np.int64)          # all nnz in the first user
    parts = hd.partition_users(rowptr, 4)
    assert parts == [(0, 1), (1, 2), (2, 3), (3, 4)]


def test_cpp_and_python_partition_agree():
    from hgaprec_amd import hostlib
    for seed, alpha in ((1, 0.5), (2, 0.9), (3, 1.3)):
        rowptr, _, _ = make_problem(1500, 200, 30000, seed, alpha_u=alpha)
        for world in (1, 2, 3, 8):
            assert hostlib.partition_users(rowptr, world) == hd.partition_users(rowptr, world)
    rowptr = np.array([0, 5, 5, 5, 5], np.int64)
    assert hostlib.partition_users(rowptr, 4) == hd.partition_users(rowptr, 4)
