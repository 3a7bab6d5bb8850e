"""-m gpu: ranking evaluation kernels (fp64 MFMA scoring, exact top-N by radix
select, rank queries) against numpy on the same state, and the oracle's
compute_precision / compute_itemrank through the CLI."""
import numpy as np
import pytest

from tests.util import copy_state, make_problem

pytestmark = pytest.mark.gpu


def _setup(orc, n, m, K, nnz, bias, seed, iters=2):
    from hgaprec_amd.capi import Hpf
    rowptr, col, val = make_problem(n, m, nnz, seed)
    M = orc.Model(n, m, K, True, bias, False)
    M.set_csr(rowptr, col, val)
    M.initialize(seed)
    D = Hpf(n, m, K, hier=True, bias=bias)
    D.upload_csr(rowptr, col, val)
    copy_state(M, D, True, bias)
    M.iterate(iters)
    D.iterate(iters)
    return M, D, rowptr, col, val


def _ref_scores(M, users, bias):
    s = M.state("THETA_E")[users] @ M.state("BETA_E").T
    if bias:
        s = s + M.state("UBIAS_E")[users][:, None] + M.state("IBIAS_E")[None, :]
    return s


@pytest.mark.parametrize("K,bias,m", [(5, False, 200), (20, True, 333), (100, False, 1000), (7, True, 70)])
def test_scores_match_dense_product(orc, K, bias, m):
    n = 150
    M, D, *_ = _setup(orc, n, m, K, 4000, bias, seed=K)
    users = np.array([0, 3, 17, 149, 5, 5, 80] + list(range(20, 45)), np.uint32)   # 32 rows: two 16-row tiles
    got = D.scores(users)
    want = _ref_scores(M, users, bias)
    assert np.max(np.abs(got - want) / want) < 1e-12


def _masked_ref(M, users, bias, rowptr, col, val, mask):
    s = _ref_scores(M, users, bias)
    for b, u in enumerate(users):
        js = np.arange(rowptr[u], rowptr[u + 1])
        s[b, col[js][val[js] > 0]] = 0.0
        s[b, mask[b]] = 0.0
    return s


def test_topn_and_ranks_match_stable_sort(orc):
    n, m, K = 120, 500, 10
    M, D, rowptr, col, val = _setup(orc, n, m, K, 6000, True, seed=4)
    rng = np.random.default_rng(0)
    users = np.sort(rng.choice(n, 40, replace=False)).astype(np.uint32)
    mask = [np.sort(rng.choice(m, rng.integers(0, 6), replace=False)).astype(np.uint32) for _ in users]
    mptr = np.zeros(users.size + 1, np.uint64)
    mptr[1:] = np.cumsum([x.size for x in mask])
    mitems = np.concatenate(mask).astype(np.uint32) if mptr[-1] else np.zeros(0, np.uint32)
    items, sc = D.rank_topn(users, 100, mptr, mitems)
    dev = D.scores(users)                       # exact device scores, then mask like the kernel
    for b, u in enumerate(users):
        js = np.arange(rowptr[u], rowptr[u + 1])
        dev[b, col[js][val[js] > 0]] = 0.0
        dev[b, mask[b]] = 0.0
    for b in range(users.size):
        order = np.argsort(-dev[b], kind="stable")           # desc, ties by ascending index
        assert np.array_equal(items[b], order[:100].astype(np.uint32))
        assert np.array_equal(sc[b], dev[b][order[:100]])
    # and the device scores themselves agree with the oracle's state
    ref = _masked_ref(M, users, True, rowptr, col, val, mask)
    assert np.max(np.abs(dev - ref)) < 1e-12 * np.max(ref)
    # rank queries: every item of three rows
    qs = np.repeat(np.array([0, 7, 39], np.uint32), m)
    qi = np.tile(np.arange(m, dtype=np.uint32), 3)
    rank, rsc = D.item_ranks(users, qs, qi, mptr, mitems)
    for b in (0, 7, 39):
        order = np.argsort(-dev[b], kind="stable")
        pos = np.empty(m, np.int64)
        pos[order] = np.arange(m)
        assert np.array_equal(rank[qs == b], pos.astype(np.uint32))
        assert np.array_equal(rsc[qs == b], dev[b])


def test_topn_more_than_items_and_all_masked(orc):
    n, m, K = 40, 30, 4
    M, D, rowptr, col, val = _setup(orc, n, m, K, 300, False, seed=9)
    users = np.array([1, 2], np.uint32)
    mptr = np.array([0, m, m], np.uint64)                     # user 1: every item masked
    mitems = np.arange(m, dtype=np.uint32)
    items, sc = D.rank_topn(users, 100, mptr, mitems)
    assert np.array_equal(items[0, :m], np.arange(m, dtype=np.uint32))   # all zero: index order
    assert np.all(sc[0] == 0.0) and np.all(items[0, m:] == 0xFFFFFFFF)
    assert np.all(items[1, m:] == 0xFFFFFFFF) and np.all(np.diff(sc[1, :m]) <= 0)
