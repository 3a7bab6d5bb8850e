"""-m gpu: BASELINE-size runs (C2: 1M x 100K, 5e7 nnz, K=100) checked through
size-independent properties of the path -- the oracle would need minutes per
iteration at this size:

 * mass conservation: each nonzero's phi sums to max(y,1), so after one
   iteration  sum(theta_shape - 0.3) == sum(beta_shape - 0.3) == sum_j max(y_j,1)
 * the rate identity  theta_rate[u,k] == E_xi_used[u] + sum_i E[beta_ik]
   re-derived on the host from exported state
 * xi update:  xi_rate == 0.3 + rowsum(E theta),  xi_shape == 0.3 + 0.3 K
 * user sharding: 2 logical ranks (host-summed exchange) == 1 rank to 1e-10
 * bit-identical repeat runs
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c2():
    import torch
    from hgaprec_amd import synth
    cfg = dict(synth.CONFIGS["C2"])
    dev = torch.device("cuda", 0)
    rowptr, col, val = synth.generate(cfg["n"], cfg["m"], cfg["nnz"], cfg["alpha_u"], cfg["alpha_i"],
                                      seed=cfg["seed"], device=dev)
    st = {
        "theta": synth.initial_state(cfg["n"], cfg["K"], 1, dev),
        "beta": synth.initial_state(cfg["m"], cfg["K"], 2, dev),
        "xi": synth.initial_state(cfg["n"], cfg["K"], 3, dev, prior_v=cfg["K"]),
        "eta": synth.initial_state(cfg["m"], cfg["K"], 4, dev, prior_v=cfg["K"]),
    }
    torch.cuda.empty_cache()
    return cfg, rowptr, col, val, st


def _make(cfg, rowptr, col, val, st, a=0, b=None, n_ranks=1, rank=0):
    from hgaprec_amd.capi import Hpf
    n = cfg["n"]
    b = n if b is None else b
    D = Hpf(b - a, cfg["m"], cfg["K"], hier=True, n_ranks=n_ranks, rank=rank, n_users_total=n)
    lo, hi = rowptr[a], rowptr[b]
    D.upload_csr(rowptr[a:b + 1] - rowptr[a], col[lo:hi], val[lo:hi])
    D.set_state("THETA_E", st["theta"]["E"][a:b]); D.set_state("THETA_ELOG", st["theta"]["Elog"][a:b])
    D.set_state("BETA_E", st["beta"]["E"]); D.set_state("BETA_ELOG", st["beta"]["Elog"])
    D.set_state("XI_E", st["xi"]["E"][a:b]); D.set_state("ETA_E", st["eta"]["E"])
    return D


def test_c2_one_iteration_properties(c2):
    cfg, rowptr, col, val, st = c2
    K = cfg["K"]
    D = _make(cfg, rowptr, col, val, st)
    D.iterate(1)
    ts, bs = D.get_state("THETA_SHAPE"), D.get_state("BETA_SHAPE")
    mass = float(np.maximum(val, 1).astype(np.float64).sum())
    assert abs((ts - 0.3).sum() - mass) / mass < 1e-11
    assert abs((bs - 0.3).sum() - mass) / mass < 1e-11
    # per-row mass on a sample of users: sum_k (shape-0.3) == sum of the user's ratings
    for u in (0, 1, 12345, cfg["n"] - 1):
        want = float(np.maximum(val[rowptr[u]:rowptr[u + 1]], 1).sum())
        assert abs((ts[u] - 0.3).sum() - want) < 1e-9 * max(want, 1.0)
    deg_mass = np.bincount(col, weights=np.maximum(val, 1).astype(np.float64), minlength=cfg["m"])
    assert np.max(np.abs((bs - 0.3).sum(1) - deg_mass) / np.maximum(deg_mass, 1.0)) < 1e-11
    # rates
    te, tr = D.get_state("THETA_E"), D.get_state("THETA_RATE")
    c = st["beta"]["E"].sum(0)
    want = st["xi"]["E"][:, None] + c[None, :]
    assert np.max(np.abs(tr - want) / want) < 1e-12
    assert np.max(np.abs(te - ts / tr) / te) < 1e-15
    xr = D.get_state("XI_RATE")
    assert np.max(np.abs(xr - (0.3 + te.sum(1))) / xr) < 1e-13
    assert np.all(D.get_state("XI_SHAPE") == 0.3 + K * 0.3)
    br = D.get_state("BETA_RATE")
    d = te.sum(0)
    assert np.max(np.abs(br - (st["eta"]["E"][:, None] + d[None, :])) / br) < 1e-11
    # Elog export = psi(shape) - log(rate)
    from scipy.special import digamma
    el = D.get_state("THETA_ELOG")[:1000]
    assert np.max(np.abs(el - (digamma(ts[:1000]) - np.log(tr[:1000])))) < 1e-13
    D.close()


def test_c2_repeat_runs_are_bit_identical_and_shards_agree(c2):
    cfg, rowptr, col, val, st = c2
    outs = []
    for _ in range(2):
        D = _make(cfg, rowptr, col, val, st)
        D.iterate(2)
        outs.append((D.get_state("THETA_E"), D.get_state("BETA_E")))
        D.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])

    from hgaprec_amd.dist import partition_users
    parts = partition_users(rowptr, 2)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    shards = [_make(cfg, rowptr, col, val, st, a, b, 2, r) for r, (a, b) in enumerate(parts)]
    for _ in range(2):
        bufs = []
        for S in shards:
            S.iterate_local(); S.synchronize()
            p, cnt = S.exchange_buffer()
            h = np.empty(cnt, np.float64)
            assert hip.hipMemcpy(h.ctypes.data, p, cnt * 8, 2) == 0
            bufs.append(h)
        tot = bufs[0] + bufs[1]
        for S in shards:
            p, cnt = S.exchange_buffer()
            assert hip.hipMemcpy(p, tot.ctypes.data, cnt * 8, 1) == 0
            S.iterate_global()
    te, be = outs[0]
    for (a, b), S in zip(parts, shards):
        got = S.get_state("THETA_E")
        assert np.max(np.abs(got - te[a:b]) / te[a:b]) < 1e-10
        assert np.max(np.abs(S.get_state("BETA_E") - be) / be) < 1e-10
        S.close()


def test_c4_bias_k200_properties():
    """BASELINE config C4 at full size (480 189 x 17 770, ~9.3e7 nnz, K=200,
    -hier -bias): with the bias slots every nonzero's phi still sums to
    max(y, 1) over K + 2 slots; the user side receives slots 0..K-1 and K, the
    item side 0..K-1 and K+1 (hgaprec.cc:1357-1366)."""
    import torch
    from hgaprec_amd import synth
    from hgaprec_amd.capi import Hpf
    cfg = dict(synth.CONFIGS["C4"])
    n, m, K = cfg["n"], cfg["m"], cfg["K"]
    dev = torch.device("cuda", 0)
    rowptr, col, val = synth.generate(n, m, cfg["nnz"], cfg["alpha_u"], cfg["alpha_i"], seed=cfg["seed"], device=dev)
    D = Hpf(n, m, K, hier=True, bias=True)
    D.upload_csr(rowptr, col, val)
    st = synth.initial_state(n, K, 1, dev)
    D.set_state("THETA_E", st["E"]); D.set_state("THETA_ELOG", st["Elog"])
    st = synth.initial_state(m, K, 2, dev)
    D.set_state("BETA_E", st["E"]); D.set_state("BETA_ELOG", st["Elog"])
    beta_E0 = st["E"]
    xi0 = synth.initial_state(n, K, 3, dev, prior_v=K)["E"]
    D.set_state("XI_E", xi0)
    D.set_state("ETA_E", synth.initial_state(m, K, 4, dev, prior_v=K)["E"])
    st = synth.initial_state(n, K, 5, dev, prior_v=m)
    D.set_state("UBIAS_E", st["E"]); D.set_state("UBIAS_ELOG", st["Elog"])
    st = synth.initial_state(m, K, 6, dev, prior_v=n)
    D.set_state("IBIAS_E", st["E"]); D.set_state("IBIAS_ELOG", st["Elog"])
    del st
    torch.cuda.empty_cache()
    D.iterate(1)
    ts, bs = D.get_state("THETA_SHAPE"), D.get_state("BETA_SHAPE")
    us, is_ = D.get_state("UBIAS_SHAPE"), D.get_state("IBIAS_SHAPE")
    w = np.maximum(val, 1).astype(np.float64)
    mass = float(w.sum())
    common = (ts - 0.3).sum()
    assert abs(common - (bs - 0.3).sum()) / mass < 1e-11              # slots 0..K-1 reach both sides
    assert abs(common + (us - 0.3).sum() + (is_ - 0.3).sum() - mass) / mass < 1e-11
    # per item: slots 0..K-1 plus the item-bias slot, plus the user-bias mass that went to its raters
    assert np.all(us > 0.3) and np.all(is_ > 0.3)
    # rates: theta as in -hier, biases constant (hgaprec.cc:1389,1393)
    tr = D.get_state("THETA_RATE")
    want = xi0[:, None] + beta_E0.sum(0)[None, :]
    assert np.max(np.abs(tr - want) / want) < 1e-12
    ue, ie = D.get_state("UBIAS_E"), D.get_state("IBIAS_E")
    assert np.max(np.abs(ue - us / (0.3 + m)) / ue) < 1e-15
    assert np.max(np.abs(ie - is_ / (0.3 + n)) / ie) < 1e-15
    # a second iteration keeps the books balanced too (new W, long item rows)
    D.iterate(1)
    ts2 = D.get_state("THETA_SHAPE")
    tot = (ts2 - 0.3).sum() + (D.get_state("UBIAS_SHAPE") - 0.3).sum() + (D.get_state("IBIAS_SHAPE") - 0.3).sum()
    assert abs(tot - mass) / mass < 1e-11
    D.close()


def test_c5_shape_binary_k50_properties():
    """BASELINE config C5's shape (-hier -binary-data, K=50, heavy-tailed
    degrees alpha = 0.9 / 1.1) at 1/50 of its size (one GPU's share of a
    50-GPU run: 1M x 40K, ~1e8 nnz): every nonzero is a 1, phi sums to 1."""
    import torch
    from hgaprec_amd import synth
    from hgaprec_amd.capi import Hpf
    cfg = dict(synth.CONFIGS["C5"])
    n, m, nnz, K = cfg["n"] // 50, cfg["m"] // 50, cfg["nnz"] // 50, cfg["K"]
    dev = torch.device("cuda", 0)
    rowptr, col, val = synth.generate(n, m, nnz, cfg["alpha_u"], cfg["alpha_i"], seed=cfg["seed"], device=dev,
                                      binary=True)
    assert val is None
    D = Hpf(n, m, K, hier=True, binary=True)
    D.upload_csr(rowptr, col, None)
    st = synth.initial_state(n, K, 1, dev)
    D.set_state("THETA_E", st["E"]); D.set_state("THETA_ELOG", st["Elog"])
    st = synth.initial_state(m, K, 2, dev)
    D.set_state("BETA_E", st["E"]); D.set_state("BETA_ELOG", st["Elog"])
    D.set_state("XI_E", synth.initial_state(n, K, 3, dev, prior_v=K)["E"])
    D.set_state("ETA_E", synth.initial_state(m, K, 4, dev, prior_v=K)["E"])
    del st
    torch.cuda.empty_cache()
    D.iterate(2)
    ts, bs = D.get_state("THETA_SHAPE"), D.get_state("BETA_SHAPE")
    mass = float(rowptr[-1])
    assert abs((ts - 0.3).sum() - mass) / mass < 1e-11
    assert abs((bs - 0.3).sum() - mass) / mass < 1e-11
    deg_u = np.diff(rowptr).astype(np.float64)
    assert np.max(np.abs((ts - 0.3).sum(1) - deg_u) / np.maximum(deg_u, 1.0)) < 1e-11
    deg_i = np.bincount(col, minlength=m).astype(np.float64)
    assert np.max(np.abs((bs - 0.3).sum(1) - deg_i) / np.maximum(deg_i, 1.0)) < 1e-11
    # held-out likelihood in its binary form: log(1 - exp(-s)) per pair, finite and negative
    hu = np.arange(0, n, max(1, n // 5000), dtype=np.uint32)
    hi = (hu.astype(np.uint64) * 7919 % m).astype(np.uint32)
    s, c = D.heldout_ll(hu, hi, np.ones(hu.size, np.int32))
    assert c == hu.size and np.isfinite(s) and s < 0
    D.close()


def test_more_than_2_31_nonzeros():
    """64-bit indexing end to end: 4M users x 200K items with ~2.4e9 nonzeros
    (-binary-data, K=4 keeps the state small).  Per-user and per-item mass must
    equal the degrees -- any 32-bit wrap in the work lists, the CSC build or the
    kernels' nonzero offsets would break it."""
    import torch
    from hgaprec_amd import synth
    from hgaprec_amd.capi import Hpf
    n, m, nnz, K = 4_000_000, 200_000, 2_500_000_000, 4
    dev = torch.device("cuda", 0)
    rowptr, col, val = synth.generate(n, m, nnz, 0.4, 0.7, seed=77, device=dev, binary=True)
    torch.cuda.empty_cache()
    assert int(rowptr[-1]) > 2**31 and val is None
    D = Hpf(n, m, K, hier=True, binary=True)
    D.upload_csr(rowptr, col, None)
    st = synth.initial_state(n, K, 1, dev)
    D.set_state("THETA_E", st["E"]); D.set_state("THETA_ELOG", st["Elog"])
    st = synth.initial_state(m, K, 2, dev)
    D.set_state("BETA_E", st["E"]); D.set_state("BETA_ELOG", st["Elog"])
    D.set_state("XI_E", synth.initial_state(n, K, 3, dev, prior_v=K)["E"])
    D.set_state("ETA_E", synth.initial_state(m, K, 4, dev, prior_v=K)["E"])
    D.iterate(2)
    ts, bs = D.get_state("THETA_SHAPE"), D.get_state("BETA_SHAPE")
    deg_u = np.diff(rowptr).astype(np.float64)
    assert np.max(np.abs((ts - 0.3).sum(1) - deg_u) / np.maximum(deg_u, 1.0)) < 1e-11
    deg_i = np.zeros(m, np.float64)
    step = 1 << 28
    for a in range(0, col.size, step):                      # bincount in slices: bounded host memory
        deg_i += np.bincount(col[a:a + step], minlength=m)
    assert np.max(np.abs((bs - 0.3).sum(1) - deg_i) / np.maximum(deg_i, 1.0)) < 1e-11
    assert abs((ts - 0.3).sum() - float(rowptr[-1])) / float(rowptr[-1]) < 1e-11
    D.close()
