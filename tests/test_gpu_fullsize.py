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
 * VALUES of a sample of owner rows (round 4): all of the above are blind to a gather of
   the wrong row -- a nonzero's phi sums to max(y, 1) whichever row was read -- so the raw
   phi sums of >= 500 users and >= 50 items (random, last, heaviest, rows cut into several
   segments, rows at the heavy / light bar and at tile boundaries) are recomputed in fp64
   from the exported Elog arrays (tests/rowcheck.py) and held to 1e-9; a deliberately
   damaged index proves the check sees what mass conservation cannot
C3 (10M x 1M, 1e9 nnz) runs whole and C5 as the real shard one of 8 GPUs owns;
both are generated, handed over and checked in HBM (hpf_upload_csr_device).
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
    assert int(rowptr[-1]) == cfg["nnz"]                 # the generator reaches BASELINE's nnz
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


def _t(rowptr, col, val, dev):
    """host CSR (numpy) -> torch tensors on the device, as tests/rowcheck.py takes them"""
    import torch
    return (torch.from_numpy(rowptr).to(dev), torch.from_numpy(col.view(np.int32)).to(dev),
            None if val is None else torch.from_numpy(val).to(dev))


def test_c2_sampled_rows_and_a_damaged_index(c2):
    """>= 500 users and >= 50 items of C2 value-checked after one iteration from the known start
    state; then ONE entry of the tiled item pass's index stream is pointed at another user: mass
    is still conserved, the owner row's values are not, and the sampled-row check says so."""
    import torch
    from tests import rowcheck
    cfg, rowptr, col, val, st = c2
    dev = torch.device("cuda", 0)
    rp, c, v = _t(rowptr, col, val, dev)
    D = _make(cfg, rowptr, col, val, st)
    wi = D.work_info()
    assert wi["tiles_item"] > 1 and wi["tile_rows_item"] > 0 and wi["heavy_min_nnz_item"] > 0, wi
    r = rowcheck.check_handle(D, rp, c, v, n_users=500, n_items=50, seed=1)
    assert r["rows_checked"]["users"] >= 500 and r["rows_checked"]["items"] >= 50
    assert r["ok"], r
    print("C2 sampled rows:", r)
    # the last position of the tiled array: the last heavy item's run inside the last tile of users
    nnz = int(rowptr[-1])
    old, owner = D.debug_poke_index(1, nnz - 1, 0)
    D.debug_poke_index(1, nnz - 1, old)
    wrong = (old + 123457) % cfg["n"]
    D.debug_poke_index(1, nnz - 1, wrong)
    assert owner < cfg["m"] and int((col == owner).sum()) >= wi["heavy_min_nnz_item"]       # a heavy item's tiled run
    bad = rowcheck.check_handle(D, rp, c, v, users=[0, 1], items=[owner, 0])
    assert not bad["ok"] and bad["worst_row"] == ["item", owner] and bad["max_rel_err"] > 1e-7, bad
    bs = D.get_state("BETA_SHAPE")
    mass = float(np.maximum(val, 1).astype(np.float64).sum())
    assert abs((bs - 0.3).sum() - mass) / mass < 1e-11                                      # mass conservation is blind to it
    D.debug_poke_index(1, nnz - 1, old)
    good = rowcheck.check_handle(D, rp, c, v, users=[0, 1], items=[owner, 0])
    assert good["ok"], good
    # the same on the row-major user side: the first nonzero of user 12345 pointed at another item
    pos = int(rowptr[12345])
    old_u, owner_u = D.debug_poke_index(0, pos, 0)
    assert owner_u == 12345 and old_u == int(col[pos])
    D.debug_poke_index(0, pos, (old_u + 4321) % cfg["m"])
    bad = rowcheck.check_handle(D, rp, c, v, users=[12345, 7], items=[0, 1])
    assert not bad["ok"] and bad["worst_row"] == ["user", 12345], bad
    D.debug_poke_index(0, pos, old_u)
    assert rowcheck.check_handle(D, rp, c, v, users=[12345, 7], items=[0, 1])["ok"]
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
    assert abs(int(rowptr[-1]) - cfg["nnz"]) <= 1e-3 * cfg["nnz"]      # 1e8, heavy users included
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
    # values of a sample of rows, bias slots included (both sides of C4 are tiled)
    from tests import rowcheck
    del ts, bs, ts2, tr, want
    r = rowcheck.check_handle(D, *_t(rowptr, col, val, dev), bias=True, n_users=500, n_items=50, seed=2)
    print("C4 sampled rows:", r, D.work_info())
    assert r["ok"], r
    D.close()


def _device_model(cfg, n_loc, rowptr, col, val, row0, n_total, n_ranks=1, rank=0, xbuf=None, seeds=(1, 2, 3, 4)):
    """a handle fed entirely from HBM: hpf_upload_csr_device + hpf_set_state_device;
    rows [row0, row0 + n_loc) of the hashed bench-mode start state"""
    import torch
    from hgaprec_amd import synth
    from hgaprec_amd.capi import Hpf
    dev = rowptr.device
    m, K = cfg["m"], cfg["K"]
    D = Hpf(n_loc, m, K, hier=True, binary=cfg["binary"], n_ranks=n_ranks, rank=rank, n_users_total=n_total)
    if xbuf is not None:                 # a list: receives the caller-owned exchange buffer [m x ld | ld]
        x = torch.zeros(D.exchange_count(), dtype=torch.float64, device=dev)
        D.bind_exchange_buffer(x.data_ptr(), x.numel())
        xbuf.append(x)
    D.upload_csr_device(rowptr, col, val)
    st = synth.initial_state_device(n_loc, K, seeds[0], dev, row0=row0)
    D.set_state_device("THETA_E", st["E"]); D.set_state_device("THETA_ELOG", st["Elog"])
    st = synth.initial_state_device(m, K, seeds[1], dev)
    D.set_state_device("BETA_E", st["E"]); D.set_state_device("BETA_ELOG", st["Elog"])
    D.set_state_device("XI_E", synth.initial_state_device(n_loc, K, seeds[2], dev, prior_v=K, row0=row0)["E"])
    D.set_state_device("ETA_E", synth.initial_state_device(m, K, seeds[3], dev, prior_v=K)["E"])
    del st
    torch.cuda.empty_cache()
    return D


def _row_mass(rowptr, w):
    """sum of w over each CSR row (torch, on device); w = None counts nonzeros"""
    import torch
    if w is None:
        return (rowptr[1:] - rowptr[:-1]).to(torch.float64)
    c = torch.zeros(w.numel() + 1, dtype=torch.float64, device=w.device)
    torch.cumsum(w, 0, out=c[1:])
    return c[rowptr[1:]] - c[rowptr[:-1]]


def test_c3_whole_properties():
    """BASELINE config C3 WHOLE on one GPU (10M x 1M, 1e9 nnz, K=100, -hier;
    ~60 GB resident), everything handed over and checked in HBM:
     * mass conservation per user and per item after each of 2 iterations
       (every nonzero's phi sums to its rating),
     * the rate identity theta_rate[u,k] = E[xi_u] + sum_i E[beta_ik] and the
       xi update xi_rate = 0.3 + rowsum(E theta) (hgaprec.cc:1370-1414),
     * a second run gives identical bits,
     * the same problem cut into 2 user shards by nnz (partition_users), their
       exchange buffers summed, agrees with the single run to 1e-10."""
    import torch
    from hgaprec_amd import synth
    from hgaprec_amd.dist import partition_users
    cfg = dict(synth.CONFIGS["C3"])
    n, m, K = cfg["n"], cfg["m"], cfg["K"]
    dev = torch.device("cuda", 0)
    rowptr, col, val = synth.generate_device(n, m, cfg["nnz"], cfg["alpha_u"], cfg["alpha_i"], seed=cfg["seed"], device=dev)
    torch.cuda.empty_cache()
    nnz = int(rowptr[-1])
    assert abs(nnz - cfg["nnz"]) <= 1e-3 * cfg["nnz"]
    w = torch.clamp(val, min=1).to(torch.float64)
    mass_u = _row_mass(rowptr, w)
    mass_i = torch.bincount(col.to(torch.int64), weights=w, minlength=m)
    total = float(w.sum())
    del w

    def run(check):
        D = _device_model(cfg, n, rowptr, col, val, 0, n)
        xi0 = synth.initial_state_device(n, K, 3, dev, prior_v=K)["E"]
        c0 = synth.initial_state_device(m, K, 2, dev)["E"].sum(0)
        for it in range(2):
            D.iterate(1)
            if not check:
                continue
            ts = D.get_state_device("THETA_SHAPE", dev)
            got_u = (ts - 0.3).sum(1)
            assert float(((got_u - mass_u).abs() / mass_u.clamp(min=1.0)).max()) < 1e-11
            assert abs(float(got_u.sum()) - total) / total < 1e-11
            if it == 0:
                tr = D.get_state_device("THETA_RATE", dev)
                want = xi0[:, None] + c0[None, :]
                assert float(((tr - want).abs() / want).max()) < 1e-12
                te = D.get_state_device("THETA_E", dev)
                assert float(((te - ts / tr).abs() / te).max()) < 1e-15
                xr = D.get_state_device("XI_RATE", dev)
                assert float(((xr - (0.3 + te.sum(1))).abs() / xr).max()) < 1e-13
                del tr, want, te, xr
            del ts, got_u
            bs = D.get_state_device("BETA_SHAPE", dev)
            got_i = (bs - 0.3).sum(1)
            assert float(((got_i - mass_i).abs() / mass_i.clamp(min=1.0)).max()) < 1e-11
            del bs, got_i
        wi = D.work_info()
        tm = D.mean_timing(1)
        out = (D.get_state_device("THETA_E", dev), D.get_state_device("BETA_E", dev))
        if check:            # values of a sample of rows (a third iteration; `out` is already taken)
            from tests import rowcheck
            r = rowcheck.check_handle(D, rowptr, col, val, n_users=500, n_items=50, seed=3)
            print("C3 whole sampled rows:", r)
            assert r["ok"], r
        D.close()
        torch.cuda.empty_cache()
        return out, wi, tm

    (te1, be1), wi, tm = run(True)
    assert wi["nnz"] == nnz and wi["item_long_rows"] > 0
    print(f"C3 whole: {nnz} nnz, {tm['iteration_ms']:.1f} ms/iteration "
          f"(phi item {tm['phi_item_ms']:.1f}, phi user {tm['phi_user_ms']:.1f})")
    (te2, be2), _, _ = run(False)
    assert torch.equal(te1, te2) and torch.equal(be1, be2)
    del te2, be2

    parts = partition_users(rowptr.cpu().numpy(), 2)
    shards, xb = [], []
    for r, (a, b) in enumerate(parts):
        lo, hi = int(rowptr[a]), int(rowptr[b])
        shards.append(_device_model(cfg, b - a, (rowptr[a:b + 1] - rowptr[a]).contiguous(), col[lo:hi], val[lo:hi],
                                    a, n, 2, r, xb))
    assert abs((int(rowptr[parts[0][1]]) - nnz // 2)) < 4 * m                # balanced by nonzeros, not by users
    for _ in range(2):
        for S in shards:
            S.iterate_local()
        for S in shards:
            S.synchronize()
        tot = xb[0] + xb[1]
        for S, x in zip(shards, xb):
            x.copy_(tot)
        torch.cuda.synchronize()
        for S in shards:
            S.iterate_global()
    for (a, b), S in zip(parts, shards):
        got = S.get_state_device("THETA_E", dev)
        assert float(((got - te1[a:b]).abs() / te1[a:b]).max()) < 1e-10
        gb = S.get_state_device("BETA_E", dev)
        assert float(((gb - be1).abs() / be1).max()) < 1e-10
        del got, gb
        S.close()


def test_c5_shard_full_size():
    """BASELINE config C5 (50M x 2M, 5e9 nnz, K=50, -hier -binary-data,
    heavy-tailed degrees alpha = 0.9 / 1.1): the REAL shard one of 8 GPUs owns --
    the first of partition_users' 8 nnz-balanced user ranges of the whole matrix
    (about 6.25M users, 6e8 nonzeros), all 2M items.  Blockbuster items are
    rated by millions of the shard's users, so their rows take the two-level
    combine (item_huge_rows > 0); per-user and per-item mass must still equal
    the degrees exactly."""
    import torch
    from hgaprec_amd import synth
    from hgaprec_amd.dist import partition_users
    cfg = dict(synth.CONFIGS["C5"])
    n, m, K = cfg["n"], cfg["m"], cfg["K"]
    dev = torch.device("cuda", 0)
    deg = synth.degrees(n, m, cfg["nnz"], cfg["alpha_u"], cfg["seed"], dev)
    planned = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(deg, 0, out=planned[1:])
    a, b = partition_users(planned.cpu().numpy(), 8)[0]
    planned_nnz = int(planned[b] - planned[a])
    del planned
    rowptr, col, val = synth.generate_device(n, m, cfg["nnz"], cfg["alpha_u"], cfg["alpha_i"], seed=cfg["seed"],
                                             device=dev, binary=True, user_range=(a, b), deg=deg)
    del deg
    torch.cuda.empty_cache()
    nnz = int(rowptr[-1])
    assert nnz == planned_nnz                            # every user reaches the planned degree
    assert val is None and 5.5e8 < nnz < 7.0e8 and 4_000_000 < b - a < 7_500_000
    D = _device_model(cfg, b - a, rowptr, col, None, a, n, n_ranks=8, rank=0)
    wi = D.work_info()
    assert wi["item_huge_rows"] > 0, wi          # a blockbuster item: > 256 segments of 512 raters
    assert wi["user_long_rows"] > 0, wi          # users at the m/2 degree cap
    deg_u = _row_mass(rowptr, None)
    deg_i = torch.bincount(col.to(torch.int64), minlength=m).to(torch.float64)
    x = torch.zeros(1, device=dev)
    for it in range(2):
        D.iterate_local()
        D.iterate_global()                       # this rank's sums only: a 1-of-8 shard run on its own
        ts = D.get_state_device("THETA_SHAPE", dev)
        assert float((((ts - 0.3).sum(1) - deg_u).abs() / deg_u.clamp(min=1.0)).max()) < 1e-11
        del ts
        bs = D.get_state_device("BETA_SHAPE", dev)
        got = (bs - 0.3).sum(1)
        assert float(((got - deg_i).abs() / deg_i.clamp(min=1.0)).max()) < 1e-11
        assert abs(float(got.sum()) - nnz) / nnz < 1e-11
        del bs, got
    from tests import rowcheck
    r = rowcheck.check_handle(D, rowptr, col, None, n_users=500, n_items=50, seed=4)
    print("C5 shard sampled rows:", r, wi)
    assert r["ok"], r
    tm = D.mean_timing(1)
    print(f"C5 shard 0 of 8: users [{a}, {b}), {nnz} nnz, {wi['item_huge_rows']} huge item rows, "
          f"{tm['iteration_ms']:.1f} ms/iteration (phi item {tm['phi_item_ms']:.1f} + combine "
          f"{tm['combine_item_ms']:.2f}, phi user {tm['phi_user_ms']:.1f})")
    # held-out likelihood in its binary form: log(1 - exp(-s)) per pair, finite and negative
    nl = b - a
    hu = np.arange(0, nl, max(1, nl // 5000), dtype=np.uint32)
    hi = (hu.astype(np.uint64) * 7919 % m).astype(np.uint32)
    s, c = D.heldout_ll(hu, hi, np.ones(hu.size, np.int32))
    assert c == hu.size and np.isfinite(s) and s < 0
    D.close()


def test_more_than_2_31_nonzeros():
    """64-bit indexing end to end: 4M users x 200K items with ~2.4e9 nonzeros
    (-binary-data, K=4 keeps the state small).  Per-user and per-item mass must
    equal the degrees -- any 32-bit wrap in the work lists, the device CSC build
    (radix tiles, scan) or the kernels' nonzero offsets would break it."""
    import torch
    from hgaprec_amd import synth
    n, m, nnz, K = 4_000_000, 200_000, 2_500_000_000, 4
    dev = torch.device("cuda", 0)
    rowptr, col, val = synth.generate_device(n, m, nnz, 0.4, 0.7, seed=77, device=dev, binary=True)
    torch.cuda.empty_cache()
    assert int(rowptr[-1]) > 2**31 and val is None
    cfg = dict(m=m, K=K, binary=True)
    D = _device_model(cfg, n, rowptr, col, None, 0, n)
    D.iterate(2)
    ts, bs = D.get_state_device("THETA_SHAPE", dev), D.get_state_device("BETA_SHAPE", dev)
    deg_u = _row_mass(rowptr, None)
    assert float((((ts - 0.3).sum(1) - deg_u).abs() / deg_u.clamp(min=1.0)).max()) < 1e-11
    deg_i = torch.zeros(m, dtype=torch.float64, device=dev)
    step = 1 << 29
    for a in range(0, col.numel(), step):                   # bincount in slices: bounded temporaries
        deg_i += torch.bincount(col[a:a + step].to(torch.int64), minlength=m)
    assert float((((bs - 0.3).sum(1) - deg_i).abs() / deg_i.clamp(min=1.0)).max()) < 1e-11
    assert abs(float((ts - 0.3).sum()) - float(rowptr[-1])) / float(rowptr[-1]) < 1e-11
    del ts, bs
    from tests import rowcheck
    r = rowcheck.check_handle(D, rowptr, col, None, n_users=500, n_items=50, seed=5)
    print(">2^31 nnz sampled rows:", r)
    assert r["ok"], r
    # the item-major view itself, on a slice: users ascending inside each of the first items
    colptr, users, _ = D.get_csc(int(rowptr[-1]), with_vals=False)
    assert colptr[-1] == int(rowptr[-1]) and np.array_equal(np.diff(colptr), deg_i.cpu().numpy().astype(np.int64))
    for i in (0, 1, m // 2, m - 1):
        seg = users[colptr[i]:colptr[i + 1]].astype(np.int64)
        assert np.all(np.diff(seg) > 0)
    D.close()


def test_tiled_lists_past_2_32_nonzeros():
    """a TILED side with more than 2^32 nonzeros (round 4: the guard that left such a side row-major is gone;
    positions are 64-bit throughout, only the number of segments and partial slots has to stay below 2^31):
    6M users x 300K items, 4.4e9 nonzeros, -binary-data, K = 4.  Mass per row and the values of a sample of
    rows -- an index that wrapped at 2^32 would gather the wrong row or drop a run."""
    import torch
    from hgaprec_amd import synth
    from tests import rowcheck
    n, m, nnz, K = 6_000_000, 300_000, 4_400_000_000, 4
    dev = torch.device("cuda", 0)
    rowptr, col, val = synth.generate_device(n, m, nnz, 0.4, 0.7, seed=79, device=dev, binary=True)
    torch.cuda.empty_cache()
    assert int(rowptr[-1]) > 2**32 and val is None
    cfg = dict(m=m, K=K, binary=True)
    D = _device_model(cfg, n, rowptr, col, None, 0, n)
    wi = D.work_info()
    assert wi["tiles_item"] > 1 and wi["nnz"] == int(rowptr[-1]), wi
    D.iterate(1)
    ts = D.get_state_device("THETA_SHAPE", dev)
    deg_u = _row_mass(rowptr, None)
    assert float((((ts - 0.3).sum(1) - deg_u).abs() / deg_u.clamp(min=1.0)).max()) < 1e-11
    del ts
    bs = D.get_state_device("BETA_SHAPE", dev)
    deg_i = rowcheck.item_degrees(col, m).to(torch.float64)
    assert float((((bs - 0.3).sum(1) - deg_i).abs() / deg_i.clamp(min=1.0)).max()) < 1e-11
    del bs
    r = rowcheck.check_handle(D, rowptr, col, None, n_users=300, n_items=40, seed=6)
    print(">2^32 nnz, tiled:", wi, r)
    assert r["ok"], r
    D.close()
