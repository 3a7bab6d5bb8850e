"""-m gpu: the hand-over path -- hpf_upload_csr / hpf_upload_csr_device /
hpf_set_state(_device) / hpf_get_state(_device) / hpf_get_csc.

The item-major (CSC) view is built in HBM by a hand-written stable LSD radix
sort on the item id (hpf_build.hpp).  Its contract: bit for bit the result of a
serial counting sort of the CSR by item -- users ascending inside an item, the
order in which the reference's serial loop reaches them (hgaprec.cc:1340-1345);
numpy's stable argsort is that order.
"""
import numpy as np
import pytest

from tests.util import compare_states, copy_state, make_problem, rel_err

pytestmark = pytest.mark.gpu


def _want_csc(rowptr, col, val, m):
    n = rowptr.size - 1
    users = np.repeat(np.arange(n, dtype=np.uint32), np.diff(rowptr))
    order = np.argsort(col, kind="stable")
    colptr = np.zeros(m + 1, np.int64)
    colptr[1:] = np.cumsum(np.bincount(col, minlength=m))
    return colptr, users[order], None if val is None else val[order]


def _random_csr(rng, n, m, nnz, heavy_item=False, empty_rows=False):
    deg = rng.multinomial(nnz, rng.dirichlet(np.full(n, 0.3)))
    if empty_rows:
        deg[rng.integers(0, n, size=n // 5)] = 0
    rowptr = np.zeros(n + 1, np.int64)
    rowptr[1:] = np.cumsum(deg)
    p = (np.arange(m) + 1.0) ** -1.1
    p = p[rng.permutation(m)] / p.sum()
    col = rng.choice(m, size=int(rowptr[-1]), p=p).astype(np.uint32)      # duplicates allowed: file order kept
    if heavy_item:
        col[rng.random(col.size) < 0.4] = np.uint32(m // 3)
    val = rng.integers(0, 256, size=col.size).astype(np.uint8)
    return rowptr, col, val


# m chooses the number of radix passes (8 bits each): 1, 2, 3 and 4
@pytest.mark.parametrize("n,m,nnz,kw", [
    (50, 1, 300, {}),                                  # a single item: the sort is the identity
    (300, 200, 20_000, dict(heavy_item=True)),         # 1 pass, several tiles, one item holds 40 %
    (2_000, 40_000, 300_000, dict(empty_rows=True)),   # 2 passes
    (5_000, 300_000, 1_200_000, {}),                   # 3 passes (m > 2^16), > 256 tiles
    (100, 17_000_000, 50_000, {}),                     # 4 passes (m > 2^24)
])
@pytest.mark.parametrize("binary", [False, True])
def test_device_csc_equals_serial_counting_sort(n, m, nnz, kw, binary):
    from hgaprec_amd.capi import Hpf
    rng = np.random.default_rng(n * 7 + m)
    rowptr, col, val = _random_csr(rng, n, m, nnz, **kw)
    if binary:
        val = None
    D = Hpf(n, m, 2, hier=False, binary=binary)
    D.upload_csr(rowptr, col, val)
    colptr, users, vals = D.get_csc(int(rowptr[-1]), with_vals=not binary)
    wp, wu, wv = _want_csc(rowptr, col, val, m)
    assert np.array_equal(colptr, wp)
    assert np.array_equal(users, wu)
    if not binary:
        assert np.array_equal(vals, wv)
    D.close()


def test_empty_matrix_and_out_of_range_item():
    from hgaprec_amd.capi import Hpf, HpfError
    D = Hpf(4, 3, 2, hier=False)
    D.upload_csr(np.zeros(5, np.int64), np.zeros(0, np.uint32), np.zeros(0, np.uint8))
    colptr, users, vals = D.get_csc(0)
    assert np.array_equal(colptr, np.zeros(4, np.int64)) and users.size == 0
    with pytest.raises(HpfError, match="out of range"):
        D.upload_csr(np.array([0, 1, 1, 1, 2], np.int64), np.array([0, 3], np.uint32), np.array([1, 1], np.uint8))
    with pytest.raises(HpfError, match="monotone"):
        D.upload_csr(np.array([0, 2, 1, 1, 2], np.int64), np.array([0, 1], np.uint32), np.array([1, 1], np.uint8))
    D.close()


def test_device_upload_equals_host_upload(orc):
    """hpf_upload_csr_device + hpf_set_state_device give the same bits as the
    host-pointer calls, and both follow the oracle"""
    import torch
    from hgaprec_amd.capi import Hpf
    n, m, K = 700, 500, 20
    rowptr, col, val = make_problem(n, m, 30000, seed=11, heavy_item=True)
    M = orc.Model(n, m, K, True, True, False)
    M.set_csr(rowptr, col, val)
    M.initialize(3)
    dev = torch.device("cuda", 0)
    A = Hpf(n, m, K, hier=True, bias=True)
    A.upload_csr(rowptr, col, val)
    copy_state(M, A, True, True)
    B = Hpf(n, m, K, hier=True, bias=True)
    B.upload_csr_device(torch.from_numpy(rowptr).to(dev), torch.from_numpy(col.view(np.int32)).to(dev),
                        torch.from_numpy(val).to(dev))
    from tests.util import init_states
    for w in init_states(True, True):
        B.set_state_device(w, torch.from_numpy(np.ascontiguousarray(M.state(w))).to(dev))
    nnz = int(rowptr[-1])
    for x, y in zip(A.get_csc(nnz), B.get_csc(nnz)):
        assert np.array_equal(x, y)
    M.iterate(3); A.iterate(3); B.iterate(3)
    for w in compare_states(True, True):
        a = A.get_state(w)
        assert np.array_equal(a, B.get_state(w)), w
        assert np.array_equal(a, B.get_state_device(w, dev).cpu().numpy()), w
        assert rel_err(a, M.state(w)) < 1e-9, w
    A.close(); B.close()


@pytest.mark.parametrize("mode", ["plain", "staged", "register"])
def test_transfer_carriers_move_the_same_bytes(monkeypatch, mode):
    """HPF_H2D picks how host buffers cross PCIe (runtime-staged, the library's
    pinned double buffer, or pinning the caller's pages); sizes straddle the
    64 MiB staging buffer and the 1 MiB small-copy cut"""
    from hgaprec_amd.capi import Hpf
    monkeypatch.setenv("HPF_EXPERIMENTAL", "1")
    monkeypatch.setenv("HPF_H2D", mode)
    rng = np.random.default_rng(5)
    n, m, K = 90_000, 50, 101                   # rows are padded to the kernel shape (ld = 112), 72.7 MB per array
    D = Hpf(n, m, K, hier=True, bias=False)
    a = rng.random((n, K))
    D.set_state("THETA_E", a)
    assert np.array_equal(D.get_state("THETA_E"), a)
    x = rng.random(n)
    D.set_state("XI_E", x)
    assert np.array_equal(D.get_state("XI_E"), x)
    D.close()
    n, K = 100_000, 100                         # 80 MB
    D = Hpf(n, m, K, hier=True, bias=False)
    a = rng.random((n, K))
    D.set_state("THETA_ELOG", a)
    assert np.array_equal(D.get_state("THETA_ELOG"), a)
    D.close()
    n, K = 3000, 7                               # bias columns: single-column blocks inside padded rows
    D = Hpf(n, m, K, hier=True, bias=True)
    a, b = rng.random((n, K)), rng.random(n)
    D.set_state("THETA_E", a); D.set_state("UBIAS_E", b)
    assert np.array_equal(D.get_state("THETA_E"), a) and np.array_equal(D.get_state("UBIAS_E"), b)
    D.close()


def test_work_info_reports_the_cut(monkeypatch):
    from hgaprec_amd.capi import Hpf
    monkeypatch.setenv("HPF_EXPERIMENTAL", "1")
    monkeypatch.setenv("HPF_SEG_MAX", "16")
    monkeypatch.setenv("HPF_HUGE_SLOTS", "8")
    n, m = 400, 3000
    rowptr, col, val = make_problem(n, m, 2000, seed=31, heavy_user=True, heavy_item=True)
    D = Hpf(n, m, 12, hier=True, bias=True)
    D.upload_csr(rowptr, col, val)
    w = D.work_info()
    assert w["nnz"] == rowptr[-1]
    assert w["ld"] == w["phi_G"] * w["phi_R"] * w["phi_V"] == w["sweep_G"] * w["sweep_R"] >= 14
    assert w["user_huge_rows"] >= 1 and w["item_huge_rows"] >= 1          # m / 16 and n / 16 segments > 8
    assert w["user_long_rows"] >= w["user_huge_rows"]
    assert w["user_segments"] >= n and w["item_segments"] >= m
    D.close()


def test_stray_tuning_variables_are_ignored(monkeypatch):
    """the library reads its tuning knobs only under HPF_EXPERIMENTAL=1: a stray
    variable in a production environment changes neither the cut nor the layout"""
    from hgaprec_amd.capi import Hpf
    n, m = 400, 3000
    rowptr, col, val = make_problem(n, m, 2000, seed=31, heavy_user=True, heavy_item=True)

    def info():
        D = Hpf(n, m, 12, hier=True, bias=True)
        D.upload_csr(rowptr, col, val)
        w = D.work_info()
        D.close()
        return w
    base = info()
    monkeypatch.setenv("HPF_SEG_MAX", "16")
    monkeypatch.setenv("HPF_HUGE_SLOTS", "8")
    monkeypatch.setenv("HPF_PHI_CFG", "16,1,2")
    monkeypatch.setenv("HPF_GRAPH", "0")
    monkeypatch.setenv("HPF_TILE", "1")
    monkeypatch.setenv("HPF_TILE_BYTES", "4096")
    monkeypatch.delenv("HPF_EXPERIMENTAL", raising=False)
    assert info() == base and base["tiles_user"] == 0 and base["tiles_item"] == 0
    monkeypatch.setenv("HPF_EXPERIMENTAL", "1")
    w = info()
    assert w["user_segments"] > base["user_segments"] and w["ld"] == 32 and w["graph_replay"] == 0
    assert w["tiles_user"] > 1 and w["tiles_item"] > 1


def test_page_locked_host_buffers_take_the_direct_route():
    """hpf_host_alloc (ABI v6): a buffer from it is the DMA's own target / source -- same values as through
    the staged route, and the block outlives nothing it should not (arrays keep it alive)"""
    from hgaprec_amd import capi
    from hgaprec_amd.capi import Hpf, pinned_empty
    n, m, K = 30_000, 9_000, 100                       # THETA_*: 24 MB, above the 1 MB bar of the direct route
    rng = np.random.default_rng(4)
    D = Hpf(n, m, K, hier=True)
    for name, rows in (("THETA", n), ("BETA", m)):
        sh = 0.3 + rng.random((rows, K))
        rt = 0.3 + rng.random((rows, K))
        pin = pinned_empty((rows, K))
        pin[...] = sh
        D.set_state(f"{name}_SHAPE", pin)              # host -> device from pinned memory
        D.set_state(f"{name}_RATE", rt)                # ... and from ordinary memory
        got_pin = D.get_state(f"{name}_SHAPE", out=pinned_empty((rows, K)))
        got = D.get_state(f"{name}_SHAPE")
        assert np.array_equal(got_pin, sh) and np.array_equal(got, sh)
        assert np.array_equal(D.get_state(f"{name}_RATE", out=pinned_empty((rows, K))), rt)
    small = pinned_empty(7)                             # below the bar: the ordinary route, same result
    x = pinned_empty((n,))
    D.set_state("XI_SHAPE", np.full(n, 1.25))
    assert np.array_equal(D.get_state("XI_SHAPE", out=x), np.full(n, 1.25))
    del small
    with pytest.raises(ValueError):
        D.get_state("THETA_SHAPE", out=np.empty((n, K), np.float32))
    D.close()
    assert got_pin[0, 0] == sh[0, 0]                    # the pinned block is still there after the handle is gone
