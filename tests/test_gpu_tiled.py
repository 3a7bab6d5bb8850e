"""-m gpu: the TILED phi pass (cache blocking of the gathered rows)
against the CPU oracle and against the row-major pass.

Tiling regroups a heavy row's nonzeros tile by tile, so the order of its sum changes: the
results agree with the oracle (and with the row-major pass) to the same 1e-9 the parity
suite holds, they are bit-identical from run to run, and `hpf_config.tiling = 1` gives the
row-major bits back.  The test matrices are small, so the tile size is forced down through
the experimental knobs (HPF_EXPERIMENTAL=1); the automatic policy is exercised on one
matrix large and skewed enough to trigger it.
"""
import numpy as np
import pytest

from tests.util import compare_states, copy_state, heldout_pairs, make_problem, rel_err

pytestmark = pytest.mark.gpu

RTOL = 1e-9


def _err(w, dev, ref):
    # rel_err is element-wise.  An Elog = psi(shape) - log(rate) may pass arbitrarily close to zero
    # (200 000 of them in the largest case here), which turns the 1e-13 that a different order of
    # summation leaves in shape and rate into any relative error one likes: Elog is held to 1e-9 of
    # max(|Elog|, 1) instead -- the quantity exp() is taken of
    if w.endswith("ELOG"):
        dev, ref = np.asarray(dev, np.float64), np.asarray(ref, np.float64)
        return float(np.max(np.abs(dev - ref) / np.maximum(np.abs(ref), 1.0)))
    return rel_err(dev, ref)


@pytest.fixture
def tile_env(monkeypatch):
    def set_(**kw):
        monkeypatch.setenv("HPF_EXPERIMENTAL", "1")
        for k, v in kw.items():
            monkeypatch.setenv(k, str(v))
    return set_


def _pair(orc, n, m, K, nnz, hier, bias, binary, seed, prob_kw=None, **hpf_kw):
    from hgaprec_amd.capi import Hpf
    rowptr, col, val = make_problem(n, m, nnz, seed, **(prob_kw or {}))
    M = orc.Model(n, m, K, hier, bias, binary)
    M.set_csr(rowptr, col, np.ones_like(val) if binary else val)
    M.initialize(seed)
    D = Hpf(n, m, K, hier=hier, bias=bias, binary=binary, **hpf_kw)
    D.upload_csr(rowptr, col, None if binary else val)
    copy_state(M, D, hier, bias)
    return M, D, (rowptr, col, val)


@pytest.mark.parametrize("K,hier,bias,binary,w_storage", [
    (5, True, False, False, 0),
    (20, True, True, False, 0),
    (50, True, False, True, 0),
    (100, True, False, False, 0),     # packed rows, (8, 6)
    (100, True, True, False, 3),      # plain fp64 rows: the other pass kernel
    (100, False, False, False, 0),
    (200, True, True, False, 0),
])
@pytest.mark.parametrize("mode", ["every_row", "heavy_rows"])
def test_tiled_pass_matches_oracle(orc, tile_env, K, hier, bias, binary, w_storage, mode):
    # 8 KiB tiles: a dozen or more tiles on either side.  every_row: HPF_TILE=1 regroups all rows
    # (no row-major rest); heavy_rows: the automatic rule with a low bar, so that both kinds of
    # work sit in one launch
    if mode == "every_row":
        tile_env(HPF_TILE=1, HPF_TILE_BYTES=8192)
    else:
        tile_env(HPF_TILE=2, HPF_TILE_BYTES=8192, HPF_TILE_RUN=2, HPF_TILE_SHARE=1)
    n, m = 700, 500
    M, D, _ = _pair(orc, n, m, K, 30000, hier, bias, binary, 17 + K, prob_kw=dict(heavy_user=True, heavy_item=True, singles=True),
                    w_storage=w_storage)
    wi = D.work_info()
    assert wi["tiles_user"] > 1 and wi["tiles_item"] > 1, wi
    hu, hi, hy = heldout_pairs(n, m, 400, seed=5)
    for it in range(5):
        M.iterate(1)
        D.iterate(1)
        for w in compare_states(hier, bias):
            e = _err(w, D.get_state(w), M.state(w))
            assert e < RTOL, f"iter {it} {w}: rel err {e:.3e}"
    so = M.heldout_sum(hu, hi, hy)
    sd, cnt = D.heldout_ll(hu, hi, hy)
    assert cnt == hu.size and abs(sd - so) / hu.size < 1e-9
    eo, ed = M.elbo(), D.elbo()
    assert abs(ed - eo) <= 1e-10 * abs(eo)
    D.close()


def test_tiled_is_reproducible_and_config_switch_restores_row_major(orc, tile_env):
    from hgaprec_amd.capi import Hpf
    tile_env(HPF_TILE=1, HPF_TILE_BYTES=8192)
    n, m, K = 900, 600, 100
    rowptr, col, val = make_problem(n, m, 40000, 5, heavy_user=True, heavy_item=True)
    M = orc.Model(n, m, K, True, False, False)
    M.set_csr(rowptr, col, val); M.initialize(5)

    def run(**kw):
        D = Hpf(n, m, K, **kw)
        D.upload_csr(rowptr, col, val)
        copy_state(M, D, True, False)
        wi = D.work_info()
        D.iterate(4)
        out = {w: D.get_state(w).copy() for w in ("THETA_E", "BETA_E", "THETA_SHAPE", "BETA_SHAPE")}
        D.close()
        return wi, out

    w1, a = run()
    w2, b = run()
    assert w1["tiles_user"] > 1 and w1 == w2
    for k in a:
        assert np.array_equal(a[k], b[k]), k                  # bit for bit, run to run
    w3, c = run(tiling=1)
    assert w3["tiles_user"] == 0 and w3["tiles_item"] == 0
    w4, d = run(tiling=1)
    for k in a:
        assert np.array_equal(c[k], d[k]), k
        assert rel_err(a[k], c[k]) < 1e-11, k                 # the order of the sums is all that differs


def test_tiled_rows_with_many_segments_are_combined_in_two_levels(orc, tile_env):
    # a blockbuster item and a user who rated everything: with 4 KiB tiles and a segment cap of 16
    # their rows are cut into hundreds of segments -- more than HPF_HUGE_SLOTS, so the two-level
    # combine of the tiled lists runs; empty rows (no segment at all) must come out as the prior
    tile_env(HPF_TILE=1, HPF_TILE_BYTES=4096, HPF_SEG_MAX=16, HPF_HUGE_SLOTS=8)
    n, m, K = 1200, 800, 20
    M, D, _ = _pair(orc, n, m, K, 25000, True, True, False, 9, prob_kw=dict(heavy_user=True, heavy_item=True, singles=True))
    wi = D.work_info()
    assert wi["tiles_user"] > 1 and wi["user_huge_rows"] >= 1 and wi["item_huge_rows"] >= 1, wi
    for it in range(4):
        M.iterate(1)
        D.iterate(1)
    for w in compare_states(True, True):
        assert _err(w, D.get_state(w), M.state(w)) < RTOL, w
    D.close()


def test_tiled_with_empty_rows_and_empty_tiles(orc, tile_env):
    # users 300..599 and items 200..399 have no rating at all: whole tiles of the gathered side are
    # never touched, and the owner rows without a segment are zeroed by the combine
    from hgaprec_amd.capi import Hpf
    tile_env(HPF_TILE=1, HPF_TILE_BYTES=8192)
    n, m, K = 600, 400, 50
    rowptr0, col0, val0 = make_problem(300, 200, 9000, 3)
    rowptr = np.concatenate([rowptr0, np.full(300, rowptr0[-1], dtype=rowptr0.dtype)])
    M = orc.Model(n, m, K, True, False, False)
    M.set_csr(rowptr, col0, val0); M.initialize(3)
    D = Hpf(n, m, K)
    D.upload_csr(rowptr, col0, val0)
    copy_state(M, D, True, False)
    assert D.work_info()["tiles_item"] > 1
    M.iterate(3); D.iterate(3)
    for w in compare_states(True, False):
        assert _err(w, D.get_state(w), M.state(w)) < RTOL, w
    D.close()


def test_automatic_policy_tiles_a_skewed_side_only(orc):
    """No knobs: 4 MiB tiles.  60 000 users x 768 B = 44 MiB of gathered rows for the item pass and a
    handful of items holding most ratings -> the item side is tiled; the user pass gathers 2 000
    item rows (1.5 MiB: they fit an L2 as they are) -> row-major."""
    from hgaprec_amd.capi import Hpf
    from hgaprec_amd import synth
    n, m, K, nnz = 60000, 2000, 100, 4_000_000
    rowptr, col, val = synth.generate(n, m, nnz, alpha_u=0.3, alpha_i=1.2, seed=4, device="cuda")
    M = orc.Model(n, m, K, True, False, False)
    M.set_csr(rowptr, col, val); M.initialize(4)
    D = Hpf(n, m, K)
    D.upload_csr(rowptr, col, val)
    copy_state(M, D, True, False)
    wi = D.work_info()
    assert wi["tiles_item"] >= 8 and wi["tiles_user"] == 0, wi
    assert wi["tile_chunk_item"] == 2 and wi["tile_chunk_user"] == 0, wi       # one wave per workgroup, two segments per chunk (ABI v8 says so)
    M.iterate(2); D.iterate(2)
    for w in compare_states(True, False):
        assert _err(w, D.get_state(w), M.state(w)) < RTOL, w
    ms = D.gather_only_ms(1, 2)                      # the probe walks the same chunks
    assert ms > 0
    D.close()


@pytest.mark.parametrize("K", [5, 100])
def test_tiled_list_longer_than_a_launch_holds(tile_env, K):
    """A launch holds fewer than 2^32 work-items, i.e. 2^20 workgroups of 256: with 1 KiB tiles this
    matrix is cut into ~10^7 segments -- more than 2^20 chunks of eight -- so the chunks must grow
    instead of the grid.  (A grid past the limit is silently cut short: found with whole-suite runs
    under forced tiling.)  The phi sums of a row add up to its ratings whatever the state is."""
    import torch
    from tests.test_gpu_fullsize import _device_model, _row_mass
    from hgaprec_amd import synth
    tile_env(HPF_TILE=1, HPF_TILE_BYTES=1024 if K == 5 else 4096)       # K = 100: 768-byte rows, five to a tile (at most 65 534 tiles)
    n, m, nnz = 200_000, 20_000, 20_000_000
    dev = torch.device("cuda", 0)
    rowptr, col, val = synth.generate_device(n, m, nnz, 0.4, 0.7, seed=5, device=dev, binary=True)
    D = _device_model(dict(m=m, K=K, binary=True), n, rowptr, col, None, 0, n)
    wi = D.work_info()
    assert wi["tiles_user"] > 100 and wi["tiles_item"] > 1000 and wi["item_segments"] > 8 * (1 << 20), wi
    if K == 5:
        assert wi["w_layout"] == 0 and wi["tile_chunk_item"] > 8, wi          # four-wave workgroups: 2^20 of them, chunks of eight grew
    else:
        # packed rows: one-wave workgroups, of which a launch holds 2^22 (capped at 2^20 like the four-wave groups,
        # a list this long got chunks of 32 and more instead of two; the cap now scales with the workgroup size)
        assert wi["w_layout"] == 3 and 2 < wi["tile_chunk_item"] <= 16, wi
        assert wi["item_segments"] / wi["tile_chunk_item"] <= (1 << 22), wi
    D.iterate(2)
    ts, bs = D.get_state_device("THETA_SHAPE", dev), D.get_state_device("BETA_SHAPE", dev)
    deg_u = _row_mass(rowptr, None)
    assert float((((ts - 0.3).sum(1) - deg_u).abs() / deg_u.clamp(min=1.0)).max()) < 1e-11
    deg_i = torch.bincount(col.to(torch.int64), minlength=m).to(torch.float64)
    assert float((((bs - 0.3).sum(1) - deg_i).abs() / deg_i.clamp(min=1.0)).max()) < 1e-11
    D.close()


@pytest.mark.parametrize("where", ["before_the_first_pass", "in_a_sweep"])
def test_fallback_from_packed_rows_cuts_the_tiled_lists_again(orc, tile_env, where):
    """A tile holds a fixed number of BYTES of gathered rows: when the rows move from the packed form to plain
    doubles (a state p59 cannot hold) the tiled lists are cut again for the longer rows, and
    from there on the handle IS one that held plain doubles from the start: same layout, same lists, same kernels.
    before_the_first_pass: the start state itself does not fit (an Elog spread of 120) -- every iteration runs on the
    new lists and the run equals the w_storage = 3 run bit for bit.  in_a_sweep: the first user sweep cannot pack its
    rows -- the first iteration's passes ran on the packed rows' lists, so the two runs differ by the order of that
    iteration's sums (1e-13), and by nothing else."""
    from hgaprec_amd.capi import Hpf
    tile_env(HPF_TILE=2, HPF_TILE_BYTES=8192, HPF_TILE_RUN=2, HPF_TILE_SHARE=1)
    n, m, K = 700, 500, 100
    rowptr, col, val = make_problem(n, m, 30000, 23, heavy_user=True, heavy_item=True)
    M = orc.Model(n, m, K, True, False, False)
    M.set_csr(rowptr, col, val); M.initialize(23)
    if where == "in_a_sweep":
        be = M.state("BETA_E").copy()
        be[:, 0] = 1e40                                   # the first user sweep's W is e^-97 of its row maximum in column 0
        M.set_state("BETA_E", be)
    else:
        el = M.state("THETA_ELOG").copy()
        el[:, 0] -= 120.0
        M.set_state("THETA_ELOG", el)
    runs = {}
    for ws in (0, 3):
        D = Hpf(n, m, K, hier=True, w_storage=ws)
        D.upload_csr(rowptr, col, val)
        wi0 = D.work_info()
        copy_state(M, D, True, False)
        D.iterate(3)
        runs[ws] = ({w: D.get_state(w) for w in compare_states(True, False)}, wi0, D.work_info())
        D.close()
    M.iterate(3)
    a0, a1 = runs[0][1], runs[0][2]
    assert a0["w_layout"] == 3 and a1["w_layout"] == 4 and a1["w_fallbacks"] == 1
    assert a0["tiles_item"] > 1 and a1["tiles_item"] > a0["tiles_item"]                # fewer of the longer rows per tile
    assert (a1["tiles_user"], a1["tiles_item"], a1["tile_rows_item"]) == tuple(runs[3][2][k] for k in ("tiles_user", "tiles_item", "tile_rows_item"))
    for w in compare_states(True, False):
        if where == "before_the_first_pass":
            assert np.array_equal(runs[0][0][w], runs[3][0][w]), w
        else:
            assert _err(w, runs[0][0][w], runs[3][0][w]) < 1e-11, w
        assert _err(w, runs[0][0][w], M.state(w)) < RTOL, w


@pytest.mark.parametrize("K,bias", [(100, False), (200, True), (50, False)])
def test_workgroup_size_and_chunking_do_not_change_a_bit(monkeypatch, K, bias):
    """Round 5: a tiled side runs one wave per workgroup with chunks of two segments (hpf_handle::phi_wg; until round 4 four
    waves shared eight segments).  Which workgroup a segment lands in decides WHEN it runs, never what it adds up: the
    segments, the order of the nonzeros inside each and the partial slots are the same, so every state array must be
    bit-identical between the default, round 4's launch (HPF_PHI_WG=256, HPF_TILE_CHUNK=8) and two other cuts."""
    from hgaprec_amd.capi import Hpf
    from tests.util import init_states
    from oracle import orc
    n, m = 900, 600
    rowptr, col, val = make_problem(n, m, 40000, 31 + K, heavy_user=True, heavy_item=True, singles=True)
    Mo = orc.Model(n, m, K, True, bias, False)
    Mo.set_csr(rowptr, col, val)
    Mo.initialize(3)
    init = {w: Mo.state(w).copy() for w in init_states(True, bias)}
    base = dict(HPF_EXPERIMENTAL="1", HPF_TILE="2", HPF_TILE_BYTES="8192", HPF_TILE_RUN="2", HPF_TILE_SHARE="1", HPF_GRAPH="0")
    results = []
    for extra in ({}, {"HPF_PHI_WG": "256", "HPF_TILE_CHUNK": "8"}, {"HPF_PHI_WG": "128", "HPF_TILE_CHUNK": "3"},
                  {"HPF_PHI_WG": "64", "HPF_TILE_CHUNK": "1"}):
        for k in ("HPF_PHI_WG", "HPF_TILE_CHUNK"):
            monkeypatch.delenv(k, raising=False)
        for k, v in {**base, **extra}.items():
            monkeypatch.setenv(k, v)
        D = Hpf(n, m, K, hier=True, bias=bias)
        D.upload_csr(rowptr, col, val)
        for w, a in init.items():
            D.set_state(w, a)
        wi = D.work_info()
        assert wi["tiles_user"] > 1 and wi["tiles_item"] > 1 and wi["w_layout"] == 3, wi
        D.iterate(4)
        results.append({w: D.get_state(w).copy() for w in compare_states(True, bias)})
        D.close()
    for other in results[1:]:
        for w in results[0]:
            assert np.array_equal(results[0][w], other[w]), w
