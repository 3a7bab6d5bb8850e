"""-m gpu: the HIP path (through the C-ABI) against the CPU oracle on the same
seeded inputs.  Tolerances (north_star: factors within 1e-4 rel-err of the CPU
reference): the device differs from the oracle only by summation order and by
last-ulp differences of exp/log/digamma, so after a handful of sweeps we hold
it to 1e-9 relative -- five orders tighter than the 1e-4 contract -- and the
held-out log-likelihood to 1e-9 absolute per pair.
"""
import numpy as np
import pytest

from tests.util import (compare_states, copy_state, heldout_pairs, init_states, make_problem, rel_err)

pytestmark = pytest.mark.gpu

RTOL = 1e-9


def _run_pair(orc, n, m, K, nnz, hier, bias, binary, iters, seed, prob_kw=None, val_mode="ratings",
              w_storage=0):
    from hgaprec_amd.capi import Hpf
    rowptr, col, val = make_problem(n, m, nnz, seed, **(prob_kw or {}))
    if val_mode == "wrap0":            # ratings that wrapped to 0 in the uint8 store
        val = val.copy()
        val[::7] = 0
    if binary:
        val_o, val_d = np.ones_like(val), None
    else:
        val_o, val_d = val, val
    M = orc.Model(n, m, K, hier, bias, binary)
    M.set_csr(rowptr, col, val_o)
    M.initialize(seed)
    D = Hpf(n, m, K, hier=hier, bias=bias, binary=binary, w_storage=w_storage)
    D.upload_csr(rowptr, col, val_d)
    copy_state(M, D, hier, bias)
    return M, D


@pytest.mark.parametrize("K,hier,bias,binary", [
    (5, True, False, False),
    (5, True, True, False),
    (5, True, False, True),
    (5, False, False, False),
    (5, False, True, False),
    (20, True, False, False),
    (21, True, True, False),     # odd K + bias: padded stride
    (50, True, False, True),
    (100, True, False, False),   # K > 64 lanes
    (100, True, True, False),
    (200, True, True, False),
    (7, True, False, False),
])
def test_iterations_match_oracle(orc, K, hier, bias, binary):
    n, m = 300, 200
    M, D = _run_pair(orc, n, m, K, 6000, hier, bias, binary, 6, seed=11 + K)
    hu, hi, hy = heldout_pairs(n, m, 500, seed=5)
    for it in range(6):
        M.iterate(1)
        D.iterate(1)
        for w in compare_states(hier, bias):
            e = rel_err(D.get_state(w), M.state(w))
            assert e < RTOL, f"iter {it} {w}: rel err {e:.3e}"
        so = M.heldout_sum(hu, hi, hy)
        sd, cnt = D.heldout_ll(hu, hi, hy)
        assert cnt == hu.size
        assert abs(sd - so) / hu.size < 1e-9, (it, sd, so)
        eo, ed = M.elbo(), D.elbo()                  # HGAPRec::logl
        assert abs(ed - eo) <= 1e-10 * abs(eo), (it, ed, eo)


def _random_case(j):
    """case j of a fixed pseudo-random series: shape, flags, row layout and work-list knobs drawn together"""
    rng = np.random.default_rng(9000 + j)
    K = int(rng.choice([1, 2, 3, 4, 6, 9, 16, 17, 31, 32, 33, 48, 51, 63, 64, 65, 99, 100, 101, 127, 129, 160, 199, 255, 257, 320]))
    hier = bool(rng.integers(2))
    bias = bool(rng.integers(2))
    binary = bool(rng.integers(4) == 0)
    n = int(rng.choice([1, 2, 17, 64, 65, 150, 333]))
    m = int(rng.choice([1, 2, 19, 64, 129, 250]))
    nnz = int(max(n, m) * rng.choice([1, 3, 12, 40]))
    env = {}
    knob = int(rng.integers(5))
    if knob == 1:
        env = dict(HPF_TILE="1", HPF_TILE_BYTES=str(int(rng.choice([2048, 8192, 30000]))))
    elif knob == 2:
        env = dict(HPF_TILE="2", HPF_TILE_BYTES="4096", HPF_TILE_RUN=str(int(rng.choice([1, 2, 5]))), HPF_TILE_SHARE="1")
    elif knob == 3:
        env = dict(HPF_SEG_MAX=str(int(rng.choice([16, 20, 64]))), HPF_HUGE_SLOTS=str(int(rng.choice([2, 4, 9]))))
    elif knob == 4:
        env = dict(HPF_GRAPH=str(int(rng.integers(2))), HPF_SWEEP_BLOCKS=str(int(rng.choice([1, 3, 64]))))
    kw = dict(heavy_user=bool(rng.integers(2)), heavy_item=bool(rng.integers(2)), singles=bool(rng.integers(2)))
    return dict(K=K, hier=hier, bias=bias, binary=binary, n=n, m=m, nnz=nnz, env=env, prob_kw=kw,
                w_storage=int(rng.choice([0, 0, 3])), val_mode=str(rng.choice(["ratings", "ratings", "wrap0"])),
                iters=int(rng.integers(1, 5)), chunk=int(rng.integers(1, 4)))


@pytest.mark.parametrize("j", range(40))
def test_random_configurations_match_the_oracle(orc, monkeypatch, j):
    """a differential sweep over combinations the parametrized tests do not enumerate: 1 .. 320 factors with and
    without bias / hier / binary data, one-row and one-column matrices, rows of one rating and rows that hold everything,
    ratings wrapped to 0, packed and plain rows, tiled and row-major lists with tiny tiles, short segments with the
    two-level combine, hipGraph replay, several iterations per call"""
    c = _random_case(j)
    monkeypatch.setenv("HPF_EXPERIMENTAL", "1")
    for k, v in c["env"].items():
        monkeypatch.setenv(k, v)
    n, m = c["n"], c["m"]
    M, D = _run_pair(orc, n, m, c["K"], c["nnz"], c["hier"], c["bias"], c["binary"], c["iters"], seed=300 + j,
                     prob_kw=c["prob_kw"], val_mode=c["val_mode"], w_storage=c["w_storage"])
    done = 0
    while done < c["iters"]:
        step = min(c["chunk"], c["iters"] - done)
        M.iterate(step)
        D.iterate(step)
        done += step
        for w in compare_states(c["hier"], c["bias"]):
            dv, rf = np.asarray(D.get_state(w), np.float64), np.asarray(M.state(w), np.float64)
            # element-wise relative error, except Elog = psi(shape) - log(rate), which may pass arbitrarily close to
            # zero (case 25: 1.9e-9 of an entry of 1e-3): held to 1e-9 of max(|Elog|, 1), the quantity exp() is taken of
            e = float(np.max(np.abs(dv - rf) / np.maximum(np.abs(rf), 1.0))) if w.endswith("ELOG") else rel_err(dv, rf)
            assert e < RTOL, f"case {j} {c}: after {done} iterations {w}: rel err {e:.3e}"
    hu, hi, hy = heldout_pairs(n, m, min(200, n * m), seed=j)
    so = M.heldout_sum(hu, hi, hy)
    sd, cnt = D.heldout_ll(hu, hi, hy)
    assert cnt == hu.size and abs(sd - so) <= 1e-9 * max(1, hu.size), (j, c, sd, so)
    eo, ed = M.elbo(), D.elbo()
    assert abs(ed - eo) <= 1e-10 * abs(eo), (j, c, ed, eo)
    D.close()


def test_vb_bias_novb_uses_the_previous_iterations_sums(orc):
    """-bias -novb without -hier: vb_bias()'s else-branch (hgaprec.cc:1276-1297).  Both rates
    come from the expectations of the previous iteration -- the item rate takes
    sum_u E[theta] of BEFORE the user update -- then everything is swapped."""
    from hgaprec_amd.capi import Hpf
    n, m, K = 300, 200, 7
    rowptr, col, val = make_problem(n, m, 6000, 21)
    M = orc.Model(n, m, K, False, True, False, novb=True)
    M.set_csr(rowptr, col, val)
    M.initialize(21)
    M0 = orc.Model(n, m, K, False, True, False)          # the default (Gauss-Seidel) order, for contrast
    M0.set_csr(rowptr, col, val)
    M0.initialize(21)
    D = Hpf(n, m, K, hier=False, bias=True, novb=True)
    D.upload_csr(rowptr, col, val)
    copy_state(M, D, False, True)
    theta_e0 = M.state("THETA_E").copy()
    for it in range(6):
        M.iterate(1); M0.iterate(1); D.iterate(1)
        for w in compare_states(False, True):
            e = rel_err(D.get_state(w), M.state(w))
            assert e < RTOL, f"iter {it} {w}: rel err {e:.3e}"
        if it == 0:      # the first item rate is 0.3 + sum_u E[theta] of the START state
            assert np.allclose(D.get_state("BETA_RATE"), 0.3 + theta_e0.sum(0), rtol=1e-12)
    assert rel_err(M.state("BETA_E"), M0.state("BETA_E")) > 1e-3          # the order matters
    D.close()
    # with -hier the reference never reads the flag: identical bits with and without it
    outs = []
    for novb in (False, True):
        D = Hpf(n, m, K, hier=True, bias=True, novb=novb)
        D.upload_csr(rowptr, col, val)
        Mh = orc.Model(n, m, K, True, True, False, novb=novb)
        Mh.set_csr(rowptr, col, val); Mh.initialize(3)
        copy_state(Mh, D, True, True)
        D.iterate(3); Mh.iterate(3)
        outs.append((D.get_state("BETA_E"), Mh.state("BETA_E").copy()))
        D.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_vb_bias_novb_across_two_ranks(orc):
    """the same Jacobi order with the users sharded over two ranks (round 4): the start state's
    sum_u E[theta] -- what the FIRST item rate is built from -- is summed over the ranks once
    (hpf_start_sums leaves each rank's part in the tail of its exchange buffer); iterating
    without that is refused, not run on a partial sum."""
    from hgaprec_amd.capi import Hpf, HpfError
    from hgaprec_amd import dist as hd
    n, m, K = 300, 200, 7
    rowptr, col, val = make_problem(n, m, 6000, 21)
    M = orc.Model(n, m, K, False, True, False, novb=True)
    M.set_csr(rowptr, col, val)
    M.initialize(21)
    init = {w: M.state(w).copy() for w in init_states(False, True)}
    parts = hd.partition_users(rowptr, 2)
    shards = []
    for r, (a, b) in enumerate(parts):
        D = Hpf(b - a, m, K, hier=False, bias=True, novb=True, n_ranks=2, rank=r, n_users_total=n)
        D.upload_csr(*hd.shard_csr(rowptr, col, val, a, b))
        hd.scatter_state(D, init, a, b, hier=False)
        shards.append(D)
    with pytest.raises(HpfError, match="hpf_start_sums"):
        shards[0].iterate_local()
    ld = shards[0].work_info()["ld"]
    assert all(D.work_info()["start_sums_pending"] == 1 for D in shards)
    for D in shards:
        D.start_sums()
    # ABI v8: the field keeps reading 1 while the tail holds this rank's PART -- a caller may look before or after
    # hpf_start_sums -- until the first pass of the next iteration
    assert all(D.work_info()["start_sums_pending"] == 1 for D in shards)
    bufs = [D.exchange_read() for D in shards]
    tail = bufs[0][-ld:] + bufs[1][-ld:]
    assert np.allclose(tail[:K], init["THETA_E"].sum(0), rtol=1e-13)
    for D, x in zip(shards, bufs):
        x[-ld:] = tail
        D.exchange_write(x)
    for it in range(5):
        M.iterate(1)
        for D in shards:
            D.iterate_local()
            assert D.work_info()["start_sums_pending"] == 0
        bufs = [D.exchange_read() for D in shards]
        tot = bufs[0] + bufs[1]
        for D in shards:
            D.exchange_write(tot)
            D.iterate_global()
        for (a, b), D in zip(parts, shards):
            for w in compare_states(False, True):
                want = M.state(w)
                if w.startswith(("THETA_", "UBIAS_")) and w != "THETA_RATE":
                    want = want[a:b]
                e = rel_err(D.get_state(w), want)
                assert e < RTOL, f"iter {it} rank rows [{a},{b}) {w}: rel err {e:.3e}"
    for D in shards:
        D.close()


def test_power_law_long_rows_and_singletons(orc):
    # one user holding every item, one item rated by every user, users with a
    # single rating; rows longer than the segment cap (512) on both sides
    n, m, K = 1500, 900, 20
    M, D = _run_pair(orc, n, m, K, 30000, True, False, False, 4, seed=3,
                     prob_kw=dict(heavy_user=True, heavy_item=True, singles=True))
    for it in range(4):
        M.iterate(1)
        D.iterate(1)
    for w in compare_states(True, False):
        assert rel_err(D.get_state(w), M.state(w)) < RTOL, w


@pytest.mark.parametrize("huge_slots", [8, 256])
def test_one_user_holds_most_ratings(orc, monkeypatch, huge_slots):
    # one user with > 50 % of all nonzeros (every item), one item
    # rated by every user.  Its row is cut into hundreds of segments whose
    # partial sums are combined in two levels (groups of partials, then the
    # group sums) once it has more than HPF_HUGE_SLOTS segments.
    monkeypatch.setenv("HPF_EXPERIMENTAL", "1")
    monkeypatch.setenv("HPF_SEG_MAX", "16")
    monkeypatch.setenv("HPF_HUGE_SLOTS", str(huge_slots))
    n, m, K = 400, 3000, 12
    M, D = _run_pair(orc, n, m, K, 2000, True, True, False, 4, seed=31,
                     prob_kw=dict(heavy_user=True, heavy_item=True))
    for it in range(4):
        M.iterate(1)
        D.iterate(1)
    for w in compare_states(True, True):
        assert rel_err(D.get_state(w), M.state(w)) < RTOL, w
    hu, hi, hy = heldout_pairs(n, m, 300, seed=2)
    so = M.heldout_sum(hu, hi, hy)
    sd, cnt = D.heldout_ll(hu, hi, hy)
    assert cnt == hu.size and abs(sd - so) / hu.size < 1e-9


def test_rating_wrapped_to_zero_is_unscaled(orc):
    # rating 256 is stored as uint8 0 by the reference; "if (y > 1) scale" then
    # leaves phi unscaled (hgaprec.cc:1355)
    M, D = _run_pair(orc, 200, 150, 10, 4000, True, False, False, 3, seed=9, val_mode="wrap0")
    M.iterate(3)
    D.iterate(3)
    for w in ("THETA_E", "BETA_E", "THETA_SHAPE", "BETA_SHAPE"):
        assert rel_err(D.get_state(w), M.state(w)) < RTOL, w


def test_duplicate_pairs_count_twice(orc):
    # a duplicated (u,i) line appears twice in the user's item list
    from hgaprec_amd.capi import Hpf
    n, m, K = 60, 40, 8
    rowptr, col, val = make_problem(n, m, 900, 21)
    # duplicate the first entry of every row
    rows = [np.concatenate([col[rowptr[u]:rowptr[u + 1]], col[rowptr[u]:rowptr[u] + 1]]) for u in range(n)]
    vals = [np.concatenate([val[rowptr[u]:rowptr[u + 1]], val[rowptr[u]:rowptr[u] + 1]]) for u in range(n)]
    rp = np.zeros(n + 1, np.int64)
    rp[1:] = np.cumsum([len(r) for r in rows])
    c, v = np.concatenate(rows).astype(np.uint32), np.concatenate(vals).astype(np.uint8)
    M = orc.Model(n, m, K, True, False, False)
    M.set_csr(rp, c, v)
    M.initialize(2)
    D = Hpf(n, m, K)
    D.upload_csr(rp, c, v)
    copy_state(M, D, True, False)
    M.iterate(3)
    D.iterate(3)
    assert rel_err(D.get_state("BETA_E"), M.state("BETA_E")) < RTOL
    assert rel_err(D.get_state("THETA_E"), M.state("THETA_E")) < RTOL


def test_run_to_run_bit_reproducible(orc):
    # no atomics anywhere on the path: two runs give identical bits
    outs = []
    for _ in range(2):
        M, D = _run_pair(orc, 400, 300, 20, 9000, True, True, False, 5, seed=4)
        D.iterate(5)
        outs.append((D.get_state("THETA_E"), D.get_state("BETA_E"), D.get_state("XI_E")))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("bias", [False, True])
def test_graph_replay_equals_eager_launches(orc, monkeypatch, bias):
    # hpf_iterate replays one captured iteration (hipGraph) when the problem is
    # launch-bound; same kernels, same order => identical bits, and get/set
    # between calls still works.  Replayed iterations report only their total.
    outs, tms = [], []
    monkeypatch.setenv("HPF_EXPERIMENTAL", "1")
    for mode in ("0", "1"):
        monkeypatch.setenv("HPF_GRAPH", mode)
        M, D = _run_pair(orc, 400, 300, 20, 9000, True, bias, False, 5, seed=4)
        D.iterate(2)
        mid = D.get_state("THETA_ELOG")
        D.set_state("THETA_ELOG", mid)             # W re-derived outside the graph
        D.iterate(3)
        D.iterate(1)
        tms.append(D.mean_timing(3))
        outs.append((D.get_state("THETA_E"), D.get_state("BETA_E"), D.get_state("XI_E"),
                     D.get_state("BETA_ELOG"), D.get_state("THETA_RATE")))
        if mode == "1":
            M.iterate(6)
            assert rel_err(outs[-1][1], M.state("BETA_E")) < RTOL
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    assert tms[0]["phi_user_ms"] > 0 and tms[0]["iteration_ms"] > 0
    assert tms[1]["phi_user_ms"] == 0 and tms[1]["iteration_ms"] > 0
    assert tms[0]["iterations"] == tms[1]["iterations"] == 6


@pytest.mark.parametrize("hier,bias,novb", [(True, True, False), (True, False, False), (False, True, True)])
def test_a_rank_of_several_replays_its_three_pieces_bit_identically(orc, monkeypatch, hier, bias, novb):
    """Round 6: with n_ranks > 1 the iteration is cut by its collectives, so it is captured as THREE graphs -- item pass | user
    pass + user sweep | item sweep -- replayed by hpf_iterate_local_items / _users / hpf_iterate_global.  Same kernels in the
    same order: the bits of the eager pieces, through a get/set in between (graphs dropped and captured again), with the -novb
    copy of the old column sums inside the user piece, and through hpf_iterate on a (one-rank) communicator of the library's."""
    from hgaprec_amd.capi import Hpf
    n, m, K = 400, 300, 20
    rowptr, col, val = make_problem(n, m, 9000, 4)
    monkeypatch.setenv("HPF_EXPERIMENTAL", "1")
    outs, infos = [], []
    for mode, comm in (("0", False), ("1", False), ("1", True)):
        monkeypatch.setenv("HPF_GRAPH", mode)
        M = orc.Model(n, m, K, hier, bias, False)
        M.set_csr(rowptr, col, val); M.initialize(4)
        D = Hpf(n, m, K, hier=hier, bias=bias, novb=novb, n_ranks=1 if comm else 2, rank=0, n_users_total=n)
        D.upload_csr(rowptr, col, val)
        copy_state(M, D, hier, bias)
        if comm:
            D.comm_init(Hpf.comm_unique_id())
        elif novb:
            D.start_sums()                          # one "rank" of two holding every user: its part IS the sum

        def it(k):
            if comm:
                D.iterate(k)
                return
            for _ in range(k):
                D.iterate_local_items(); D.iterate_local_users(); D.iterate_global()
        it(2)
        mid = D.get_state("BETA_E")                 # a synchronising export between replays
        it(2)
        D.iterate_local_phi(); D.iterate_local_sweep(); D.iterate_global()      # the other cut stays eager, in any mode
        it(1)
        infos.append((D.work_info()["graph_replay"], D.mean_timing(1)))
        outs.append([D.get_state(w) for w in ("THETA_E", "BETA_E", "THETA_SHAPE", "BETA_SHAPE")] + [mid])
        D.close()
    for a, b, c in zip(*outs):
        assert np.array_equal(a, b) and np.array_equal(a, c)
    assert [i[0] for i in infos] == [0, 2, 2]
    # replayed pieces: the events inside a piece coincide (a few microseconds apart), the piece's time sits in its first field
    assert infos[0][1]["sweep_user_ms"] > 0 and infos[1][1]["sweep_user_ms"] < 0.02 and infos[1][1]["phi_user_ms"] > 0
    assert infos[1][1]["phi_item_ms"] > 0 and infos[1][1]["sweep_item_ms"] > 0 and infos[1][1]["iteration_ms"] > 0


@pytest.mark.parametrize("hier,bias", [(True, True), (False, False)])
def test_snapshot_restore_continues_bit_identically(orc, hier, bias):
    # hpf_snapshot_save / _load: the loop's device arrays verbatim.  A second handle that
    # loads the blob after 3 iterations must stay bit-identical to the uninterrupted
    # one -- state, held-out likelihood and ELBO -- which hpf_set_state cannot give
    # (it re-derives W from Elog and re-sums the expectations).
    from hgaprec_amd.capi import Hpf, HpfError
    n, m, K = 400, 300, 20
    M, A = _run_pair(orc, n, m, K, 9000, hier, bias, False, 8, seed=14)
    A.iterate(3)
    blob = A.snapshot()
    B = Hpf(n, m, K, hier=hier, bias=bias)
    rowptr, col, val = make_problem(n, m, 9000, 14)
    B.upload_csr(rowptr, col, val)
    B.restore(blob)
    A.iterate(5); B.iterate(5)
    for w in compare_states(hier, bias):
        assert np.array_equal(A.get_state(w), B.get_state(w)), w
    hu, hi, hy = heldout_pairs(n, m, 500, seed=2)
    assert A.heldout_ll(hu, hi, hy) == B.heldout_ll(hu, hi, hy)
    assert A.elbo() == B.elbo()
    M.iterate(8)
    assert rel_err(B.get_state("BETA_E"), M.state("BETA_E")) < RTOL
    # a blob of another shape, a truncated blob and a damaged header are refused
    C2 = Hpf(n, m, K + 1, hier=hier, bias=bias)
    with pytest.raises(HpfError):
        C2.restore(blob)
    with pytest.raises(HpfError):
        B.restore(blob[:-8])
    bad = blob.copy(); bad[0] ^= 1
    with pytest.raises(HpfError):
        B.restore(bad)
    # nor a snapshot of another job of the same dimensions: other priors, other
    # ratings, another rank's shard -- and a refused blob leaves the handle as it was
    before = B.get_state("THETA_E")
    for kw, other_data in ((dict(s_prior=0.4), False), (dict(n_ranks=2, rank=1, n_users_total=2 * n), False), ({}, True)):
        F = Hpf(n, m, K, hier=hier, bias=bias, **kw)
        rp2, c2, v2 = make_problem(n, m, 8000 if other_data else 9000, 15 if other_data else 14)
        F.upload_csr(rp2, c2, v2)
        with pytest.raises(HpfError):
            F.restore(blob)
        F.close()
    assert np.array_equal(B.get_state("THETA_E"), before)
    # before the first iteration too: the start state itself round-trips
    M2, D = _run_pair(orc, n, m, K, 9000, hier, bias, False, 2, seed=14)
    E = Hpf(n, m, K, hier=hier, bias=bias)
    E.upload_csr(rowptr, col, val)
    E.restore(D.snapshot())
    D.iterate(2); E.iterate(2)
    assert np.array_equal(D.get_state("THETA_E"), E.get_state("THETA_E"))
    assert np.array_equal(D.get_state("THETA_RATE"), E.get_state("THETA_RATE"))


def test_twenty_iterations_within_contract(orc):
    # the north_star contract itself: 1e-4 relative on the factors
    M, D = _run_pair(orc, 500, 400, 20, 15000, True, False, False, 20, seed=8)
    M.iterate(20)
    D.iterate(20)
    for w in ("THETA_E", "BETA_E"):
        assert rel_err(D.get_state(w), M.state(w)) < 1e-4, w
    # and far inside it in practice
    assert rel_err(D.get_state("THETA_E"), M.state("THETA_E")) < 1e-7


@pytest.mark.parametrize("hier,bias", [(True, False), (True, True), (False, False)])
def test_three_hundred_iterations_within_contract(orc, hier, bias):
    # north_star: "factors within 1e-4 rel-err of CPU reference".  The CAVI map amplifies
    # rounding differences from sweep to sweep, so the margin is checked where a real run
    # ends: 300 iterations (the reference's stop rule typically fires between 40 and 200).
    M, D = _run_pair(orc, 600, 400, 20, 20000, hier, bias, False, 300, seed=21)
    M.iterate(300)
    D.iterate(300)
    worst = max(rel_err(D.get_state(w), M.state(w)) for w in ("THETA_E", "BETA_E", "THETA_SHAPE", "BETA_SHAPE"))
    assert worst < 1e-4, worst
    assert worst < 1e-6, worst            # what the fp64 path actually holds (the f32-storage mode does not)
    hu, hi, hy = heldout_pairs(600, 400, 2000, seed=2)
    assert abs(D.heldout_ll(hu, hi, hy)[0] - M.heldout_sum(hu, hi, hy)) / hu.size < 1e-9


def test_two_logical_ranks_equal_one(orc):
    # user sharding with a host-side sum standing in for the all-reduce:
    # iterate_local -> sum exchange buffers -> iterate_global on both shards
    import ctypes as C
    from hgaprec_amd.capi import Hpf
    n, m, K = 400, 250, 20
    rowptr, col, val = make_problem(n, m, 8000, 17)
    M = orc.Model(n, m, K, True, True, False)
    M.set_csr(rowptr, col, val)
    M.initialize(5)
    cut = int(np.searchsorted(rowptr, rowptr[-1] // 2))
    shards = []
    for r, (a, b) in enumerate([(0, cut), (cut, n)]):
        D = Hpf(b - a, m, K, hier=True, bias=True, n_ranks=2, rank=r, n_users_total=n)
        rp = rowptr[a:b + 1] - rowptr[a]
        D.upload_csr(rp, col[rowptr[a]:rowptr[b]], val[rowptr[a]:rowptr[b]])
        for w in ("THETA_SHAPE", "THETA_E", "THETA_ELOG", "XI_SHAPE", "XI_RATE", "XI_E", "XI_ELOG",
                  "UBIAS_SHAPE", "UBIAS_E", "UBIAS_ELOG"):
            D.set_state(w, M.state(w)[a:b])
        for w in ("BETA_SHAPE", "BETA_E", "BETA_ELOG", "ETA_SHAPE", "ETA_RATE", "ETA_E", "ETA_ELOG",
                  "IBIAS_SHAPE", "IBIAS_E", "IBIAS_ELOG"):
            D.set_state(w, M.state(w))
        shards.append((a, b, D))
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    for it in range(4):
        M.iterate(1)
        bufs = []
        for _, _, D in shards:
            D.iterate_local()
            D.synchronize()
            p, cnt = D.exchange_buffer()
            h = np.empty(cnt, np.float64)
            assert hip.hipMemcpy(h.ctypes.data, p, cnt * 8, 2) == 0
            bufs.append(h)
        tot = bufs[0] + bufs[1]
        for _, _, D in shards:
            p, cnt = D.exchange_buffer()
            assert hip.hipMemcpy(p, tot.ctypes.data, cnt * 8, 1) == 0
            D.iterate_global()
    for a, b, D in shards:
        assert rel_err(D.get_state("THETA_E"), M.state("THETA_E")[a:b]) < RTOL
        assert rel_err(D.get_state("BETA_E"), M.state("BETA_E")) < RTOL
        assert rel_err(D.get_state("IBIAS_E"), M.state("IBIAS_E")) < RTOL
        assert rel_err(D.get_state("UBIAS_E"), M.state("UBIAS_E")[a:b]) < RTOL


def test_empty_shard_and_empty_rows(orc):
    # users without any nonzero, items nobody rated
    from hgaprec_amd.capi import Hpf
    n, m, K = 50, 40, 6
    rowptr, col, val = make_problem(n, m, 600, 31)
    rp = np.concatenate([rowptr, np.full(5, rowptr[-1])]).astype(np.int64)
    n2, m2 = n + 5, m + 3
    M = orc.Model(n2, m2, K, True, False, False)
    M.set_csr(rp, col, val)
    M.initialize(3)
    D = Hpf(n2, m2, K)
    D.upload_csr(rp, col, val)
    copy_state(M, D, True, False)
    M.iterate(3)
    D.iterate(3)
    for w in compare_states(True, False):
        assert rel_err(D.get_state(w), M.state(w)) < RTOL, w
    assert np.allclose(D.get_state("THETA_SHAPE")[n:], 0.3)     # prior only


def test_set_elog_again_mid_run(orc):
    # re-seeding one side's Elog after sweeps ran (L is rebuilt lazily on the device)
    M, D = _run_pair(orc, 120, 90, 8, 2500, True, True, False, 2, seed=12)
    M.iterate(2)
    D.iterate(2)
    new = M.state("THETA_ELOG") * 0.5 - 0.1
    M.set_state("THETA_ELOG", new)
    D.set_state("THETA_ELOG", new)
    assert rel_err(D.get_state("UBIAS_ELOG"), M.state("UBIAS_ELOG")) < RTOL   # untouched column survived
    M.iterate(2)
    D.iterate(2)
    for w in compare_states(True, True):
        assert rel_err(D.get_state(w), M.state(w)) < RTOL, w


def test_library_rccl_allreduce_world1(orc):
    # the dlopen'ed RCCL path on real hardware: a 1-rank communicator, in-place
    # sum all-reduce of the exchange buffer (a no-op numerically)
    from hgaprec_amd.capi import Hpf
    M, D = _run_pair(orc, 80, 60, 6, 900, True, False, False, 1, seed=2)
    D.iterate(1)
    before = D.exchange_read()
    D.comm_init(Hpf.comm_unique_id())
    D.allreduce_exchange()
    D.synchronize()
    assert np.array_equal(D.exchange_read(), before)
    D.exchange_write(before * 2.0)
    assert np.array_equal(D.exchange_read(), before * 2.0)


def test_overlapped_exchange_sequence_world1(orc):
    # the order `hgaprec -ngpus N` uses, on a 1-rank communicator: item pass,
    # all-reduce of the item sums started on the library's second stream, user
    # pass + user sweep meanwhile, tail all-reduce, replicated item sweep.
    # Must equal plain hpf_iterate bit for bit (same kernels, same order).
    from hgaprec_amd.capi import Hpf
    M, D = _run_pair(orc, 300, 200, 20, 7000, True, True, False, 4, seed=6)
    _, D2 = _run_pair(orc, 300, 200, 20, 7000, True, True, False, 4, seed=6)
    D.comm_init(Hpf.comm_unique_id())
    for _ in range(4):
        D.iterate_local_items()
        D.allreduce_items_begin()
        D.iterate_local_users()
        D.allreduce_exchange()
        D.iterate_global()
    D2.iterate(4)
    M.iterate(4)
    for w in compare_states(True, True):
        assert np.array_equal(D.get_state(w), D2.get_state(w)), w
        assert rel_err(D.get_state(w), M.state(w)) < RTOL, w
    t = D.last_timing()
    assert t["phi_item_ms"] > 0 and t["phi_user_ms"] > 0 and t["sweep_item_ms"] > 0


def test_iteration_pieces_and_their_call_order(orc):
    from hgaprec_amd.capi import HpfError
    M, D = _run_pair(orc, 200, 150, 8, 4000, True, False, False, 3, seed=9)
    _, D2 = _run_pair(orc, 200, 150, 8, 4000, True, False, False, 3, seed=9)
    for bad in (D.iterate_global, D.iterate_local_users, D.iterate_local_sweep):
        with pytest.raises(HpfError, match="call order"):
            bad()
    D.iterate_local_items()
    with pytest.raises(HpfError, match="call order"):
        D.iterate_local_sweep()                        # the user pass has not run yet
    D.iterate_local_users()
    with pytest.raises(HpfError, match="call order"):
        D.iterate_local_users()
    D.iterate_global()
    D.iterate_local_phi(); D.iterate_local_sweep(); D.iterate_global()      # the other cut
    D.iterate_local(); D.iterate_global()
    D2.iterate(3)
    for w in compare_states(True, False):
        assert np.array_equal(D.get_state(w), D2.get_state(w)), w


@pytest.mark.parametrize("K,bias", [(1, False), (1, True), (2, False), (333, False), (512, False), (510, True),
                                    (700, False), (1024, False), (1022, True)])
def test_extreme_factor_counts(orc, K, bias):
    n, m = 40, 30
    M, D = _run_pair(orc, n, m, K, 400, True, bias, False, 2, seed=K)
    M.iterate(2)
    D.iterate(2)
    for w in compare_states(True, bias):
        assert rel_err(D.get_state(w), M.state(w)) < RTOL, w


def test_unsupported_and_invalid_inputs_fail_loudly():
    from hgaprec_amd.capi import Hpf, HpfError
    with pytest.raises(HpfError):
        Hpf(10, 10, 1025)                        # K > HPF_MAX_COLUMNS
    with pytest.raises(HpfError):
        Hpf(10, 10, 1023, bias=True)             # K + 2 > HPF_MAX_COLUMNS
    D = Hpf(4, 3, 2)
    with pytest.raises(HpfError):
        D.iterate(1)                             # no CSR, no state
    with pytest.raises(HpfError):
        D.upload_csr(np.array([0, 1, 2, 3, 4]), np.array([0, 1, 2, 7], np.uint32))   # item 7 out of range
    with pytest.raises(HpfError):
        D.upload_csr(np.array([0, 2, 1, 3, 4]), np.array([0, 1, 2, 1], np.uint32))   # rowptr not monotone
    D.upload_csr(np.array([0, 1, 2, 3, 4]), np.array([0, 1, 2, 1], np.uint32))
    with pytest.raises(HpfError):
        D.iterate(1)                             # state never set
    with pytest.raises(HpfError):
        D.elbo()
    with pytest.raises(HpfError):
        D.heldout_ll(np.array([9], np.uint32), np.array([0], np.uint32), np.array([1], np.int32))
    D2 = Hpf(10, 10, 4, n_ranks=2, rank=0, n_users_total=20)
    with pytest.raises(HpfError, match="hpf_comm_init"):
        D2.iterate(1)                            # several ranks, but nobody owns the exchange


def test_single_user_and_no_nonzeros(orc):
    from hgaprec_amd.capi import Hpf
    for rp, col in ((np.array([0, 3]), np.array([0, 1, 2], np.uint32)), (np.array([0, 0]), np.zeros(0, np.uint32))):
        val = np.full(col.size, 2, np.uint8)
        M = orc.Model(1, 3, 4, True, False, False)
        M.set_csr(rp, col, val)
        M.initialize(1)
        D = Hpf(1, 3, 4)
        D.upload_csr(rp, col, val)
        copy_state(M, D, True, False)
        M.iterate(2)
        D.iterate(2)
        for w in compare_states(True, False):
            assert rel_err(D.get_state(w), M.state(w)) < RTOL, w


@pytest.mark.parametrize("K,hier,bias", [(5, True, False), (21, True, True), (100, True, False), (50, False, True)])
def test_f32_stored_w_experimental_mode(orc, K, hier, bias):
    """EXPERIMENTAL storage mode (hpf_config.w_storage = 1; never the default):
    W = exp(Elog - rowmax) kept in fp32, arithmetic and accumulators fp64.  Each
    W carries a 2^-24 relative rounding which the CAVI map amplifies: measured
    (tests/w32_error_growth.py) 3e-7 after 5 sweeps, 2e-5 after 20, 1e-3 after
    60, up to 1e-1 after 300 -- it leaves the 1e-4 contract after ~30 sweeps,
    which is why the product stores W in fp64 (drift 1e-9 at 300 sweeps).
    The mode itself must work, be deterministic and be accurate for short runs."""
    outs = []
    for rep in range(2):
        M, D = _run_pair(orc, 400, 300, K, 12000, hier, bias, False, 5, seed=5, w_storage=1)
        D.iterate(5)
        outs.append((D.get_state("THETA_E"), D.get_state("BETA_E")))
    M.iterate(5)
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    et, eb = rel_err(outs[0][0], M.state("THETA_E")), rel_err(outs[0][1], M.state("BETA_E"))
    assert et < 2e-6 and eb < 2e-6, (et, eb)
    assert et > 1e-12                                   # it really is a different storage precision


@pytest.mark.parametrize("K,hier,bias,binary", [(5, True, False, False), (20, True, True, False), (100, True, False, False),
                                                (102, True, True, False), (50, False, True, True), (7, False, False, False)])
def test_48_bit_stored_w_opt_in_mode(orc, K, hier, bias, binary):
    """Opt-in storage mode hpf_config.w_storage = 2 (never the default): W keeps the top 48 bits of
    its fp64 value (36 mantissa bits, rounded to nearest even: 2^-37 relative), rows are 25-30 %
    shorter, arithmetic and accumulators stay fp64.  Its rounding is 8 000 times finer than the f32
    mode's and is amplified by the CAVI map the same way (tests/w32_error_growth.py): ~1e-11 per
    sweep early on, inside the 1e-4 contract after hundreds of sweeps."""
    outs = []
    for rep in range(2):
        M, D = _run_pair(orc, 400, 300, K, 12000, hier, bias, binary, 6, seed=5, w_storage=2)
        wi = D.work_info()
        D.iterate(6)
        outs.append([D.get_state(w) for w in compare_states(hier, bias)])
    M.iterate(6)
    assert wi["ld"] >= K + (2 if bias else 0) and wi["phi_V"] == 0 and wi["w_layout"] == 2     # the 48-bit kernel shape
    assert all(np.array_equal(a, b) for a, b in zip(*outs))                   # deterministic
    worst = 0.0
    for w, a in zip(compare_states(hier, bias), outs[0]):
        e = rel_err(a, M.state(w))
        assert e < (1e-6 if w.endswith("ELOG") else 1e-7), (w, e)   # 2-3 orders inside the contract (an Elog entry near zero weighs most)
        worst = max(worst, e)
    hu, hi, hy = heldout_pairs(400, 300, 500, seed=5)
    assert abs(D.heldout_ll(hu, hi, hy)[0] - M.heldout_sum(hu, hi, hy)) / hu.size < 1e-8
    assert abs(D.elbo() - M.elbo()) <= 1e-8 * abs(M.elbo())
    if K >= 20:
        assert worst > 1e-13                                                  # it really is a different storage precision


@pytest.mark.parametrize("K,hier,bias,binary", [(5, True, False, False), (5, True, True, False), (7, False, True, False),
                                                (21, True, True, False), (50, True, False, True), (100, True, False, False),
                                                (102, False, True, False), (200, True, True, False)])
@pytest.mark.parametrize("pack", ["packed", "plain", "plain_kernel"])
def test_lossless_packed_rows_and_plain_rows_both_match_the_oracle(orc, monkeypatch, K, hier, bias, binary, pack):
    """W rows are stored either as plain fp64 or LOSSLESSLY packed at 59 bits per element (sign and
    four exponent bits of a positive double <= 1 carry nothing).  The library packs by default where
    that saves a 128-byte line per row (K = 100: yes); here every shape is run three ways -- packing
    forced through HPF_W_PACK=1; w_storage = 3: plain doubles, in the packed shape's 16-byte pieces
    where the default would pack (layout 4: what a packed handle falls back to), in plain rows where
    it would not; and the plain-row KERNEL forced through its shape knob -- against the oracle at the
    same tolerance as the default path."""
    if pack != "plain":
        monkeypatch.setenv("HPF_EXPERIMENTAL", "1")
    if pack == "packed":
        monkeypatch.setenv("HPF_W_PACK", "1")
    if pack == "plain_kernel":
        cols = K + (2 if bias else 0)
        monkeypatch.setenv("HPF_PHI_CFG", "8,%d,2" % -(-cols // 16) if cols <= 128 else "16,%d,2" % -(-cols // 32))
    M, D = _run_pair(orc, 300, 200, K, 6000, hier, bias, binary, 6, seed=11 + K, w_storage=3 if pack == "plain" else 0)
    wi = D.work_info()
    packs_by_default = K in (50, 100, 102, 200)
    assert wi["w_layout"] == (3 if pack == "packed" else 0 if pack == "plain_kernel" else 4 if packs_by_default else 0), wi
    hu, hi, hy = heldout_pairs(300, 200, 500, seed=5)
    for it in range(6):
        M.iterate(1)
        D.iterate(1)
        for w in compare_states(hier, bias):
            e = rel_err(D.get_state(w), M.state(w))
            assert e < RTOL, f"{pack} iter {it} {w}: rel err {e:.3e}"
    assert abs(D.heldout_ll(hu, hi, hy)[0] - M.heldout_sum(hu, hi, hy)) / hu.size < 1e-9
    assert abs(D.elbo() - M.elbo()) <= 1e-10 * abs(M.elbo())


@pytest.mark.parametrize("ws", [0, 3], ids=["p59", "plain_doubles"])
@pytest.mark.parametrize("K", [13, 30, 64, 96, 128, 150, 256, 300, 500, 700])
def test_every_packed_kernel_shape_matches_the_oracle(orc, monkeypatch, K, ws):
    """a K sweep with the packing forced: lane groups of 8 to 64 lanes, 1 to 8 pieces per lane, sweep
    shapes with and without masked columns in the last slot (built in registers, or in LDS for 64-lane
    groups) -- and the same shapes holding plain doubles (w_storage = 3: up to 9 pieces per lane, what a
    packed handle falls back to) -- three sweeps each against the oracle"""
    monkeypatch.setenv("HPF_EXPERIMENTAL", "1")
    monkeypatch.setenv("HPF_W_PACK", "1")
    M, D = _run_pair(orc, 150, 120, K, 2500, True, K % 2 == 0, False, 3, seed=K, w_storage=ws)
    wi = D.work_info()
    assert wi["w_layout"] == (3 if ws == 0 else 4) and wi["ld"] >= K
    M.iterate(3); D.iterate(3)
    for w in compare_states(True, K % 2 == 0):
        e = rel_err(D.get_state(w), M.state(w))
        assert e < RTOL, (wi["phi_G"], wi["phi_R"], wi["sweep_G"], wi["sweep_R"], w, e)


@pytest.mark.parametrize("K,bias", [(100, False), (50, True), (202, False)])
def test_packed_rows_are_lossless_against_plain_rows(orc, monkeypatch, K, bias):
    """The 59-bit packing drops only bits that carry nothing: a packed run and a run with the same
    rows held as plain doubles (w_storage = 3: the packed shape's pieces, the same lane <-> column
    map) give identical bits after six sweeps, where the lossy 48-bit mode is visible at once."""
    from hgaprec_amd.capi import Hpf
    monkeypatch.setenv("HPF_EXPERIMENTAL", "1")
    monkeypatch.setenv("HPF_W_PACK", "1")
    n, m = 400, 300
    rowptr, col, val = make_problem(n, m, 12000, 5)
    M = orc.Model(n, m, K, True, bias, False)
    M.set_csr(rowptr, col, val); M.initialize(5)
    runs = {}
    for ws in (0, 3, 2):
        D = Hpf(n, m, K, hier=True, bias=bias, w_storage=ws)
        D.upload_csr(rowptr, col, val)
        copy_state(M, D, True, bias)
        D.iterate(6)
        runs[ws] = (D.work_info()["w_layout"], [D.get_state(w) for w in ("THETA_E", "BETA_E", "XI_E", "ETA_E")])
        D.close()
    assert [runs[ws][0] for ws in (0, 3, 2)] == [3, 4, 2]
    lossless = max(rel_err(a, b) for a, b in zip(runs[0][1], runs[3][1]))
    lossy = max(rel_err(a, b) for a, b in zip(runs[2][1], runs[3][1]))
    # the same lanes own the same columns in both forms: not a bit differs
    assert lossless == 0.0, lossless
    assert lossy > 1e-12, lossy                      # the 48-bit rounding is visible at once; the packing is not


@pytest.mark.parametrize("K,ws", [(100, 0), (100, 3), (100, 2), (50, 0)])
def test_gather_only_probe_leaves_the_model_alone(orc, K, ws):
    """hpf_gather_only: a phi pass with the arithmetic taken out (same work list, indices, rows), the
    ceiling bench.py prints beside the pass.  It must run for packed and plain 16-byte-piece rows,
    return a time, and change nothing."""
    M, D = _run_pair(orc, 500, 400, K, 20000, True, False, False, 2, seed=6, w_storage=ws)
    D.iterate(1)
    before = [D.get_state(w) for w in ("THETA_E", "BETA_E", "XI_E")]
    for side in (0, 1):
        assert D.gather_only_ms(side, 2) > 0.0
    D.iterate(1)
    M.iterate(2)
    assert rel_err(D.get_state("BETA_E"), M.state("BETA_E")) < RTOL
    E = _run_pair(orc, 500, 400, K, 20000, True, False, False, 2, seed=6, w_storage=ws)[1]
    E.iterate(1)
    assert all(np.array_equal(a, E.get_state(w)) for a, w in zip(before, ("THETA_E", "BETA_E", "XI_E")))


@pytest.mark.parametrize("K,layout", [(900, 3), (1000, 0), (1022, 0)])
def test_widest_rows_pack_when_the_sweep_has_a_shape_for_them(orc, K, layout):
    """K near HPF_MAX_COLUMNS: 900 columns pack (64 lanes x 15 = 960 columns, 56 lines instead of 64);
    from 961 live columns on the packed stride would be 1088 columns, for which the sweep has no
    shape: rows stay plain (1024 columns) instead of the handle being refused."""
    M, D = _run_pair(orc, 60, 50, K, 600, True, K == 1022, False, 2, seed=4)
    wi = D.work_info()
    assert wi["w_layout"] == layout and wi["ld"] >= K, wi
    M.iterate(2); D.iterate(2)
    for w in compare_states(True, K == 1022):
        assert rel_err(D.get_state(w), M.state(w)) < RTOL, w


def test_default_packs_k100_rows_and_falls_back_when_a_state_does_not_fit(orc):
    """K = 100: 104 columns x 59 bits = 768 bytes, six lines instead of seven, chosen by default.
    A W entry below 2^-126 of its row maximum (an Elog spread above 88 inside a row -- no HPF state
    has one) cannot be packed.  Round 4: the library does not fail on it -- it moves the rows to
    plain doubles and goes on.  Here the spread is forced MID-RUN through hpf_set_state; the run
    completes without intervention, matches the oracle at 1e-9 and a w_storage = 3 run bit for bit."""
    from hgaprec_amd.capi import Hpf
    n, m, K = 200, 150, 100
    M, D = _run_pair(orc, n, m, K, 4000, True, False, False, 2, seed=3)
    wi = D.work_info()
    assert wi["w_layout"] == 3 and wi["ld"] == 104 and wi["phi_G"] * wi["phi_R"] * 16 == 768 and wi["w_fallbacks"] == 0
    D.iterate(2); M.iterate(2)
    assert rel_err(D.get_state("BETA_E"), M.state("BETA_E")) < RTOL
    P = Hpf(n, m, K, hier=True, w_storage=3)
    P.upload_csr(*make_problem(n, m, 4000, 3))
    assert P.work_info()["w_layout"] == 4 and P.work_info()["ld"] == 104
    # the state of the run so far, exported once and handed to all three (D itself included: what it keeps inside --
    # W from its sweep, column sums from the sweep's own E -- is a rounding away from what the exported arrays give)
    st = {w: D.get_state(w) for w in init_states(True, False)}
    st["THETA_ELOG"][:, 0] -= 120.0                     # exp(-120) = 7.7e-53 < 2^-126
    for X in (D, P, M):
        for w, v in st.items():
            X.set_state(w, v)
    D.iterate(3); P.iterate(3); M.iterate(3)
    wd, wp = D.work_info(), P.work_info()
    assert wd["w_layout"] == 4 and wd["w_fallbacks"] == 1 and wp["w_layout"] == 4 and wp["w_fallbacks"] == 0, (wd, wp)
    for w in compare_states(True, False):
        assert np.array_equal(D.get_state(w), P.get_state(w)), w
        assert rel_err(D.get_state(w), M.state(w)) < RTOL, w
    # a snapshot taken after the move loads into a fresh (packing) handle, which follows
    blob = D.snapshot()
    F = Hpf(n, m, K, hier=True)
    F.upload_csr(*make_problem(n, m, 4000, 3))
    F.restore(blob)
    assert F.work_info()["w_layout"] == 4
    F.iterate(1); D.iterate(1)
    assert np.array_equal(F.get_state("BETA_E"), D.get_state("BETA_E"))
    for X in (D, P, F):
        X.close()


@pytest.mark.parametrize("graph", ["0", "1"])
def test_a_sweep_that_cannot_pack_its_row_stops_the_passes_and_nothing_is_lost(orc, monkeypatch, graph):
    """The same in the place where it would really happen: inside hpf_iterate, in a SWEEP.  A column of
    E[beta] of 1e40 makes the first user sweep's rate in that column ~1e42: W there is e^-97 of the row
    maximum.  The sweep raises the flag; the passes launched after it return at once (they must not read
    those rows); at the next synchronisation point the library moves to plain doubles, repeats the sweeps'
    W from what they read, and runs the iterations that were skipped.  Four iterations launched without a
    look in between end where a w_storage = 3 run ends, bit for bit -- also under hipGraph replay."""
    from hgaprec_amd.capi import Hpf
    monkeypatch.setenv("HPF_EXPERIMENTAL", "1")
    monkeypatch.setenv("HPF_GRAPH", graph)
    n, m, K = 300, 200, 100
    rowptr, col, val = make_problem(n, m, 8000, 7, heavy_item=True)
    M = orc.Model(n, m, K, True, False, False)
    M.set_csr(rowptr, col, val); M.initialize(7)
    be = M.state("BETA_E").copy()
    be[:, 0] = 1e40
    M.set_state("BETA_E", be)
    runs = {}
    for ws in (0, 3):
        D = Hpf(n, m, K, hier=True, w_storage=ws)
        D.upload_csr(rowptr, col, val)
        copy_state(M, D, True, False)
        D.iterate(4)                                    # one call, no synchronisation inside
        runs[ws] = ({w: D.get_state(w) for w in compare_states(True, False)}, D.work_info(), D.last_timing()["iterations"])
        D.close()
    M.iterate(4)
    assert runs[0][1]["w_layout"] == 4 and runs[0][1]["w_fallbacks"] == 1 and runs[3][1]["w_fallbacks"] == 0, runs[0][1]
    assert runs[0][2] == 4 and runs[3][2] == 4
    for w in compare_states(True, False):
        assert np.array_equal(runs[0][0][w], runs[3][0][w]), w
        assert rel_err(runs[0][0][w], M.state(w)) < RTOL, w


def test_the_same_on_two_ranks_the_host_looks_before_every_iteration(orc):
    """several ranks: a pass that skipped would leave its rank's sums out of the all-reduce, so there the
    host looks at the flag before every iteration and repairs the rows first (hpf_iterate_local_items)"""
    from hgaprec_amd.capi import Hpf
    from hgaprec_amd import dist as hd
    n, m, K = 300, 200, 100
    rowptr, col, val = make_problem(n, m, 8000, 7, heavy_item=True)
    M = orc.Model(n, m, K, True, False, False)
    M.set_csr(rowptr, col, val); M.initialize(7)
    be = M.state("BETA_E").copy()
    be[:, 0] = 1e40
    M.set_state("BETA_E", be)
    init = {w: M.state(w).copy() for w in init_states(True, False)}
    parts = hd.partition_users(rowptr, 2)
    shards = []
    for r, (a, b) in enumerate(parts):
        D = Hpf(b - a, m, K, hier=True, n_ranks=2, rank=r, n_users_total=n)
        D.upload_csr(*hd.shard_csr(rowptr, col, val, a, b))
        hd.scatter_state(D, init, a, b, hier=True)
        shards.append(D)
    for it in range(4):
        M.iterate(1)
        for D in shards:
            D.iterate_local()
        tot = shards[0].exchange_read() + shards[1].exchange_read()
        for D in shards:
            D.exchange_write(tot)
            D.iterate_global()
    for (a, b), D in zip(parts, shards):
        assert D.work_info()["w_fallbacks"] == 1 and D.work_info()["w_layout"] == 4
        for w in compare_states(True, False):
            want = M.state(w)
            if w.startswith(("THETA_", "XI_")):
                want = want[a:b]
            assert rel_err(D.get_state(w), want) < RTOL, w
        D.close()


def test_48_bit_stored_w_stays_inside_the_contract_over_a_long_run(orc):
    M, D = _run_pair(orc, 400, 300, 50, 12000, True, True, False, 150, seed=5, w_storage=2)
    M.iterate(150)
    D.iterate(150)
    for w in ("THETA_E", "BETA_E", "XI_E", "ETA_E", "UBIAS_E", "IBIAS_E"):
        assert rel_err(D.get_state(w), M.state(w)) < 1e-4, w           # north_star's tolerance


def test_fp64_drift_stays_tiny_over_many_sweeps(orc):
    # the default path against the oracle after 150 sweeps (contract: 1e-4)
    M, D = _run_pair(orc, 400, 300, 50, 12000, True, True, False, 150, seed=5)
    M.iterate(150)
    D.iterate(150)
    for w in ("THETA_E", "BETA_E", "XI_E", "ETA_E", "UBIAS_E", "IBIAS_E"):
        assert rel_err(D.get_state(w), M.state(w)) < 1e-8, w


def test_softmax_underflow_is_reported_not_hidden(orc):
    """rows whose Elog spread exceeds what the stored W can represent make every
    product of a nonzero vanish; the device raises a flag and the next
    synchronising call fails instead of silently dropping the nonzero"""
    from hgaprec_amd.capi import Hpf, HpfError
    n, m, K = 30, 20, 2
    rowptr, col, val = make_problem(n, m, 200, 3)
    M = orc.Model(n, m, K, True, False, False)
    M.set_csr(rowptr, col, val)
    M.initialize(1)
    lt = np.tile(np.array([0.0, -150.0]), (n, 1))
    lb = np.tile(np.array([-150.0, 0.0]), (m, 1))
    for ws, should_fail in ((1, True), (0, False)):     # e^-150 is 0 in fp32, fine in fp64
        D = Hpf(n, m, K, w_storage=ws)
        D.upload_csr(rowptr, col, val)
        copy_state(M, D, True, False)
        D.set_state("THETA_ELOG", lt)
        D.set_state("BETA_ELOG", lb)
        D.iterate(1)
        if should_fail:
            with pytest.raises(HpfError, match="underflow"):
                D.synchronize()
        else:
            D.synchronize()
            assert np.all(np.isfinite(D.get_state("THETA_E")))


def test_bound_heldout_sets_give_the_sum_of_the_unbound_call_bit_for_bit(orc):
    """hpf_heldout_bind / hpf_heldout_ll_bound (ABI v8): a set validated and uploaded once gives, at every later state, the
    very sum hpf_heldout_ll gives for the same pairs; slots are independent; an index out of range is refused at the bind
    and leaves the slot as it was; an unbound slot says so; binding again replaces the set; an empty set sums to 0."""
    from hgaprec_amd.capi import HpfError
    M, D = _run_pair(orc, 300, 200, 20, 6000, True, True, False, 2, seed=4)
    hu, hi, hy = heldout_pairs(300, 200, 700, seed=5)
    hu2, hi2, hy2 = heldout_pairs(300, 200, 90, seed=6)
    D.heldout_bind(0, hu, hi, hy)
    D.heldout_bind(1, hu2, hi2, hy2)
    for it in range(3):
        assert D.heldout_ll_bound(0) == D.heldout_ll(hu, hi, hy)
        assert D.heldout_ll_bound(1) == D.heldout_ll(hu2, hi2, hy2)
        M.iterate(1); D.iterate(1)
    assert abs(D.heldout_ll_bound(0)[0] - M.heldout_sum(hu, hi, hy)) / hu.size < 1e-9
    with pytest.raises(HpfError):
        D.heldout_ll_bound(2)
    bad = hu.copy(); bad[3] = 300
    with pytest.raises(HpfError):
        D.heldout_bind(0, bad, hi, hy)
    assert D.heldout_ll_bound(0) == D.heldout_ll(hu, hi, hy)          # the refused bind left the slot alone
    D.heldout_bind(0, hu2, hi2, hy2)
    assert D.heldout_ll_bound(0) == D.heldout_ll_bound(1)
    D.heldout_bind(3, hu[:0], hi[:0], hy[:0])
    assert D.heldout_ll_bound(3) == (0.0, 0)
    with pytest.raises(HpfError):
        D.heldout_bind(4, hu, hi, hy)
    D.close()
