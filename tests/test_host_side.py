"""CPU: the product's C++ host side (libhgaprec_host.so: Env naming, TSV
reader, MT19937 start state, writers, stop rule) against the reference-made
golden fixtures and against the oracle's independent restatement."""
import json
import os
from pathlib import Path

import numpy as np
import pytest

from hgaprec_amd import hostlib
from hgaprec_amd.capi import STATE_NAMES

GOLD = Path(__file__).parent / "golden"


def unhex(lst):
    return np.array([float.fromhex(s) for s in lst], dtype=np.float64)


def write_tsv(path, triples):
    with open(path, "w") as f:
        for u, i, y in triples:
            f.write(f"{u}\t{i}\t{y}\n")


# ---------------------------------------------------------------- Env --------
def test_prefix_matches_reference_env():
    d = json.loads((GOLD / "env.json").read_text())
    for c in d["cases"]:
        assert hostlib.prefix(c["args"]) == c["prefix"], c["args"]


def test_file_str_matches_reference_env():
    # Env::file_str (env.hh:209-214): every output path is prefix + name
    d = json.loads((GOLD / "env.json").read_text())
    for c in d["cases"]:
        assert hostlib.prefix(c["args"]) + "/x.tsv" == c["file_str_x_tsv"], c["args"]


def test_param_txt_head_matches_reference_env(tmp_path):
    d = json.loads((GOLD / "env.json").read_text())
    cwd = os.getcwd()
    try:
        for k, c in enumerate(d["cases"]):
            wd = tmp_path / f"c{k}"
            wd.mkdir()
            os.chdir(wd)
            p = hostlib.open_output(c["args"])
            assert p == c["prefix"]
            assert (wd / p / "param.txt").read_text() == c["param_txt"]
            assert sorted(os.listdir(wd / p)) == c["files"]
    finally:
        os.chdir(cwd)


def test_unknown_option_is_rejected():
    with pytest.raises(ValueError):
        hostlib.prefix(["-dir", "x", "-frobnicate"])


# ---------------------------------------------------------------- RNG --------
@pytest.mark.parametrize("seed", [0, 1, 7, 2 ** 31, 12345.9])
def test_mt19937_stream(orc, seed):
    got = hostlib.mt_u32(seed, 1500)
    r = orc.Rng(0)
    if seed:
        r = orc.Rng(int(seed))
    want = np.array([r.u32() for _ in range(1500)], np.uint32)
    assert np.array_equal(got, want)
    rs = np.random.RandomState(int(seed) if seed else 4357)
    assert np.array_equal(got.astype(np.uint64), rs.randint(0, 2 ** 32, size=1500, dtype=np.uint64))


def test_gsl_rng_seed_env(monkeypatch):
    monkeypatch.setenv("GSL_RNG_SEED", "99")
    a = hostlib.mt_u32(0, 10)
    monkeypatch.delenv("GSL_RNG_SEED")
    assert np.array_equal(a, hostlib.mt_u32(99, 10))


def test_host_digamma(orc):
    xs = np.concatenate([np.linspace(0.3, 0.32, 40), np.geomspace(1e-3, 1e7, 200)])
    got, want = hostlib.digamma(xs), orc.psi(xs)
    assert np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))) < 2e-15


# --------------------------------------------------------- start state -------
@pytest.mark.parametrize("hier,bias", [(True, False), (True, True), (False, False), (False, True)])
@pytest.mark.parametrize("seed", [0, 7])
def test_initial_state_matches_oracle(orc, hier, bias, seed):
    n, m, K = 37, 23, 6
    st = hostlib.initial_state(seed, n, m, K, hier, bias)
    M = orc.Model(n, m, K, hier, bias, False)
    M.initialize(seed)
    for name in STATE_NAMES:
        try:
            want = M.state(name)
        except KeyError:
            assert name not in st or st[name].size == 0
            continue
        got = st[name]
        if name.endswith("ELOG"):
            assert np.max(np.abs(got - want)) < 1e-14, name      # psi: two implementations
        else:
            assert np.array_equal(got.reshape(want.shape), want), name   # same MT words, same IEEE ops


@pytest.mark.parametrize("hier,bias", [(True, False), (True, True), (False, True)])
def test_a_ranks_slice_of_the_start_state_is_the_slice_of_the_whole(hier, bias):
    """a rank of several draws the whole stream and keeps rows [lo, hi) of the user side (block-wise
    draws, dropped draws for the rows of others): the slice of the one-process state, bit for bit, the
    item side whole, and the generator left at the same word"""
    n, m, K = 700, 300, 130                                   # rows straddle the 624-word blocks of the generator
    whole = hostlib.initial_state(5, n, m, K, hier, bias, rows=(0, n))
    plain = hostlib.initial_state(5, n, m, K, hier, bias)
    for name, a in plain.items():
        assert np.array_equal(a, whole[name]), name
    user_side = {"THETA_SHAPE", "THETA_E", "THETA_ELOG", "XI_SHAPE", "XI_RATE", "XI_E", "XI_ELOG",
                 "UBIAS_SHAPE", "UBIAS_RATE", "UBIAS_E", "UBIAS_ELOG"} | ({"THETA_RATE"} if hier else set())
    for lo, hi in ((0, 1), (1, 699), (250, 251), (333, 700), (700, 700)):
        part = hostlib.initial_state(5, n, m, K, hier, bias, rows=(lo, hi))
        assert part["_word_after"] == whole["_word_after"]
        for name, a in plain.items():
            got = part.get(name, np.zeros(0))
            want = a[lo:hi] if name in user_side else a
            assert np.array_equal(np.asarray(got).reshape(np.asarray(want).shape), want), (name, lo, hi)


def test_initial_state_on_many_threads_is_the_one_thread_state(orc, monkeypatch):
    """above 2^16 elements the digamma / log half of a start state runs on the host's threads:
    same bits as on one thread (HGAPREC_SAVE_THREADS=1), same values as the oracle"""
    n, m, K = 1500, 700, 50                                  # theta 75 000, beta 35 000 elements
    st = hostlib.initial_state(3, n, m, K, True, False)
    monkeypatch.setenv("HGAPREC_SAVE_THREADS", "1")
    one = hostlib.initial_state(3, n, m, K, True, False)
    monkeypatch.delenv("HGAPREC_SAVE_THREADS")
    monkeypatch.setenv("HGAPREC_SAVE_THREADS", "7")          # pieces that do not divide the count
    odd = hostlib.initial_state(3, n, m, K, True, False)
    for name in st:
        assert np.array_equal(st[name], one[name]) and np.array_equal(st[name], odd[name]), name
    M = orc.Model(n, m, K, True, False, False)
    M.initialize(3)
    for name in ("THETA_E", "BETA_E", "THETA_SHAPE", "BETA_RATE"):
        assert np.array_equal(st[name].reshape(M.state(name).shape), M.state(name)), name
    for name in ("THETA_ELOG", "BETA_ELOG"):
        assert np.max(np.abs(st[name].reshape(M.state(name).shape) - M.state(name))) < 1e-14, name


# -------------------------------------------------------------- reader -------
@pytest.fixture(params=["token-by-token", "on-threads"], autouse=True)
def reader_mode(request, monkeypatch):
    """every test of this file runs with both readers: the token-by-token one (small files, its default)
    and the one that parses a well-formed file in pieces on the host's threads, forced onto files of
    any size with an odd number of threads (it must hand everything else back to the first)"""
    if request.param == "on-threads":
        monkeypatch.setenv("HGAPREC_READ_PARALLEL_MIN", "0")
        monkeypatch.setenv("HGAPREC_READ_THREADS", "3")
    else:
        monkeypatch.setenv("HGAPREC_READ_THREADS", "1")
    return request.param


def test_reader_on_threads_equals_token_by_token_on_a_file_that_takes_the_fast_path(orc, tmp_path, monkeypatch, reader_mode):
    """3e5 records, users grouped and not, ratings 0 mixed in, ids that first appear in a later piece;
    2 .. 13 threads; capacities that bind (the fall-back inside the fast path) and that do not"""
    if reader_mode != "on-threads":
        pytest.skip("compares the two readers itself")
    rng = np.random.default_rng(11)
    R = 300_000
    u = np.sort(rng.integers(5, 40_000, R)).astype(np.int64)
    u[R // 2:] = rng.integers(5, 60_000, R - R // 2)          # second half: not grouped, new users late
    it = (rng.zipf(1.3, R) % 9_000).astype(np.int64) + 1
    y = rng.integers(0, 6, R).astype(np.int64)
    rows = np.stack([u, it, y], 1)
    np.savetxt(tmp_path / "train.tsv", rows, fmt="%d", delimiter="\t")
    np.savetxt(tmp_path / "validation.tsv", rows[::7] + np.array([0, 0, 1]), fmt="%d", delimiter="\t")
    np.savetxt(tmp_path / "test.tsv", rows[3::11][:, [0, 1, 2]] + np.array([1, 0, 0]), fmt="%d", delimiter=" ")
    for cap_n, cap_m in ((100_000, 100_000), (30_000, 100_000), (100_000, 2_000)):
        def read(threads):
            monkeypatch.setenv("HGAPREC_READ_THREADS", str(threads))
            H = hostlib.Ratings(cap_n, cap_m, False, 1)
            assert H.read_train(tmp_path / "train.tsv") == 0
            for w, name in ((0, "validation.tsv"), (1, "test.tsv")):
                assert H.read_heldout(tmp_path / name, w) == 0
            return H
        ref = read(1)
        assert ref.nnz > 0.7 * R or cap_n < 100_000 or cap_m < 100_000
        for threads in (2, 5, 13):
            _assert_same(read(threads), ref)
    O = orc.Ratings(30_000, 100_000, False, 1)                  # and the oracle on the case with a binding capacity
    assert O.read_train(tmp_path / "train.tsv") == 0
    monkeypatch.setenv("HGAPREC_READ_THREADS", "5")
    H = hostlib.Ratings(30_000, 100_000, False, 1)
    assert H.read_train(tmp_path / "train.tsv") == 0
    assert (H.n, H.m, H.nnz) == (O.n, O.m, O.nnz) and np.array_equal(H.seq2user(), O.seq2user())
    for a, b in zip(H.csr(), O.csr()):
        assert np.array_equal(a, b)


def test_reader_hands_unusual_files_back(tmp_path, monkeypatch, reader_mode):
    """a sign, a letter, a token count that is no multiple of three: the token-by-token reader's result"""
    cases = {"sign": "1 2 3\n4 -5 1\n6 7 2\n", "short": "1 2 3\n4 5 1\n6 7\n", "letter": "1 2 3\n4 x 1\n", "plus": "+1 2 3\n4 5 +1\n"}
    for name, text in cases.items():
        (tmp_path / f"{name}.tsv").write_text(text)
        H = hostlib.Ratings(10, 10)
        rc = H.read_train(tmp_path / f"{name}.tsv")
        monkeypatch.setenv("HGAPREC_READ_THREADS", "1")
        S = hostlib.Ratings(10, 10)
        assert S.read_train(tmp_path / f"{name}.tsv") == rc
        monkeypatch.setenv("HGAPREC_READ_THREADS", "3")
        assert (H.n, H.m, H.nnz) == (S.n, S.m, S.nnz)
        if rc == 0:
            for a, b in zip(H.csr(), S.csr()):
                assert np.array_equal(a, b)
    assert hostlib.Ratings(10, 10).read_train(tmp_path / "letter.tsv") == -2


def _both_readers(orc, tmp_path, train, valid, test, cap_n, cap_m, binary=False, thr=1):
    write_tsv(tmp_path / "train.tsv", train)
    write_tsv(tmp_path / "validation.tsv", valid)
    write_tsv(tmp_path / "test.tsv", test)
    H = hostlib.Ratings(cap_n, cap_m, binary, thr)
    O = orc.Ratings(cap_n, cap_m, binary, thr)
    assert H.read_train(tmp_path / "train.tsv") == 0
    assert O.read_train(tmp_path / "train.tsv") == 0
    for w, name in ((0, "validation.tsv"), (1, "test.tsv")):
        assert H.read_heldout(tmp_path / name, w) == 0
        assert O.read_heldout(tmp_path / name, w) == 0
    return H, O


def _assert_same(H, O):
    assert (H.n, H.m, H.nnz) == (O.n, O.m, O.nnz)
    for a, b in zip(H.csr(), O.csr()):
        assert np.array_equal(a, b)
    assert np.array_equal(H.seq2user(), O.seq2user())
    assert np.array_equal(H.seq2item(), O.seq2item())
    for w in (0, 1):
        for a, b in zip(H.heldout(w), O.heldout(w)):
            assert np.array_equal(a, b)


def test_reader_edge_cases(orc, tmp_path):
    train = [
        (50, 900, 3), (10, 901, 5), (50, 901, 0),     # rating 0 dropped
        (77, 902, 0),                                # user 77 only has a dropped rating: never registered
        (50, 900, 4),                                # duplicate (u,i): listed twice, last rating wins
        (10, 900, 300),                              # 300 -> uint8 44
        (10, 903, 256),                              # 256 -> uint8 0 (kept: class is 256, value wraps)
        (11, 904, 1), (12, 905, 2), (13, 900, 1),     # ids out of order
    ]
    valid = [(50, 903, 2), (99, 900, 1), (10, 999, 1), (50, 903, 5), (12, 900, 0)]  # unseen ids skipped; dup pair
    test = [(13, 901, 4), (11, 900, 260)]
    H, O = _both_readers(orc, tmp_path, train, valid, test, 100, 100)
    _assert_same(H, O)
    rp, col, val = H.csr()
    assert H.n == 5 and H.m == 5                            # item 902 only had a dropped rating
    assert list(H.seq2user()) == [50, 10, 11, 12, 13]
    assert list(col[rp[0]:rp[1]]) == [0, 0]                  # the duplicate appears twice
    assert list(val[rp[0]:rp[1]]) == [4, 4]                  # both carry the last rating
    assert list(val[rp[1]:rp[2]]) == [5, 300 % 256, 0]
    hu, hi, hy = H.heldout(0)
    assert list(zip(hu, hi, hy)) == [(0, 2, 5)]              # (50,903) overwritten by the later value
    tu, ti, ty = H.heldout(1)
    assert list(zip(tu, ti, ty)) == [(2, 0, 260), (4, 1, 4)]  # map order; int kept, wrapped at use


def test_dataset_cache_round_trip_and_staleness(orc, tmp_path):
    # -cache (extension): the binary image must give back exactly
    # what the three TSV parses gave, and must be refused when anything it was
    # built from changed
    rng = np.random.default_rng(3)
    train = [(int(u), int(i), int(r)) for u, i, r in
             zip(rng.integers(1000, 1400, 6000), rng.integers(1, 300, 6000), rng.integers(0, 7, 6000))]
    valid = [(int(u), int(i), int(r)) for u, i, r in
             zip(rng.integers(1000, 1420, 400), rng.integers(1, 320, 400), rng.integers(0, 6, 400))]
    test = valid[::-1][:150] + [(1001, 5, 2)]
    H, O = _both_readers(orc, tmp_path, train, valid, test, 1000, 1000)
    assert H.save_cache(tmp_path) == 0
    C = hostlib.Ratings(1000, 1000, False, 1)
    assert C.load_cache(tmp_path) == 0
    _assert_same(C, O)
    # the id -> seq maps are rebuilt: test_users.tsv lookups work on a cached dataset
    (tmp_path / "test_users.tsv").write_text("1001\n999999\n%d\n1001\n" % int(H.seq2user()[7]))
    assert np.array_equal(C.test_users(tmp_path / "test_users.tsv"), H.test_users(tmp_path / "test_users.tsv"))
    assert len(C.test_users(tmp_path / "test_users.tsv")) == 2
    # other parameters: refused
    assert hostlib.Ratings(999, 1000, False, 1).load_cache(tmp_path) == 1
    assert hostlib.Ratings(1000, 1000, True, 1).load_cache(tmp_path) == 1
    assert hostlib.Ratings(1000, 1000, False, 2).load_cache(tmp_path) == 1
    # damaged image: refused (and nothing half-loaded)
    img = (tmp_path / "hgaprec.cache.bin").read_bytes()
    (tmp_path / "hgaprec.cache.bin").write_bytes(img[:-9])
    D = hostlib.Ratings(1000, 1000, False, 1)
    assert D.load_cache(tmp_path) == 1 and D.n == 0 and D.nnz == 0
    (tmp_path / "hgaprec.cache.bin").write_bytes(img)
    assert hostlib.Ratings(1000, 1000, False, 1).load_cache(tmp_path) == 0
    # a source file changed after the image was written: refused
    with open(tmp_path / "validation.tsv", "a") as f:
        f.write("1001\t7\t3\n")
    assert hostlib.Ratings(1000, 1000, False, 1).load_cache(tmp_path) == 1
    # no image at all
    (tmp_path / "hgaprec.cache.bin").unlink()
    assert hostlib.Ratings(1000, 1000, False, 1).load_cache(tmp_path) == 1


def test_reader_capacity_limits(orc, tmp_path):
    rng = np.random.default_rng(5)
    train = [(int(u), int(i), int(y)) for u, i, y in
             zip(rng.integers(0, 40, 400), rng.integers(0, 30, 400), rng.integers(0, 6, 400))]
    valid = [(int(u), int(i), int(y)) for u, i, y in
             zip(rng.integers(0, 45, 60), rng.integers(0, 35, 60), rng.integers(0, 6, 60))]
    H, O = _both_readers(orc, tmp_path, train, valid, valid[:20], cap_n=15, cap_m=12)
    _assert_same(H, O)
    assert H.n == 15 and H.m == 12                       # more distinct ids than -n / -m: skipped


def test_reader_binary_and_threshold(orc, tmp_path):
    rng = np.random.default_rng(6)
    tr = [(int(u), int(i), int(y)) for u, i, y in
          zip(rng.integers(0, 30, 300), rng.integers(0, 20, 300), rng.integers(0, 6, 300))]
    for thr in (1, 4):
        H, O = _both_readers(orc, tmp_path, tr, tr[:50], tr[50:90], 100, 100, binary=True, thr=thr)
        _assert_same(H, O)
        assert set(H.csr()[2].tolist()) <= {1}
        assert set(H.heldout(0)[2].tolist()) <= {1}


def test_reader_whitespace_and_no_trailing_newline(orc, tmp_path):
    (tmp_path / "train.tsv").write_text("1 2 3\n4\t5\t1\n\n  6\t7   2")
    H = hostlib.Ratings(10, 10)
    O = orc.Ratings(10, 10)
    assert H.read_train(tmp_path / "train.tsv") == 0 and O.read_train(tmp_path / "train.tsv") == 0
    assert H.nnz == 3 and O.nnz == 3
    for a, b in zip(H.csr(), O.csr()):
        assert np.array_equal(a, b)


def test_reader_empty_file_is_the_references_error(orc, tmp_path):
    (tmp_path / "e.tsv").write_text("")
    assert hostlib.Ratings(5, 5).read_train(tmp_path / "e.tsv") == -2     # "unexpected lines" exit(-1)
    assert orc.Ratings(5, 5).read_train(tmp_path / "e.tsv") == -1
    assert hostlib.Ratings(5, 5).read_train(tmp_path / "missing.tsv") == -1


def test_marginals_files(orc, tmp_path):
    rng = np.random.default_rng(8)
    tr = [(int(u) + 100, int(i) + 500, int(y)) for u, i, y in
          zip(rng.integers(0, 25, 200), rng.integers(0, 15, 200), rng.integers(1, 6, 200))]
    H, O = _both_readers(orc, tmp_path, tr, tr[:5], tr[:5], 100, 100)
    H.write_marginals(tmp_path / "hu.tsv", tmp_path / "hi.tsv")
    O.write_marginals(tmp_path / "ou.tsv", tmp_path / "oi.tsv")
    assert (tmp_path / "hu.tsv").read_text() == (tmp_path / "ou.tsv").read_text()
    assert (tmp_path / "hi.tsv").read_text() == (tmp_path / "oi.tsv").read_text()
    first = (tmp_path / "hu.tsv").read_text().splitlines()[0].split("\t")
    assert first[0] == "0" and int(first[1]) == tr[0][0]


# ------------------------------------------------------------- writers -------
def test_writers_match_reference_format(tmp_path):
    d = json.loads((GOLD / "save.json").read_text())
    for c in d["cases"]:
        A = unhex(c["A"]).reshape(c["rows"], c["cols"])
        ids = np.array(c["ids"], np.uint32)
        hostlib.save_matrix(tmp_path / "m.tsv", A, ids)
        hostlib.save_vector(tmp_path / "v.tsv", unhex(c["v"]), ids)
        assert (tmp_path / "m.tsv").read_text() == c["matrix_tsv"]
        assert (tmp_path / "v.tsv").read_text() == c["vector_tsv"]


def test_a_save_over_a_longer_file_leaves_no_tail(tmp_path):
    """model files are rewritten in place (no truncation up front, the length set at the end): a shorter matrix or
    vector over a longer file must leave exactly the new text"""
    rng = np.random.default_rng(2)
    big, small = rng.gamma(0.3, 1.0, size=(400, 7)), rng.gamma(0.3, 1.0, size=(13, 7))
    for a in (big, small, big[:50]):
        assert hostlib.save_matrix(tmp_path / "m.tsv", a, None) == 0
        assert hostlib.save_matrix(tmp_path / "fresh.tsv", a, None) == 0
        assert (tmp_path / "m.tsv").read_bytes() == (tmp_path / "fresh.tsv").read_bytes()
        (tmp_path / "fresh.tsv").unlink()
        assert hostlib.save_vector(tmp_path / "v.tsv", a[:, 0].copy(), None) == 0
        assert len((tmp_path / "v.tsv").read_text().splitlines()) == a.shape[0]


def test_a_save_that_fails_midway_is_marked_and_cut(tmp_path):
    """the in-place rewrite lost what fopen("w") gave -- a run that dies in the middle of a save left a
    full-length file, new text over old.  Now "<file>.writing" lies beside a file from the first byte until its length
    has been set: gone after a good save, still there after a failed one, and the failed file is cut at the last byte
    written (no old tail).  The failure: a file-size limit (RLIMIT_FSIZE) that the second save runs into."""
    import subprocess
    import sys
    code = r"""
import resource, signal, sys
import numpy as np
sys.path.insert(0, %r)
from hgaprec_amd import hostlib
path = sys.argv[1]
rng = np.random.default_rng(1)
big = rng.gamma(0.3, 1.0, size=(3000, 20))
assert hostlib.save_matrix(path, big, None) == 0
import os
assert not os.path.exists(path + ".writing")
full = os.path.getsize(path)
signal.signal(signal.SIGXFSZ, signal.SIG_IGN)
resource.setrlimit(resource.RLIMIT_FSIZE, (full // 3, resource.RLIM_INFINITY))
rc = hostlib.save_matrix(path, big * 2.0, None)
print(rc, full, os.path.getsize(path), os.path.exists(path + ".writing"))
""" % str(Path(__file__).resolve().parent.parent)
    r = subprocess.run([sys.executable, "-c", code, str(tmp_path / "m.tsv")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    rc, full, now, marker = r.stdout.split()
    assert int(rc) != 0 and marker == "True"
    assert int(now) <= int(full) // 3                      # cut where the writing stopped: no old text behind the new
    # the next good save clears the marker
    assert hostlib.save_matrix(tmp_path / "m.tsv", np.ones((5, 3)), None) == 0
    assert not (tmp_path / "m.tsv.writing").exists() and len((tmp_path / "m.tsv").read_text().splitlines()) == 5


def test_a_target_that_cannot_be_opened_gets_no_marker(tmp_path):
    """the marker used to be created before the open of the file itself, so a target that could not be opened -- an
    old file nobody had touched, still whole -- got a "<file>.writing" beside it, which says "not whole"; and a target that
    is no regular file (/dev/null) got one as well.  Now the marker follows a successful open, beside regular files only."""
    import os
    d = tmp_path / "ro"
    d.mkdir()
    good = d / "old.tsv"
    assert hostlib.save_matrix(good, np.ones((4, 3)), None) == 0
    before = good.read_text()
    # a directory in place of the target: open(O_WRONLY) fails whoever runs the test (root ignores file modes)
    (d / "isdir.tsv").mkdir()
    assert hostlib.save_matrix(d / "isdir.tsv", np.ones((4, 3)), None) != 0
    assert not (d / "isdir.tsv.writing").exists()
    # a path under a missing directory: the target cannot be created
    assert hostlib.save_matrix(d / "missing" / "x.tsv", np.ones((2, 2)), None) != 0
    assert not (d / "missing").exists()
    assert good.read_text() == before and not (d / "old.tsv.writing").exists()
    # not a regular file: written, never marked
    assert hostlib.save_vector("/dev/null", np.ones(5), None) == 0
    assert not os.path.exists("/dev/null.writing")


def test_read_threads_override_is_clamped(tmp_path, monkeypatch):
    """HGAPREC_READ_THREADS = -3 or garbage used to become 64 threads through an unsigned cast; and text too
    large for a quarter of the free memory (HGAPREC_READ_PARALLEL_MAX stands in) goes to the token-by-token reader --
    the same result either way"""
    rng = np.random.default_rng(8)
    n, m, cnt = 300, 200, 40000
    u, i, y = rng.integers(1, n + 1, cnt), rng.integers(1, m + 1, cnt), rng.integers(0, 6, cnt)
    (tmp_path / "train.tsv").write_text("".join(f"{a}\t{b}\t{c}\n" for a, b, c in zip(u, i, y)))
    monkeypatch.setenv("HGAPREC_READ_PARALLEL_MIN", "1")
    got = []
    for threads, pmax in (("4", None), ("-3", None), ("zebra", None), ("4", "1000")):
        monkeypatch.setenv("HGAPREC_READ_THREADS", threads)
        if pmax:
            monkeypatch.setenv("HGAPREC_READ_PARALLEL_MAX", pmax)
        R = hostlib.Ratings(n, m)
        R.read_train(tmp_path / "train.tsv")
        rp, c, v = R.csr()
        got.append((rp.copy(), c.copy(), v.copy(), R.seq2user().copy(), R.seq2item().copy()))
    for g in got[1:]:
        assert all(np.array_equal(a, b) for a, b in zip(got[0], g))


# ----------------------------------------------------------- stop rule -------
def test_parallel_matrix_writer_is_byte_identical(tmp_path, monkeypatch):
    # large matrices are formatted by several threads, a wave of row blocks at a
    # time; the file must be the serial writer's, byte for byte
    rng = np.random.default_rng(4)
    a = rng.gamma(0.3, 2.0, size=(30011, 83))
    a[::53, 7] = 0.0
    a[11, 3] = 1e-9
    a[12, 4] = 98765.4321
    ids = rng.permutation(10 * a.shape[0])[: a.shape[0]].astype(np.uint32)
    out = {}
    for nt in ("1", "3", "8"):
        monkeypatch.setenv("HGAPREC_SAVE_THREADS", nt)
        assert hostlib.save_matrix(tmp_path / f"m{nt}.tsv", a, ids) == 0
        out[nt] = (tmp_path / f"m{nt}.tsv").read_bytes()
    assert out["1"] == out["3"] == out["8"]
    lines = out["8"].decode().splitlines()
    assert len(lines) == a.shape[0]
    for r in (0, 11, 12, 53, a.shape[0] - 1):
        assert lines[r] == "\t".join([str(r), str(ids[r])] + ["%.8f" % v for v in a[r]])


def test_stop_rule():
    # hgaprec.cc:1476-1492: nothing before iter > 30; why=0 on a tiny relative gain,
    # why=1 after three consecutive decreases (nh > 2)
    it = np.arange(0, 100, 10)
    a = np.array([-3.0, -2.5, -2.2, -2.1, -2.0999999, -2.0, -1.9, -1.8, -1.7, -1.6])
    at, why = hostlib.stop_rule(it, a)
    assert at == 4 and why[4] == 0 and why[:4] == [-1] * 4
    a = np.array([-3.0, -2.0, -2.1, -2.2, -2.3, -2.4, -2.5, -2.6, -1.0, -1.0])
    at, why = hostlib.stop_rule(it, a)
    assert at == 6 and why[6] == 1              # decreases at iters 40, 50, 60 -> nh = 3
    a = np.array([-3.0, -2.0, -2.1, -2.2, -2.3, -2.25, -2.4, -2.5, -2.45, -2.6])
    at, why = hostlib.stop_rule(it, a)
    assert at == -1                              # an increase resets the counter


def test_fast_fixed8_formatter_equals_printf():
    """the TSV writers format "%.8f" themselves (error-free product + exact
    tie handling); it must agree with printf on every double, ties included"""
    rng = np.random.default_rng(11)
    vals = [0.0, 0.3, 1.0, 0.001953125, 0.005859375, 1.5e-9, 4.999999999e-9, 5e-9, 0.99999999, 0.999999995,
            0.9999999949999999, 12345.678901234567, 1e-30, 2.5e-9, 7.5e-9, 123456789.0, 1e15 + 0.5,
            9007199254740991.0, 1e16, 1e300, float("inf"), float("nan"), -0.5, -1e-12, 8191.00000001]
    vals += [k / 512.0 for k in range(1, 64, 2)]                      # exact ties at the 8th decimal
    vals += [(2 * k + 1) * 390625 / 2.0 ** 27 for k in range(40)]
    arr = np.concatenate([np.array(vals), rng.gamma(0.3, 1.0, 200000), rng.random(200000) * 1e-6,
                          rng.random(100000) * 1e5, np.exp(rng.normal(0, 8, 200000)),
                          np.round(rng.random(100000), 8) + rng.integers(-3, 4, 100000) * 2.0 ** -60])
    got = hostlib.format_fixed8(arr)
    assert len(got) == arr.size
    for v, g in zip(arr.tolist(), got):
        assert g == "%.8f" % v, (v.hex() if v == v else v, g, "%.8f" % v)
