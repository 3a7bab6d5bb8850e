"""-m gpu: the `hgaprec` CLI end to end (TSV in -> output directory) against
the oracle's end-to-end run on the same files.  Comparison rule of SURVEY.md
8(c): factor TSVs |a-b| <= 1e-4*|b| + 5e-9 (the 5e-9 absorbs %.8f rounding),
LL series abs diff <= 1e-6, integer columns exact, seconds column ignored."""
import os
import subprocess
from pathlib import Path

import json

import numpy as np
import pytest

from tests.util import make_problem

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
EXE = ROOT / "hgaprec_amd" / "hgaprec"


def write_dataset(d, n, m, nnz, seed, **problem_kw):
    rng = np.random.default_rng(seed)
    rowptr, col, val = make_problem(n, m, nnz, seed, **problem_kw)
    uid = rng.permutation(10 * n)[:n] + 1            # raw ids, not in seq order
    iid = rng.permutation(10 * m)[:m] + 1
    u = np.repeat(np.arange(n), np.diff(rowptr))
    order = rng.permutation(u.size)                   # shuffle the file: seq ids = first-seen order
    split = rng.random(u.size)
    d.mkdir(parents=True, exist_ok=True)
    with open(d / "train.tsv", "w") as ft, open(d / "validation.tsv", "w") as fv, open(d / "test.tsv", "w") as fs:
        for j in order:
            line = f"{uid[u[j]]}\t{iid[col[j]]}\t{val[j]}\n"
            (fv if split[j] < 0.05 else fs if split[j] < 0.15 else ft).write(line)


def read_tsv(p):
    rows = [l.split("\t") for l in Path(p).read_text().splitlines()]
    ints = np.array([[int(r[0]), int(r[1])] for r in rows])
    vals = np.array([[float(x) for x in r[2:]] for r in rows])
    return ints, vals


def series(p):
    out = []
    for l in Path(p).read_text().splitlines():
        a = l.split("\t")
        out.append((int(a[0]), float(a[2]), int(a[3])))
    return out


@pytest.mark.parametrize("flags,K,maxit", [
    (["-hier"], 5, 20),
    (["-hier", "-bias"], 8, 12),
    (["-hier", "-binary-data", "-rating-threshold", "3"], 5, 12),
    ([], 5, None),                 # vb(): runs until the stop rule fires
    (["-bias"], 5, None),          # vb_bias()
    (["-bias", "-novb"], 5, None), # vb_bias(), -novb: both rates from the previous iteration (hgaprec.cc:1276-1297)
    (["-hier", "-novb"], 5, 8),    # vb_hier() never reads the flag; only the directory name changes
    (["-hier"], 100, 6),               # K = 100: rows of W packed at 59 bits (the library's default there)
    (["-hier", "-plain-rows"], 100, 4),  # ... and kept plain (extension flag, hpf_config.w_storage = 3)
    (["-hier", "-no-tiles"], 100, 4),    # extension flag, hpf_config.tiling = 1 (this matrix is row-major either way)
    (["-hier", "-bias", "-logl"], 6, 8),
    (["-logl"], 4, None),
    (["-hier", "-rfreq", "50"], 5, 100),      # iteration 100: ranking.tsv + itemrank.tsv + meanrank.txt
])
def test_cli_matches_oracle_run(orc, tmp_path, flags, K, maxit):
    n, m = 300, 200
    data = tmp_path / "data"
    write_dataset(data, n, m, 9000, seed=17)
    hier, bias, binary = "-hier" in flags, "-bias" in flags, "-binary-data" in flags
    logl, novb = "-logl" in flags, "-novb" in flags
    thr = 3 if binary else 1
    rfreq = 2 if hier else 10
    if "-rfreq" in flags:
        rfreq = int(flags[flags.index("-rfreq") + 1])
        flags = [f for k, f in enumerate(flags) if f != "-rfreq" and (k == 0 or flags[k - 1] != "-rfreq")]
    if not hier:
        # the batch loops without -hier end through the stop rule -> do_on_stop ->
        # gen_ranking_for_users, which needs <dir>/test_users.tsv (raw user ids)
        ids = sorted({int(l.split("\t")[0]) for l in (data / "test.tsv").read_text().splitlines()})
        (data / "test_users.tsv").write_text("".join(f"{u}\n" for u in ids[:60]) + "999999999\n")
    args = ["-dir", str(data), "-n", str(n), "-m", str(m), "-k", str(K), "-seed", "7", "-rfreq", str(rfreq)] + flags
    if maxit is not None:
        args += ["-max-iterations", str(maxit)]
    r = subprocess.run([str(EXE)] + args, cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    outs = [p for p in tmp_path.iterdir() if p.is_dir() and p.name.startswith(f"n{n}-m{m}-k{K}")]
    assert len(outs) == 1
    out = outs[0]
    exp = "-".join(x for x in [f"n{n}-m{m}-k{K}", "batch", "bin" if binary else "", "bias" if bias else "",
                               "hier" if hier else "", "" if novb else "vb", "seed7"] if x)
    assert out.name == exp

    ref = tmp_path / "oracle_out"
    ref.mkdir()
    last = orc.run(data, ref, n, m, K, hier=hier, bias=bias, binary=binary, rating_threshold=thr,
                   rfreq=rfreq, max_iterations=maxit if maxit is not None else 1000, seed=7, logl=logl, novb=novb)
    assert last >= 0

    for f in ("byusers.tsv", "byitems.tsv"):
        assert (out / f).read_text() == (ref / f).read_text()
    for f in ("validation.txt", "test.txt"):
        a, b = series(out / f), series(ref / f)
        assert [x[0] for x in a] == [x[0] for x in b], f          # same report iterations (same stop)
        assert [x[2] for x in a] == [x[2] for x in b]
        assert max(abs(x[1] - y[1]) for x, y in zip(a, b)) <= 1e-6
    # ranking evaluation (compute_precision / compute_itemrank / gen_ranking_for_users)
    assert (out / "precision.txt").read_text() == (ref / "precision.txt").read_text()
    assert len((out / "precision.txt").read_text().splitlines()) >= 2
    for f in ("ranking.tsv", "itemrank.tsv", "meanrank.txt"):
        assert (out / f).exists() == (ref / f).exists(), f
        if (ref / f).exists():
            assert (out / f).read_text() == (ref / f).read_text(), f
    if not hier or maxit == 100:
        assert (out / "ranking.tsv").exists() and (out / "meanrank.txt").exists()
    la = [float(x) for x in (out / "logl.txt").read_text().split()]
    lb = [float(x) for x in (ref / "logl.txt").read_text().split()]
    assert len(la) == len(lb) and (len(la) > 0) == logl
    assert all(abs(x - y) <= 2e-5 + 1e-10 * abs(y) for x, y in zip(la, lb))
    ma, mb = (out / "max.txt").read_text().split("\t"), (ref / "max.txt").read_text().split("\t")
    assert ma[0] == mb[0] and ma[3] == mb[3] and abs(float(ma[2]) - float(mb[2])) <= 2e-5

    names = (["hbeta", "htheta", "betarate", "thetarate"] if hier else ["beta", "theta"])
    if bias:
        names += ["betabias", "thetabias"]
    for nm in names:
        for suf in ("", "_shape", "_rate"):
            ia, va = read_tsv(out / f"{nm}{suf}.tsv")
            ib, vb = read_tsv(ref / f"{nm}{suf}.tsv")
            assert np.array_equal(ia, ib), nm + suf
            assert np.all(np.abs(va - vb) <= 1e-4 * np.abs(vb) + 5e-9), nm + suf
            assert np.max(np.abs(va - vb)) <= 2.1e-8 + 1e-9 * np.max(np.abs(vb)), nm + suf   # in practice: print rounding only
    # files the reference's constructor creates even when they stay empty
    for f in ("heldout.txt", "logl.txt", "ndcg.txt", "rmse.txt", "infer.log", "param.txt"):
        assert (out / f).exists()
    keys = [l.split(":")[0] for l in (out / "param.txt").read_text().splitlines()]
    assert keys[:20] == ["n", "k", "t", "test_ratio", "validation_ratio", "seed", "a", "b", "c", "d",
                         "reportfreq", "vb", "bias", "hier", "nmf", "lda", "wals_l", "wals_C",
                         "mle_user", "mle_item"]
    assert keys[20:27] == ["training ratings", "post pruning nusers", "post pruning nitems", "statistics",
                           "infer n", "test ratings", "validation ratings"]


@pytest.mark.parametrize("flags,K,maxit", [
    (["-hier", "-bias", "-logl"], 6, 12),
    ([], 5, None),                 # vb(): stop rule -> do_on_stop -> gen_ranking_for_users on every rank
    (["-hier", "-rfreq", "50"], 5, 100),
    (["-bias", "-novb"], 5, None), # vb_bias()'s else-branch across ranks: the start state's sum_u E[theta] is reduced once (round 4)
])
def test_two_process_cli_matches_oracle(orc, tmp_path, flags, K, maxit):
    """`-ngpus 2`: two processes (here both on GPU 0, all-reduce staged through
    the host: `-comm host`), users sharded by nnz; the output directory must be
    what the single-process reference semantics give."""
    n, m = 300, 200
    data = tmp_path / "data"
    write_dataset(data, n, m, 9000, seed=17)
    hier, bias = "-hier" in flags, "-bias" in flags
    logl, novb = "-logl" in flags, "-novb" in flags
    rfreq = 2 if hier else 10
    if "-rfreq" in flags:
        rfreq = int(flags[flags.index("-rfreq") + 1])
        flags = [f for k, f in enumerate(flags) if f != "-rfreq" and (k == 0 or flags[k - 1] != "-rfreq")]
    if not hier:
        ids = sorted({int(l.split("\t")[0]) for l in (data / "test.tsv").read_text().splitlines()})
        (data / "test_users.tsv").write_text("".join(f"{u}\n" for u in ids[:60]))
    args = ["-dir", str(data), "-n", str(n), "-m", str(m), "-k", str(K), "-seed", "7", "-rfreq", str(rfreq)] + flags
    if maxit is not None:
        args += ["-max-iterations", str(maxit)]
    r = subprocess.run([str(EXE)] + args + ["-ngpus", "2", "-device", "0", "-comm", "host"], cwd=tmp_path,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    outs = [p for p in tmp_path.iterdir() if p.is_dir() and p.name.startswith(f"n{n}-m{m}-k{K}")]
    assert len(outs) == 1
    out = outs[0]
    assert not [p for p in out.iterdir() if ".part" in p.name]          # all part files merged
    # rank 0 alone parsed the TSVs; rank 1 took the parsed data set from its image, which is gone again
    assert "[rank 1] ratings handed over by rank 0" in r.stderr and not (out / "ranks.cache.bin").exists()
    ref = tmp_path / "oracle_out"
    ref.mkdir()
    orc.run(data, ref, n, m, K, hier=hier, bias=bias, rfreq=rfreq,
            max_iterations=maxit if maxit is not None else 1000, seed=7, logl=logl, novb=novb)
    for f in ("validation.txt", "test.txt"):
        a, b = series(out / f), series(ref / f)
        assert [x[0] for x in a] == [x[0] for x in b] and [x[2] for x in a] == [x[2] for x in b]
        assert max(abs(x[1] - y[1]) for x, y in zip(a, b)) <= 1e-6
    assert (out / "precision.txt").read_text() == (ref / "precision.txt").read_text()
    for f in ("ranking.tsv", "itemrank.tsv", "meanrank.txt"):
        assert (out / f).exists() == (ref / f).exists(), f
        if (ref / f).exists():
            assert (out / f).read_text() == (ref / f).read_text(), f
    la = [float(x) for x in (out / "logl.txt").read_text().split()]
    lb = [float(x) for x in (ref / "logl.txt").read_text().split()]
    assert len(la) == len(lb) and all(abs(x - y) <= 2e-5 + 1e-10 * abs(y) for x, y in zip(la, lb))
    names = (["hbeta", "htheta", "betarate", "thetarate"] if hier else ["beta", "theta"])
    if bias:
        names += ["betabias", "thetabias"]
    for nm in names:
        for suf in ("", "_shape", "_rate"):
            ia, va = read_tsv(out / f"{nm}{suf}.tsv")
            ib, vb = read_tsv(ref / f"{nm}{suf}.tsv")
            assert np.array_equal(ia, ib), nm + suf
            assert np.max(np.abs(va - vb)) <= 2.1e-8 + 1e-9 * np.max(np.abs(vb)), nm + suf


def test_checkpoint_and_resume(tmp_path):
    """-checkpoint N / -resume (an extension: the reference cannot resume).  A
    run stopped after 6 iterations and resumed to 14 must end where the
    uninterrupted run ends -- bit for bit: the checkpoint carries the loop's
    device arrays verbatim (hpf_snapshot_save) -- and append to the same report
    files."""
    n, m, K = 300, 200, 6
    data = tmp_path / "data"
    write_dataset(data, n, m, 9000, seed=17)
    base = ["-dir", str(data), "-n", str(n), "-m", str(m), "-k", str(K), "-seed", "3", "-rfreq", "2", "-hier", "-bias"]
    a, b = tmp_path / "straight", tmp_path / "resumed"
    a.mkdir(); b.mkdir()
    r = subprocess.run([str(EXE)] + base + ["-max-iterations", "14"], cwd=a, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    r = subprocess.run([str(EXE)] + base + ["-max-iterations", "6", "-checkpoint", "6"], cwd=b, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    outb = [p for p in b.iterdir() if p.is_dir()][0]
    assert (outb / "checkpoint.r0of1.bin").exists()
    assert len(series(outb / "validation.txt")) == 4                      # iterations 0, 2, 4, 6
    r = subprocess.run([str(EXE)] + base + ["-max-iterations", "14", "-resume"], cwd=b, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    outa = [p for p in a.iterdir() if p.is_dir()][0]
    sa, sb = series(outa / "validation.txt"), series(outb / "validation.txt")
    assert [x[0] for x in sa] == [x[0] for x in sb] == list(range(0, 15, 2))
    assert [x[1] for x in sa] == [x[1] for x in sb]                       # the %.9f series, digit for digit
    assert (outa / "precision.txt").read_text() == (outb / "precision.txt").read_text()   # same RNG stream
    for nm in ("htheta", "hbeta", "thetarate", "betarate", "thetabias", "betabias"):
        for suf in ("", "_shape", "_rate"):
            assert (outa / f"{nm}{suf}.tsv").read_text() == (outb / f"{nm}{suf}.tsv").read_text(), nm + suf
    # a checkpoint of another configuration is refused
    r = subprocess.run([str(EXE)] + base[:-1] + ["-max-iterations", "14", "-resume"], cwd=b, capture_output=True, text=True)
    assert r.returncode != 0


def test_checkpoint_and_resume_two_ranks(tmp_path):
    """the same with users sharded over two processes (`-ngpus 2 -comm host`, both on GPU
    0): every rank writes and reloads its own snapshot; the merged output files of the
    resumed run are the uninterrupted run's, text for text."""
    n, m, K = 300, 200, 6
    data = tmp_path / "data"
    write_dataset(data, n, m, 9000, seed=17)
    base = ["-dir", str(data), "-n", str(n), "-m", str(m), "-k", str(K), "-seed", "3", "-rfreq", "2", "-hier",
            "-ngpus", "2", "-device", "0", "-comm", "host"]
    a, b = tmp_path / "straight", tmp_path / "resumed"
    a.mkdir(); b.mkdir()
    r = subprocess.run([str(EXE)] + base + ["-max-iterations", "10"], cwd=a, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    r = subprocess.run([str(EXE)] + base + ["-max-iterations", "4", "-checkpoint", "4"], cwd=b, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    outb = [p for p in b.iterdir() if p.is_dir()][0]
    assert (outb / "checkpoint.r0of2.bin").exists() and (outb / "checkpoint.r1of2.bin").exists()
    r = subprocess.run([str(EXE)] + base + ["-max-iterations", "10", "-resume"], cwd=b, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    outa = [p for p in a.iterdir() if p.is_dir()][0]
    sa, sb = series(outa / "validation.txt"), series(outb / "validation.txt")
    assert [(x[0], x[1], x[2]) for x in sa] == [(x[0], x[1], x[2]) for x in sb]
    for nm in ("htheta", "hbeta", "thetarate", "betarate"):
        for suf in ("", "_shape", "_rate"):
            assert (outa / f"{nm}{suf}.tsv").read_text() == (outb / f"{nm}{suf}.tsv").read_text(), nm + suf
    assert (outa / "precision.txt").read_text() == (outb / "precision.txt").read_text()


def test_resume_of_vb_bias_novb_on_two_ranks_keeps_the_reduced_start_sums(tmp_path):
    """`-bias -novb` without `-hier` on two ranks, `-comm host`.  The snapshot a rank reloads carries
    the ALL-REDUCED sum_u E[theta] in the tail of its exchange buffer; reducing that tail once more on resume made the
    first resumed item rate world x too large.  hpf_work_info.start_sums_pending (ABI v7) now says whether the tail
    still has to be reduced.  vb_bias() has no -max-iterations (hgaprec.cc:1219-1319): the first run goes to its stop
    rule and leaves its last checkpoint (every 7 iterations) a few iterations before the end; the resumed run must
    stop at the same iteration with the same files."""
    n, m, K = 300, 200, 5
    data = tmp_path / "data"
    write_dataset(data, n, m, 9000, seed=17)
    base = ["-dir", str(data), "-n", str(n), "-m", str(m), "-k", str(K), "-seed", "7", "-rfreq", "10", "-bias", "-novb",
            "-ngpus", "2", "-device", "0", "-comm", "host"]
    a, b = tmp_path / "straight", tmp_path / "resumed"
    a.mkdir(); b.mkdir()
    r = subprocess.run([str(EXE)] + base, cwd=a, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-1500:]
    r = subprocess.run([str(EXE)] + base + ["-checkpoint", "7"], cwd=b, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-1500:]
    outa = [p for p in a.iterdir() if p.is_dir()][0]
    outb = [p for p in b.iterdir() if p.is_dir()][0]
    assert (outb / "checkpoint.r0of2.bin").exists() and (outb / "checkpoint.r1of2.bin").exists()
    sa = series(outa / "validation.txt")
    stop_iter = sa[-1][0]
    assert stop_iter >= 40 and stop_iter % 7 != 0          # so the checkpoint is older than the last iteration
    before = series(outb / "validation.txt")
    assert before == sa
    r = subprocess.run([str(EXE)] + base + ["-checkpoint", "7", "-resume"], cwd=b, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-1500:]
    said = r.stderr + (outb / "infer.log").read_text()
    assert "resumed from" in said and f"at iteration {stop_iter // 7 * 7 + 1}" in said
    sb = series(outb / "validation.txt")
    # the resumed run appended the report steps it repeated: they must be the straight run's, digit for digit
    extra = sb[len(before):]
    assert extra and extra[-1][0] == stop_iter
    tail = {x[0]: x for x in sa}
    assert all((x[0], x[1], x[2]) == (tail[x[0]][0], tail[x[0]][1], tail[x[0]][2]) for x in extra)
    for nm in ("theta", "beta", "thetabias", "betabias"):
        for suf in ("", "_shape", "_rate"):
            assert (outa / f"{nm}{suf}.tsv").read_text() == (outb / f"{nm}{suf}.tsv").read_text(), nm + suf


def test_dataset_cache_runs_are_identical(tmp_path):
    """-cache (extension): first run parses the TSVs and writes the binary image,
    the second loads it; every output file must be byte-identical (same CSR,
    same id maps, same held-out sets => same everything), also with 2 ranks."""
    n, m, K = 300, 200, 6
    data = tmp_path / "data"
    write_dataset(data, n, m, 9000, seed=23)
    ids = sorted({int(l.split("\t")[0]) for l in (data / "test.tsv").read_text().splitlines()})
    (data / "test_users.tsv").write_text("".join(f"{u}\n" for u in ids[:40]))
    base = ["-dir", str(data), "-n", str(n), "-m", str(m), "-k", str(K), "-seed", "5", "-rfreq", "5", "-hier",
            "-max-iterations", "10"]
    outs = []
    for tag, extra in (("plain", []), ("write", ["-cache"]), ("load", ["-cache"]),
                       ("load2", ["-cache", "-ngpus", "2", "-device", "0", "-comm", "host"])):
        d = tmp_path / tag
        d.mkdir()
        r = subprocess.run([str(EXE)] + base + extra, cwd=d, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-1500:]
        out = [p for p in d.iterdir() if p.is_dir()][0]
        log = (out / "infer.log").read_text()
        if tag == "write":
            assert "-cache: wrote" in log and (data / "hgaprec.cache.bin").exists()
        if tag.startswith("load"):
            assert "-cache: loaded" in log and "-cache: wrote" not in log
        outs.append(out)
    names = ["validation.txt", "test.txt", "byusers.tsv", "byitems.tsv", "precision.txt", "htheta.tsv", "hbeta.tsv",
             "htheta_shape.tsv", "hbeta_rate.tsv", "thetarate.tsv", "betarate_rate.tsv"]
    for nm in names:
        ref = (outs[0] / nm).read_text().splitlines()
        for o in outs[1:3]:
            got = (o / nm).read_text().splitlines()
            if nm in ("validation.txt", "test.txt"):          # column 2 is wall-clock seconds
                strip = lambda ls: [l.split("\t")[:1] + l.split("\t")[2:] for l in ls]
                assert strip(got) == strip(ref), nm
            else:
                assert got == ref, nm
    # two ranks on the cached dataset: same integers, values within the sharded-sum tolerance
    for nm in ("htheta.tsv", "hbeta.tsv"):
        ia, va = read_tsv(outs[0] / nm)
        ib, vb = read_tsv(outs[3] / nm)
        assert np.array_equal(ia, ib) and np.max(np.abs(va - vb)) <= 2.1e-8 + 1e-9 * np.max(np.abs(va))
    # two ranks, -cache and NO image yet: whether the image is there is one decision for the whole
    # job -- a rank that saw the image rank 0 had just written would skip the hand-over's collectives and leave
    # the ranks out of step (a hang, or mixed values).  Rank 0 parses, hands over, and writes the image afterwards;
    # the run must end and give the bits of the two-rank run that loaded the image.
    import shutil
    data2 = tmp_path / "data2"
    shutil.copytree(data, data2)
    (data2 / "hgaprec.cache.bin").unlink()
    d = tmp_path / "write2"
    d.mkdir()
    base2 = [str(data2) if a == str(data) else a for a in base]
    r = subprocess.run([str(EXE)] + base2 + ["-cache", "-ngpus", "2", "-device", "0", "-comm", "host"], cwd=d,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    out = [p for p in d.iterdir() if p.is_dir()][0]
    log = (out / "infer.log").read_text()
    assert "-cache: wrote" in log and "-cache: loaded" not in log and (data2 / "hgaprec.cache.bin").exists()
    assert "[rank 1] ratings handed over by rank 0" in r.stderr
    for nm in ("htheta.tsv", "hbeta.tsv", "thetarate.tsv", "precision.txt"):
        assert (out / nm).read_text() == (outs[3] / nm).read_text(), nm


def write_c1(tmp_path):
    """BASELINE config C1 as train / validation / test.tsv under tmp_path/ml -> (dir, n, m, K)"""
    import torch
    from hgaprec_amd import synth
    cfg = synth.CONFIGS["C1"]
    n, m, K = cfg["n"], cfg["m"], cfg["K"]
    rowptr, col, val = synth.generate(n, m, 1_000_209, cfg["alpha_u"], cfg["alpha_i"], seed=cfg["seed"],
                                      device="cuda" if torch.cuda.is_available() else "cpu")
    rng = np.random.default_rng(cfg["seed"])
    uid = rng.permutation(10 * n)[:n] + 1
    iid = rng.permutation(10 * m)[:m] + 1
    u = np.repeat(np.arange(n), np.diff(rowptr))
    split = rng.random(u.size)
    order = rng.permutation(u.size)
    data = tmp_path / "ml"
    data.mkdir()
    lines = np.char.add(np.char.add(np.char.add(uid[u].astype(str), "\t"), np.char.add(iid[col].astype(str), "\t")),
                        np.char.add(val.astype(str), "\n"))
    for name, sel in (("train.tsv", split[order] >= 0.20), ("validation.tsv", split[order] < 0.01),
                      ("test.tsv", (split[order] >= 0.01) & (split[order] < 0.20))):
        (data / name).write_text("".join(lines[order][sel].tolist()))
    return data, n, m, K


def test_w48_opt_in_runs_c1_to_the_same_stop(tmp_path):
    """`hgaprec -w48` ("opt-in perf mode with its own measured tolerance"): W kept
    as the top 48 bits of its fp64 value (hpf_config.w_storage = 2; arithmetic, sums and every exported number stay fp64).
    Never the default, never the headline.  C1 through the TSV path, -hier, run to the STOP RULE (hgaprec.cc:1439-1501) with
    and without the flag: the same stop iteration, the held-out series within 1e-6, every factor file within the rule of
    SURVEY.md 8(c), 1e-4 |b| + 5e-9 -- a convergence-length statement, where round 4 had a 150-sweep drift figure."""
    data, n, m, K = write_c1(tmp_path)
    args = ["-dir", str(data), "-n", str(n), "-m", str(m), "-k", str(K), "-hier", "-rfreq", "10"]
    outs = {}
    for tag, extra in (("exact", []), ("w48", ["-w48"])):
        d = tmp_path / tag
        d.mkdir()
        r = subprocess.run([str(EXE)] + args + extra, cwd=d, capture_output=True, text=True, timeout=1800)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = d / f"n{n}-m{m}-k{K}-batch-hier-vb"
        assert outs[tag].is_dir()
    log48 = (outs["w48"] / "infer.log").read_text()
    assert "48-bit (opt-in)" in log48 and "48-bit" not in (outs["exact"] / "infer.log").read_text()
    for f in ("validation.txt", "test.txt"):
        a, b = series(outs["w48"] / f), series(outs["exact"] / f)
        assert [x[0] for x in a] == [x[0] for x in b], f                  # the same report steps: the same stop iteration
        assert [x[2] for x in a] == [x[2] for x in b]
        assert max(abs(x[1] - y[1]) for x, y in zip(a, b)) <= 1e-6, f
    stop = series(outs["exact"] / "validation.txt")[-1][0]
    assert stop >= 40                                                      # the stop rule cannot fire before (A.4)
    ma, mb = (outs["w48"] / "max.txt").read_text().split("\t"), (outs["exact"] / "max.txt").read_text().split("\t")
    assert ma[0] == mb[0] and ma[3] == mb[3] and abs(float(ma[2]) - float(mb[2])) <= 1e-5       # iteration, why, LL (%.5f)
    worst = 0.0
    for nm in ("hbeta", "htheta", "betarate", "thetarate"):
        for suf in ("", "_shape", "_rate"):
            ia, va = read_tsv(outs["w48"] / f"{nm}{suf}.tsv")
            ib, vb = read_tsv(outs["exact"] / f"{nm}{suf}.tsv")
            assert np.array_equal(ia, ib), nm + suf
            assert np.all(np.abs(va - vb) <= 1e-4 * np.abs(vb) + 5e-9), nm + suf
            worst = max(worst, float(np.max(np.abs(va - vb) / (np.abs(vb) + 1e-4))))
    assert (outs["w48"] / "precision.txt").read_text() == (outs["exact"] / "precision.txt").read_text()
    print(f"-w48 on C1: stop at iteration {stop} both ways; worst |a-b| / (|b| + 1e-4) over the factor files {worst:.2e}")


def test_c1_movielens_shaped_end_to_end(orc, tmp_path):
    """BASELINE config C1: the MovieLens-1M stand-in (the real example/ tarball is
    not in the reference mount): 6040 x 3681, ~1.0M ratings split 80/1/19,
    K=20, -hier, through the TSV path, CLI vs the oracle's end-to-end run."""
    data, n, m, K = write_c1(tmp_path)
    args = ["-dir", str(data), "-n", str(n), "-m", str(m), "-k", str(K), "-hier", "-rfreq", "5", "-max-iterations", "10"]
    r = subprocess.run([str(EXE)] + args, cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = tmp_path / f"n{n}-m{m}-k{K}-batch-hier-vb"          # -dir is an absolute path: no 2-letter tag
    assert out.is_dir()
    ref = tmp_path / "oracle_out"
    ref.mkdir()
    orc.run(data, ref, n, m, K, hier=True, rfreq=5, max_iterations=10, seed=0)
    for f in ("byusers.tsv", "byitems.tsv", "precision.txt"):
        assert (out / f).read_text() == (ref / f).read_text(), f
    for f in ("validation.txt", "test.txt"):
        a, b = series(out / f), series(ref / f)
        assert [x[0] for x in a] == [x[0] for x in b] == [0, 5, 10] and [x[2] for x in a] == [x[2] for x in b]
        assert max(abs(x[1] - y[1]) for x, y in zip(a, b)) <= 1e-6
    for nm in ("hbeta", "htheta", "betarate", "thetarate"):
        for suf in ("", "_shape", "_rate"):
            ia, va = read_tsv(out / f"{nm}{suf}.tsv")
            ib, vb = read_tsv(ref / f"{nm}{suf}.tsv")
            assert np.array_equal(ia, ib), nm + suf
            assert np.all(np.abs(va - vb) <= 1e-4 * np.abs(vb) + 5e-9), nm + suf
            assert np.max(np.abs(va - vb)) <= 2.1e-8 + 1e-9 * np.max(np.abs(vb)), nm + suf
    # the host side of that run was threaded (an 11 MB train.tsv is parsed in pieces, the CSR built per user range,
    # the start state's expectations and the factor files formatted on all threads): one thread gives the same files
    solo = tmp_path / "solo"
    solo.mkdir()
    r = subprocess.run([str(EXE)] + args, cwd=solo, capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, HGAPREC_READ_THREADS="1", HGAPREC_SAVE_THREADS="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    skip = {"infer.log", "validation.txt", "test.txt", "max.txt"}        # these carry wall-clock seconds
    names = sorted(p.name for p in out.iterdir() if p.name not in skip)
    assert names == sorted(p.name for p in (solo / out.name).iterdir() if p.name not in skip)
    for nm in names:
        assert (out / nm).read_bytes() == (solo / out.name / nm).read_bytes(), nm
    assert [x[1] for x in series(out / "validation.txt")] == [x[1] for x in series(solo / out.name / "validation.txt")]


def test_sigterm_saves_state_on_the_following_iterations(tmp_path):
    """main.cc:19-30 + hgaprec.cc:1430-1433: SIGTERM only sets save_state_now; from then on
    EVERY iteration logs "Saving state at iteration ..." and runs do_on_stop() -- save_model
    plus gen_ranking_for_users -- and the run goes on (the reference never exits on the signal)."""
    import re
    import signal
    import time
    n, m, K = 300, 200, 5
    data = tmp_path / "data"
    write_dataset(data, n, m, 9000, seed=17)
    ids = sorted({int(l.split("\t")[0]) for l in (data / "test.tsv").read_text().splitlines()})
    (data / "test_users.tsv").write_text("".join(f"{u}\n" for u in ids[:40]))
    args = ["-dir", str(data), "-n", str(n), "-m", str(m), "-k", str(K), "-seed", "7", "-hier",
            "-rfreq", "100000000", "-max-iterations", "1000000000"]
    so = open(tmp_path / "stdout.txt", "wb")
    p = subprocess.Popen([str(EXE)] + args, cwd=tmp_path, stdout=so, stderr=subprocess.DEVNULL)
    try:
        out = tmp_path / f"n{n}-m{m}-k{K}-batch-hier-vb-seed7"

        def wait_for(cond, what, secs=180):
            t0 = time.time()
            while not cond():
                assert p.poll() is None, f"hgaprec exited ({p.returncode}) while waiting for {what}"
                assert time.time() - t0 < secs, f"timed out waiting for {what}"
                time.sleep(0.05)

        def last_iteration():
            txt = (tmp_path / "stdout.txt").read_bytes()[-200:].decode(errors="replace")
            its = re.findall(r"iteration (\d+)", txt)
            return int(its[-1]) if its else -1

        # iteration 0 is a report step (0 % rfreq == 0): it writes the model once, no ranking
        wait_for(lambda: (out / "htheta.tsv").exists() and last_iteration() >= 20, "the run to get going")
        before = (out / "htheta.tsv").read_text()
        assert not (out / "ranking.tsv").exists()
        assert "Saving state" not in (out / "infer.log").read_text()
        p.send_signal(signal.SIGTERM)
        saves = lambda: re.findall(r"Saving state at iteration (\d+)", (out / "infer.log").read_text())
        wait_for(lambda: len(saves()) >= 3, "three post-signal saves")
        its = [int(x) for x in saves()[:3]]
        assert its[1] == its[0] + 1 and its[2] == its[1] + 1          # every following iteration
        assert p.poll() is None                                       # and the run goes on
        assert (out / "ranking.tsv").exists()                         # gen_ranking_for_users ran
        got = {}

        def whole_file():          # the file is being rewritten every iteration now: take a complete one
            t = (out / "htheta.tsv").read_text()
            if t.endswith("\n") and len(t.splitlines()) == n and all(len(l.split("\t")) == K + 2 for l in t.splitlines()):
                got["t"] = t
            return "t" in got
        wait_for(whole_file, "a complete htheta.tsv")
        assert got["t"] != before                                     # save_model ran on the current state
    finally:
        p.kill()
        p.wait()
        so.close()


def _full_cases():
    root = ROOT / "tests" / "golden" / "full"
    return sorted(p.name for p in root.iterdir() if (p / "case.json").exists()) if root.exists() else []


@pytest.mark.parametrize("case", _full_cases() or [None])
def test_cli_matches_reference_fixtures(tmp_path, case):
    """End-to-end against outputs of the REFERENCE BINARY itself:
    tests/golden/full/ is produced by `make -C oracle ref-full && python
    tests/golden/make_golden.py --full` on a machine with a genuine GSL.  This image has
    none, so no such fixture exists yet and the test skips -- end-to-end parity is
    "unpinned" until it does."""
    if case is None:
        pytest.skip("no tests/golden/full/: the whole reference needs GSL, which this image lacks")
    cdir = ROOT / "tests" / "golden" / "full" / case
    spec = json.loads((cdir / "case.json").read_text())
    args = [str(cdir / "data") if a == "DATA" else a for a in spec["args"]]
    r = subprocess.run([str(EXE)] + args, cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = tmp_path / spec["prefix"]
    assert out.is_dir(), [p.name for p in tmp_path.iterdir()]
    for name, ref in spec["files"].items():
        got = (out / name).read_text().splitlines()
        assert len(got) == ref["rows"], name
        got = got[::ref["every"]]
        if name in ("byusers.tsv", "byitems.tsv", "precision.txt", "ranking.tsv", "itemrank.tsv", "meanrank.txt"):
            assert got == ref["lines"], name
            continue
        if name in ("param.txt", "infer.log", "heldout.txt"):
            continue
        for a, b in zip(got, ref["lines"]):
            fa, fb = a.split("\t"), b.split("\t")
            assert len(fa) == len(fb), name
            for col_no, (x, y) in enumerate(zip(fa, fb)):
                if name in ("validation.txt", "test.txt", "max.txt") and col_no == 1:
                    continue                                      # wall-clock seconds
                try:
                    xv, yv = float(x), float(y)
                except ValueError:
                    assert x == y, name
                    continue
                assert abs(xv - yv) <= 1e-4 * abs(yv) + 2.1e-8, (name, a, b)


def test_eight_process_cli_with_a_one_user_rank_equals_the_one_process_run(tmp_path):
    """`hgaprec -ngpus 8 -comm host` (eight real processes on GPU 0) on a C4-shaped job -- `-hier -bias`,
    capacities -n 3000 -m 200 -- whose nnz-balanced cut leaves one rank a SINGLE user (a user holding an eighth of the
    ratings, sitting on a cut): the output directory must be the one-process run's under the comparison rule of SURVEY.md
    8(c) (factors 1e-4 |b| + 5e-9, LL series 1e-6, integer columns exact), and no part file may outlive the run."""
    import re
    rng = np.random.default_rng(3)
    m, singles, heavy_at = 200, 800, 375
    data = tmp_path / "data"
    data.mkdir()
    uid = rng.permutation(50000)[: singles + 1] + 1
    iid = rng.permutation(5000)[:m] + 1
    with open(data / "train.tsv", "w") as ft, open(data / "validation.tsv", "w") as fv, open(data / "test.tsv", "w") as fs:
        seq = 0
        for k in range(singles + 1):
            if k == heavy_at:                              # seq 375: 200 ratings = an eighth of the 1000 training ratings
                for i in rng.permutation(m):
                    ft.write(f"{uid[k]}\t{iid[i]}\t{1 + int(rng.integers(5))}\n")
                continue
            a, b = rng.choice(m, 2, replace=False)
            ft.write(f"{uid[k]}\t{iid[a]}\t{1 + int(rng.integers(5))}\n")
            (fv if k % 3 == 0 else fs).write(f"{uid[k]}\t{iid[b]}\t{1 + int(rng.integers(5))}\n")
            seq += 1
    args = ["-dir", str(data), "-n", "3000", "-m", "200", "-k", "8", "-hier", "-bias", "-seed", "5", "-rfreq", "3", "-max-iterations", "9"]
    outs = {}
    for world in (1, 8):
        wd = tmp_path / f"w{world}"
        wd.mkdir()
        r = subprocess.run([str(EXE)] + args + (["-ngpus", "8", "-device", "0", "-comm", "host"] if world > 1 else []), cwd=wd,
                           capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
        ds = [p for p in wd.iterdir() if p.is_dir() and p.name.startswith("n3000-m200-k8")]
        assert len(ds) == 1
        outs[world] = ds[0]
        if world == 8:
            ranges = {int(x[0]): (int(x[1]), int(x[2]), int(x[3])) for x in
                      re.findall(r"\[rank (\d)\] users \[(\d+), (\d+)\) of \d+: (\d+) of \d+ ratings", r.stderr)}
            assert sorted(ranges) == list(range(8)), r.stderr[-2000:]
            assert any(b - a == 1 for a, b, _ in ranges.values()), ranges               # the rank with one user
            assert sum(z for _, _, z in ranges.values()) == 1000 and ranges[0][0] == 0 and ranges[7][1] == singles + 1
            assert all(ranges[k][1] == ranges[k + 1][0] for k in range(7))
            leftovers = [p.name for p in ds[0].iterdir() if ".part" in p.name or p.name.endswith(".writing")]
            assert not leftovers, leftovers
    one, eight = outs[1], outs[8]
    names = sorted(p.name for p in one.iterdir())
    assert names == sorted(p.name for p in eight.iterdir())
    for f in ("validation.txt", "test.txt"):
        a, b = series(eight / f), series(one / f)
        assert [x[0] for x in a] == [x[0] for x in b] and [x[2] for x in a] == [x[2] for x in b]
        assert max(abs(x[1] - y[1]) for x, y in zip(a, b)) <= 1e-6
    for f in ("byusers.tsv", "byitems.tsv", "precision.txt"):
        assert (eight / f).read_text() == (one / f).read_text(), f
    for nm in ("hbeta", "htheta", "betarate", "thetarate", "betabias", "thetabias"):
        for suf in ("", "_shape", "_rate"):
            ia, va = read_tsv(eight / f"{nm}{suf}.tsv")
            ib, vb = read_tsv(one / f"{nm}{suf}.tsv")
            assert np.array_equal(ia, ib), nm + suf
            assert np.all(np.abs(va - vb) <= 1e-4 * np.abs(vb) + 5e-9), nm + suf
