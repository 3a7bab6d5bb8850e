#!/usr/bin/env python
"""Generate tests/golden/*.json from the REFERENCE's own code, in this container.

Runs oracle/_ref/refpart -- the GSL-free translation units of the reference
(matrix.hh, env.hh, log.cc) compiled in place from /root/reference/src by
oracle/Makefile -- on seeded inputs and stores inputs + outputs.  Doubles are
stored as hex strings (float.hex) so the fixtures are bit-exact.  The rest of
the reference needs GSL (absent here), so there are no end-to-end fixtures:
see DESIGN.md "Oracle".

    python tests/golden/make_golden.py        # rewrites tests/golden/
    python tests/golden/make_golden.py rows   # only the named fixture files
"""
from __future__ import annotations

import json
import os
import struct
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
REFPART = ROOT / "oracle" / "_ref" / "refpart"
GOLD = ROOT / "tests" / "golden"


def hexl(a):
    return [float(x).hex() for x in np.asarray(a, np.float64).ravel()]


def softmax_cases():
    rng = np.random.default_rng(20260929)
    recs = []
    sizes = [1, 2, 3, 5, 7, 20, 22, 50, 100, 102, 200]
    for n in sizes:
        for scale, shift in ((1.0, 0.0), (5.0, -10.0), (30.0, -40.0)):
            x = rng.normal(size=n) * scale + shift
            y = int(rng.integers(0, 6))
            recs.append((x, y))
    # hand-picked: ties, huge spread (underflow of the small terms), zeros
    recs.append((np.zeros(8), 1))
    recs.append((np.array([-700.0, 0.0, -1.0]), 3))
    recs.append((np.array([0.0, -745.0, -800.0, -0.5]), 2))
    recs.append((np.array([3.0, 3.0, 3.0, 3.0]), 5))
    recs.append((np.linspace(-60, 5, 100), 4))
    with tempfile.TemporaryDirectory() as td:
        fin, fout = Path(td) / "in.bin", Path(td) / "out.bin"
        with open(fin, "wb") as f:
            f.write(struct.pack("<I", len(recs)))
            for x, y in recs:
                f.write(struct.pack("<II", x.size, y))
                f.write(np.asarray(x, "<f8").tobytes())
        subprocess.run([str(REFPART), "softmax", str(fin), str(fout)], check=True)
        raw = np.fromfile(fout, "<f8")
    out, pos = [], 0
    for x, y in recs:
        ls = raw[pos]; phi = raw[pos + 1: pos + 1 + x.size]; pos += 1 + x.size
        out.append({"x": hexl(x), "y": y, "logsum": float(ls).hex(), "phi": hexl(phi)})
    return {"source": "D1Array<double>::logsum/lognormalize/scale, /root/reference/src/matrix.hh:367-406",
            "cases": out}


def accumulate_cases():
    rng = np.random.default_rng(7)
    cases = []
    for rows, K, bias in ((6, 5, False), (9, 20, True), (4, 100, True)):
        width = K + 2 if bias else K
        nrec = 40
        recs = [(int(rng.integers(rows)), int(rng.integers(0, 6)), rng.normal(size=width) * 3 - 5)
                for _ in range(nrec)]
        with tempfile.TemporaryDirectory() as td:
            fin, fout = Path(td) / "in.bin", Path(td) / "out.bin"
            with open(fin, "wb") as f:
                f.write(struct.pack("<IIII", rows, K, width, nrec))
                for row, y, x in recs:
                    f.write(struct.pack("<II", row, y))
                    f.write(np.asarray(x, "<f8").tobytes())
            subprocess.run([str(REFPART), "accumulate", str(fin), str(fout)], check=True)
            Mx = np.fromfile(fout, "<f8")
        cases.append({"rows": rows, "K": K, "width": width,
                      "recs": [{"row": r, "y": y, "x": hexl(x)} for r, y, x in recs],
                      "M": hexl(Mx)})
    return {"source": "lognormalize + scale + D2Array::add_slice (matrix.hh:1060-1067): adds only the "
                      "first K entries of a K+2 wide phi; rows start at the 0.3 prior", "cases": cases}


def arrays_cases():
    """D1Array::sum (a plain left-to-right double sum -- what sum_rows / sum_cols of gpbase.hh
    are made of), D2Array::set_elements, D1Array::zero"""
    rng = np.random.default_rng(11)
    cases = []
    for n in (1, 2, 7, 100, 333):
        x = rng.gamma(0.3, 1.0, size=n) * 10.0 ** rng.integers(-8, 6, size=n)
        maxn, fill = max(1, n // 2), 0.3
        with tempfile.TemporaryDirectory() as td:
            fin, fout = Path(td) / "in.bin", Path(td) / "out.bin"
            with open(fin, "wb") as f:
                f.write(struct.pack("<IId", n, maxn, fill))
                f.write(np.asarray(x, "<f8").tobytes())
            subprocess.run([str(REFPART), "arrays", str(fin), str(fout)], check=True)
            raw = np.fromfile(fout, "<f8")
        cases.append({"x": hexl(x), "maxn": maxn, "fill": float(fill).hex(), "sum": float(raw[0]).hex(),
                      "sum_maxn": float(raw[1]).hex(), "set_elements": hexl(raw[2:2 + 3 * n]),
                      "zero": hexl(raw[2 + 3 * n:2 + 4 * n])})
    return {"source": "D1Array<double>::sum matrix.hh:327-335, zero 200-203, D2Array<double>::set_elements 930-936",
            "cases": cases}


def rows_cases():
    """the matrix.hh calls the sweep methods of gpbase.hh are made of -- D2Array::set_elements(row, v) (946-951),
    add_slice with the same vector on every row (1060-1067), D1Array::operator+= (464-472), D2Array / D1Array::swap
    (971-977, 449-454), set_elements(prior) -- run by the reference's own containers in the order gpbase.hh calls
    them (the order is the harness's: gpbase.hh needs GSL).  Pins the oracle's gp_set_prior_rate /
    gp_update_rate_next / gp_array_* / gp_swap, bit for bit."""
    rng = np.random.default_rng(5)
    cases = []
    for mode, rows, k in ((0, 7, 5), (0, 3, 100), (0, 1, 1), (1, 6, 4), (1, 2, 100), (2, 9, 1), (2, 1, 1), (3, 8, 1)):
        sprior, rprior = 0.3, 0.3
        v = float(0.3 * 20) if mode == 2 else float(200 + rows) if mode == 3 else 0.0
        snext = 0.3 + rng.gamma(0.5, 2.0, size=(rows, k)) * 10.0 ** rng.integers(-6, 4, size=(rows, k))
        ev = rng.gamma(2.0, 0.1, size=rows)
        u = rng.gamma(1.0, 50.0, size=rows if mode == 2 else k)
        with tempfile.TemporaryDirectory() as td:
            fin, fout = Path(td) / "in.bin", Path(td) / "out.bin"
            with open(fin, "wb") as f:
                f.write(struct.pack("<IIIddd", mode, rows, k, sprior, rprior, v))
                f.write(snext.astype("<f8").tobytes()); f.write(ev.astype("<f8").tobytes()); f.write(u.astype("<f8").tobytes())
            subprocess.run([str(REFPART), "rows", str(fin), str(fout)], check=True)
            raw = np.fromfile(fout, "<f8")
        ns, nr = rows * k, (k if mode == 1 else rows * k)
        assert raw.size == 2 * ns + 2 * nr
        cases.append({"mode": mode, "rows": rows, "k": k, "sprior": float(sprior).hex(), "rprior": float(rprior).hex(),
                      "v": float(v).hex(), "snext_in": hexl(snext), "ev": hexl(ev), "u": hexl(u),
                      "scurr": hexl(raw[:ns]), "rcurr": hexl(raw[ns:ns + nr]),
                      "snext": hexl(raw[ns + nr:2 * ns + nr]), "rnext": hexl(raw[2 * ns + nr:])})
    return {"source": "D2Array<double>::set_elements(row, v) matrix.hh:946-951, add_slice 1060-1067, swap 971-977; "
                      "D1Array<double>::operator+= 464-472, swap 449-454, set_elements 193-197 -- called in the order of "
                      "gpbase.hh:149-246 (GPMatrix, mode 0; update_rate_next_all, mode 3), 522-578 (GPMatrixGR, mode 1), "
                      "858-903 (GPArray, mode 2)", "cases": cases}


def save_cases():
    rng = np.random.default_rng(3)
    cases = []
    for rows, cols, nids in ((5, 4, 5), (6, 1, 3), (3, 7, 0)):
        A = np.concatenate([rng.gamma(0.3, 1.0, size=(rows, cols)),]).astype(np.float64)
        A.flat[0] = 1e-9; A.flat[-1] = 123456.789012345
        v = rng.gamma(2.0, 3.0, size=rows)
        ids = rng.integers(1, 10 ** 6, size=nids).astype(np.uint32)
        with tempfile.TemporaryDirectory() as td:
            fin = Path(td) / "in.bin"
            with open(fin, "wb") as f:
                f.write(struct.pack("<III", rows, cols, nids))
                f.write(ids.astype("<u4").tobytes())
                f.write(A.astype("<f8").tobytes())
                f.write(v.astype("<f8").tobytes())
            subprocess.run([str(REFPART), "save", str(fin), str(Path(td) / "m.tsv"), str(Path(td) / "v.tsv")],
                           check=True)
            mt, vt = (Path(td) / "m.tsv").read_text(), (Path(td) / "v.tsv").read_text()
        cases.append({"rows": rows, "cols": cols, "ids": ids.tolist(), "A": hexl(A), "v": hexl(v),
                      "matrix_tsv": mt, "vector_tsv": vt})
    return {"source": "D2Array<double>::save matrix.hh:1140-1166, D1Array<double>::save matrix.hh:725-744",
            "cases": cases}


def env_cases():
    arglists = [
        ["-dir", "data/ml", "-n", "300", "-m", "200", "-k", "5", "-hier", "-seed", "7"],
        ["-dir", "data/ml", "-n", "300", "-m", "200", "-k", "5", "-hier", "-bias", "-rfreq", "1"],
        ["-dir", "/abs/path", "-n", "6040", "-m", "3681", "-k", "20", "-hier", "-binary-data"],
        ["-dir", "ml", "-n", "10", "-m", "20", "-k", "3"],
        ["-dir", "movielens", "-n", "10", "-m", "20", "-k", "3", "-bias", "-label", "run1"],
        ["-dir", "netflix", "-n", "480189", "-m", "17770", "-k", "200", "-hier", "-bias", "-a", "0.5",
         "-b", "0.25", "-c", "1", "-d", "2", "-seed", "2147483648"],
        ["-dir", "9data", "-n", "1", "-m", "1", "-k", "1", "-hier", "-seed", "1234567", "-max-iterations", "50"],
        ["-dir", "xy", "-n", "1", "-m", "1", "-k", "1", "-hier", "-seed", "0.5"],
    ]
    cases = []
    for args in arglists:
        with tempfile.TemporaryDirectory() as td:
            r = subprocess.run([str(REFPART), "env"] + args, cwd=td, check=True, capture_output=True, text=True)
            prefix = r.stdout.strip().splitlines()[-1]
            file_str = r.stdout.strip().splitlines()[-2]
            param = (Path(td) / prefix / "param.txt").read_text()
            files = sorted(os.listdir(Path(td) / prefix))
        cases.append({"args": args, "prefix": prefix, "file_str_x_tsv": file_str, "param_txt": param, "files": files})
    return {"source": "Env::Env /root/reference/src/env.hh:216-408 (directory name, param.txt head), "
                      "Logger::initialize log.cc:9-118", "cases": cases}


# ---------------------------------------------------------------------------
# End-to-end fixtures from the WHOLE reference binary.  Needs
# oracle/_ref/hgaprec_ref, which `make -C oracle ref-full` builds only where a genuine GSL
# is installed -- not in this image, so nothing below has ever produced a file here and
# end-to-end parity stays "unpinned" (DESIGN.md section 7).  On a machine with GSL:
#     make -C oracle ref-full && python tests/golden/make_golden.py --full
# writes tests/golden/full/<case>/{case.json, data/*.tsv}; tests/test_gpu_cli.py
# (test_cli_matches_reference_fixtures) then runs the MI355X CLI on the same inputs.
# ---------------------------------------------------------------------------
REFBIN = ROOT / "oracle" / "_ref" / "hgaprec_ref"
FULL_CASES = [   # name, (n, m, nnz, seed, problem kwargs), K, flags
    ("F1_hier", (300, 200, 9000, 17, {}), 5, ["-hier", "-rfreq", "1", "-max-iterations", "20"]),
    ("F2_hier_bias", (300, 200, 9000, 17, {}), 5, ["-hier", "-bias", "-rfreq", "1", "-max-iterations", "20"]),
    ("F3_hier_binary", (300, 200, 9000, 17, {}), 5, ["-hier", "-binary-data", "-rating-threshold", "4", "-rfreq", "1", "-max-iterations", "20"]),
    ("F4_vb", (300, 200, 9000, 17, {}), 5, []),
    ("F4_vb_bias", (300, 200, 9000, 17, {}), 5, ["-bias"]),
    ("F4_vb_bias_novb", (300, 200, 9000, 17, {}), 5, ["-bias", "-novb"]),
    ("F5_k100", (500, 300, 20000, 23, {}), 100, ["-hier", "-rfreq", "2", "-max-iterations", "6"]),
    ("F6_power_law", (400, 300, 12000, 31, {"heavy_user": True, "heavy_item": True, "singles": True}), 8,
     ["-hier", "-rfreq", "2", "-max-iterations", "10"]),
    ("F8_seed0", (300, 200, 9000, 17, {}), 5, ["-hier", "-rfreq", "1", "-max-iterations", "0", "-seed", "0"]),
    ("F8_seed_2_31", (300, 200, 9000, 17, {}), 5, ["-hier", "-rfreq", "1", "-max-iterations", "0", "-seed", "2147483648"]),
    ("F9_logl", (300, 200, 9000, 17, {}), 6, ["-hier", "-bias", "-logl", "-rfreq", "2", "-max-iterations", "8"]),
]


def full_fixtures():
    sys.path.insert(0, str(ROOT))
    from tests.test_gpu_cli import write_dataset
    from tests.util import make_problem           # noqa: F401  (write_dataset draws from it)
    out_root = GOLD / "full"
    for name, (n, m, nnz, seed, kw), K, flags in FULL_CASES:
        cdir = out_root / name
        data = cdir / "data"
        if cdir.exists():
            import shutil
            shutil.rmtree(cdir)
        write_dataset(data, n, m, nnz, seed=seed, **kw)
        ids = sorted({int(l.split("\t")[0]) for l in (data / "test.tsv").read_text().splitlines()})
        (data / "test_users.tsv").write_text("".join(f"{u}\n" for u in ids[:60]))
        args = ["-dir", str(data), "-n", str(n), "-m", str(m), "-k", str(K)] + (flags if "-seed" in flags else ["-seed", "7"] + flags)
        with tempfile.TemporaryDirectory() as td:
            r = subprocess.run([str(REFBIN)] + args, cwd=td, capture_output=True, text=True, timeout=3600)
            if r.returncode != 0:
                sys.exit(f"{name}: the reference exited with {r.returncode}\n{r.stderr[-2000:]}")
            outs = [p for p in Path(td).iterdir() if p.is_dir()]
            assert len(outs) == 1, outs
            files = {}
            for f in sorted(outs[0].iterdir()):
                if f.suffix in (".tsv", ".txt") and f.stat().st_size:
                    lines = f.read_text().splitlines()
                    step = max(1, len(lines) // 400) if f.suffix == ".tsv" and len(lines) > 400 else 1
                    files[f.name] = {"rows": len(lines), "every": step, "lines": lines[::step]}
            case = {"source": "premgopalan/hgaprec built by oracle/Makefile ref-full against the installed GSL",
                    "args": [a if a != str(data) else "DATA" for a in args], "prefix": outs[0].name,
                    "n": n, "m": m, "K": K, "files": files}
        (cdir / "case.json").write_text(json.dumps(case, indent=1))
        print("wrote", cdir)


def main():
    if "--full" in sys.argv:
        if not REFBIN.exists():
            sys.exit("oracle/_ref/hgaprec_ref missing: `make -C oracle ref-full` builds it only where a genuine "
                     "GSL is installed (this image has none); no end-to-end fixtures can be generated here")
        full_fixtures()
        return
    if not REFPART.exists():
        sys.exit("oracle/_ref/refpart missing: run `make -C oracle ref` where /root/reference exists")
    GOLD.mkdir(parents=True, exist_ok=True)
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    for name, fn in (("softmax", softmax_cases), ("accumulate", accumulate_cases), ("arrays", arrays_cases),
                     ("rows", rows_cases), ("save", save_cases), ("env", env_cases)):
        if only and name not in only:
            continue
        (GOLD / f"{name}.json").write_text(json.dumps(fn(), indent=1))
        print("wrote", GOLD / f"{name}.json")


if __name__ == "__main__":
    main()
