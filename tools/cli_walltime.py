#!/usr/bin/env python
"""Where the `hgaprec` CLI spends wall time, end to end, on a TSV dataset.

    python tools/cli_walltime.py [--config mid|C2] [--iters 100] [--rfreq 10] [--out FILE.json]

mid = 200K x 20K, 1e7 ratings; C2 = BASELINE.json's one-GPU configuration (1M x 100K, 5e7 ratings)
written out as train / validation / test .tsv the way the reference expects them.  The binary runs
with HPF_CLI_TIMING=1 and prints the wall seconds of its phases on stderr (`[timing] ...`): reading
the three files, the hand-over to the device, the MT19937 start state, the iterations themselves,
the report steps (held-out likelihood, save_model, ranking evaluation).  Run twice: as is and with
`-cache` (second run: the binary image instead of the text)."""
import argparse, json, os, re, subprocess, sys, tempfile, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from hgaprec_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="mid")
ap.add_argument("--iters", type=int, default=100)
ap.add_argument("--rfreq", type=int, default=10)
ap.add_argument("--out", default=None)
ap.add_argument("--ranks", type=int, default=1, help="> 1: also run `-ngpus N -comm host` with all ranks on GPU 0 (what the rank plumbing costs: "
                                                     "hand-over of the parsed data, part files of the user-side matrices)")
a = ap.parse_args()

if a.config == "mid":
    n, m, nnz, K, au, ai, seed = 200_000, 20_000, 10_000_000, 100, 0.5, 0.8, 5
else:
    c = synth.CONFIGS[a.config]
    n, m, nnz, K, au, ai, seed = c["n"], c["m"], c["nnz"], c["K"], c["alpha_u"], c["alpha_i"], c["seed"]
rowptr, col, val = synth.generate(n, m, nnz, au, ai, seed=seed, device="cuda" if torch.cuda.is_available() else "cpu")
td = Path(tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp")))
u = np.repeat(np.arange(n, dtype=np.int64), np.diff(rowptr))
rng = np.random.default_rng(0)
split = rng.random(u.size)


def write_tsv(path, sel):
    import pyarrow as pa, pyarrow.csv as pc
    t = pa.table({"u": u[sel] + 1, "i": np.asarray(col)[sel].astype(np.int64) + 1, "y": np.asarray(val)[sel].astype(np.int64)})
    pc.write_csv(t, path, pc.WriteOptions(include_header=False, delimiter="\t"))


t0 = time.time()
write_tsv(td / "train.tsv", split >= 0.002)
write_tsv(td / "validation.tsv", split < 0.001)
write_tsv(td / "test.tsv", (split >= 0.001) & (split < 0.002))
res = {"config": a.config, "n": n, "m": m, "nnz": int(u.size), "K": K, "iters": a.iters, "rfreq": a.rfreq,
       "train_tsv_MB": round((td / "train.tsv").stat().st_size / 1e6, 1), "runs": {}}
print(f"wrote TSVs ({res['train_tsv_MB']:.0f} MB train) in {time.time() - t0:.1f}s", flush=True)
exe = Path(__file__).resolve().parent.parent / "hgaprec_amd" / "hgaprec"
env = dict(os.environ, HPF_CLI_TIMING="1")
cases = [("text", []), ("cache_write", ["-cache"]), ("cache_read", ["-cache"])]
if a.ranks > 1:
    cases.append((f"ranks{a.ranks}_one_gpu", ["-ngpus", str(a.ranks), "-comm", "host", "-device", "0"]))
for label, extra in cases:
    t0 = time.time()
    r = subprocess.run([str(exe), "-dir", str(td), "-n", str(n), "-m", str(m), "-k", str(K), "-hier",
                        "-rfreq", str(a.rfreq), "-max-iterations", str(a.iters)] + extra, cwd=td, capture_output=True, text=True, env=env)
    wall = time.time() - t0
    ph = {m_.group(1).strip(): float(m_.group(2)) for m_ in re.finditer(r"^\[timing\] (.+?)\s+([0-9.]+)(?: s)?$", r.stderr, re.M)}
    ep = {m_.group(1): float(m_.group(2)) for m_ in re.finditer(r"^\[timing-epoch\] (\w+) ([0-9.]+)$", r.stderr, re.M)}
    if "main" in ep and "exit" in ep:       # what the binary's own phases cannot see: loading it, and leaving it
        ph["before main (exec, shared libraries)"] = round(ep["main"] - t0, 3)
        ph["after exit() was called"] = round(t0 + wall - ep["exit"], 3)
    res["runs"][label] = {"wall_s": round(wall, 2), "rc": r.returncode, "phases_s": ph}
    print(f"{label}: {wall:.2f}s rc={r.returncode}", flush=True)
    for k, v in ph.items():
        print(f"    {k:30s} {v:8.3f}", flush=True)
    if r.returncode:
        print(r.stderr[-2000:])
print(json.dumps(res))
if a.out:
    Path(a.out).write_text(json.dumps(res, indent=1) + "\n")
