#!/usr/bin/env python
"""Where the `hgaprec` CLI spends wall time on a mid-size TSV dataset
(default: 200K x 20K, 1e7 ratings, K=100, -hier, 20 iterations, rfreq 10)."""
import subprocess, sys, time, os, tempfile
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from hgaprec_amd import synth

n, m, nnz, K = 200_000, 20_000, 10_000_000, 100
rowptr, col, val = synth.generate(n, m, nnz, 0.5, 0.8, seed=5, device="cuda" if torch.cuda.is_available() else "cpu")
td = Path(tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp")))
u = np.repeat(np.arange(n), np.diff(rowptr))
rng = np.random.default_rng(0)
split = rng.random(u.size)
t0 = time.time()
for name, sel in (("train.tsv", split >= 0.02), ("validation.tsv", split < 0.01), ("test.tsv", (split >= 0.01) & (split < 0.02))):
    np.savetxt(td / name, np.stack([u[sel] + 1, col[sel] + 1, val[sel]], 1), fmt="%d", delimiter="\t")
print(f"wrote TSVs ({(td / 'train.tsv').stat().st_size / 1e6:.0f} MB train) in {time.time() - t0:.1f}s", flush=True)
exe = Path(__file__).resolve().parent.parent / "hgaprec_amd" / "hgaprec"
for iters in (0, 20):
    t0 = time.time()
    r = subprocess.run([str(exe), "-dir", str(td), "-n", str(n), "-m", str(m), "-k", str(K), "-hier",
                        "-rfreq", "10", "-max-iterations", str(iters)], cwd=td, capture_output=True, text=True)
    print(f"max-iterations {iters}: {time.time() - t0:.2f}s  rc={r.returncode}", flush=True)
