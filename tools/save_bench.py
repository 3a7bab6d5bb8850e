#!/usr/bin/env python
"""save_matrix alone (no GPU): one 1M x 100 matrix, and three of them side by side the way save_object
writes an object; per thread count.  Where a model save's time goes on the host."""
import os, sys, time, threading
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from hgaprec_amd import hostlib
rows, cols = 1_000_000, 100
rng = np.random.default_rng(0)
a = [rng.gamma(0.3, 1.0, size=(rows, cols)) for _ in range(3)]
ids = np.arange(rows, dtype=np.uint32)
d = os.environ.get("TMPDIR", "/tmp")
for nt in os.environ.get("THREADS", "1 8 21 64").split():
    os.environ["HGAPREC_SAVE_THREADS"] = nt
    for rep in range(2):
        t = time.time(); hostlib.save_matrix(f"{d}/sb0.tsv", a[0], ids); one = time.time() - t
        th = [threading.Thread(target=hostlib.save_matrix, args=(f"{d}/sb{i}.tsv", a[i], ids)) for i in range(3)]
        t = time.time(); [x.start() for x in th]; [x.join() for x in th]; three = time.time() - t
        sz = os.path.getsize(f"{d}/sb0.tsv")
        print(f"threads {nt:>3}: one file {one:.3f} s ({sz / one / 1e9:.2f} GB/s)   three side by side {three:.3f} s ({3 * sz / three / 1e9:.2f} GB/s)", flush=True)
for i in range(3):
    os.unlink(f"{d}/sb{i}.tsv")
