#!/usr/bin/env python
"""How far the compact W storage modes (f32: w_storage = 1, 48-bit: w_storage = 2) drift from the
fp64 CPU oracle as iterations go by, next to the default fp64 storage."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import orc
from hgaprec_amd.capi import Hpf
from tests.util import make_problem, copy_state, rel_err

for K, hier, bias in ((5, True, False), (21, True, True), (100, True, False), (50, False, True)):
    n, m = 400, 300
    rowptr, col, val = make_problem(n, m, 12000, 5)
    M = orc.Model(n, m, K, hier, bias, False); M.set_csr(rowptr, col, val); M.initialize(5)
    D = {ws: Hpf(n, m, K, hier=hier, bias=bias, w_storage=ws) for ws in (0, 1, 2)}
    for d in D.values():
        d.upload_csr(rowptr, col, val); copy_state(M, d, hier, bias)
    out = []
    done = 0
    for upto in (5, 20, 60, 150, 300):
        M.iterate(upto - done)
        for d in D.values():
            d.iterate(upto - done)
        done = upto
        te, be = M.state("THETA_E"), M.state("BETA_E")
        out.append((upto,) + tuple(max(rel_err(D[ws].get_state("THETA_E"), te), rel_err(D[ws].get_state("BETA_E"), be)) for ws in (0, 1, 2)))
    print(f"K={K} hier={hier} bias={bias}: " + "  ".join(f"it{u}: f64 {a:.1e} f32W {b:.1e} f48W {c:.1e}" for u, a, b, c in out), flush=True)
