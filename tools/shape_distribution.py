#!/usr/bin/env python
"""How large are the Gamma shapes the row sweep evaluates psi() on?  (VERDICT r01 #7
proposed a cheaper path for shapes >= 10.)  At C2 after 1 / 5 / 20 / 60 iterations:
theta shapes below 10: 0.991 / 0.991 / 0.953 / 0.945 (below 1: 0.001 / 0.026 / 0.680 /
0.768 -- phi concentrates on a few factors); beta shapes below 10: 0.66 / 0.66 / 0.79 /
0.97.  The ten-term shift IS the common path; a fast path for large shapes would run on
~5 % of the lanes."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from hgaprec_amd import synth
from hgaprec_amd.capi import Hpf
cfg = dict(synth.CONFIGS["C2"]); n, m, K = cfg["n"], cfg["m"], cfg["K"]
dev = torch.device("cuda", 0)
rp, c, v = synth.generate_device(n, m, cfg["nnz"], cfg["alpha_u"], cfg["alpha_i"], seed=cfg["seed"], device=dev)
D = Hpf(n, m, K, hier=True); D.upload_csr_device(rp, c, v)
for w, rows, s in (("THETA", n, 1), ("BETA", m, 2)):
    st = synth.initial_state_device(rows, K, s, dev); D.set_state_device(w + "_E", st["E"]); D.set_state_device(w + "_ELOG", st["Elog"])
D.set_state_device("XI_E", synth.initial_state_device(n, K, 3, dev, prior_v=K)["E"])
D.set_state_device("ETA_E", synth.initial_state_device(m, K, 4, dev, prior_v=K)["E"])
for it in (1, 5, 20, 60):
    D.iterate(it - D.last_timing()["iterations"])
    ts = D.get_state_device("THETA_SHAPE", dev); bs = D.get_state_device("BETA_SHAPE", dev)
    print(it, "theta shape < 10:", float((ts < 10).double().mean()), " beta shape < 10:", float((bs < 10).double().mean()),
          " theta < 1:", float((ts < 1).double().mean()))
