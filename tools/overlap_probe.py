#!/usr/bin/env python
"""EXPERIMENT ("Next round" 4): the fp64-issue-bound user sweep on a second stream underneath the
fabric-bound item pass -- concurrency where fusion lost.

  python tools/overlap_probe.py [C2|C3s|C4] [--cus 16,32,64]

For every mode of HPF_OVERLAP (0: today's order; 1: a plain second stream; 2: the sweep's stream masked to N CUs;
3: also the item pass masked to the other CUs) the same workload runs warm-up + 10 iterations from the same start
state; prints wall ms / iteration, the per-kernel hipEvent means and whether THETA / BETA are BIT-IDENTICAL to
mode 0 after the same number of iterations (same arithmetic on the same data: they must be).
C3s = what one of 8 GPUs holds of C3.
"""
import json
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    import torch
    from bench import OTHER_CONFIGS, start_state
    from hgaprec_amd import synth
    from hgaprec_amd.capi import Hpf
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    name = args[0] if args else "C2"
    cus = [32]
    if "--cus" in sys.argv:
        cus = [int(x) for x in sys.argv[sys.argv.index("--cus") + 1].split(",")]
    over = {}
    base = name
    if name == "C3s":
        base, over = "C3", dict(OTHER_CONFIGS[1][2])
    cfg = dict(synth.CONFIGS[base])
    cfg.update({k: v for k, v in over.items() if not k.startswith("_")})
    n, m, K = cfg["n"], cfg["m"], cfg["K"]
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    rowptr, col, val = synth.generate_device(n, m, cfg["nnz"], cfg["alpha_u"], cfg["alpha_i"], seed=cfg["seed"], device=dev,
                                             binary=cfg["binary"])
    nnz = int(rowptr[-1])
    torch.cuda.empty_cache()
    os.environ["HPF_EXPERIMENTAL"] = "1"
    ref = None
    out = {"workload": f"{name}: {n} x {m}, {nnz} nnz, K={K}", "runs": []}
    modes = [(0, 0)] + [(1, 0)] + [(md, c) for md in (2, 3) for c in cus] + [(0, 0)]
    for mode, c in modes:
        os.environ["HPF_OVERLAP"] = str(mode)
        if c:
            os.environ["HPF_OVERLAP_CUS"] = str(c)
        D = Hpf(n, m, K, hier=cfg["hier"], bias=cfg["bias"], binary=cfg["binary"], device=0, stream=stream.cuda_stream,
                n_users_total=cfg.get("n_users_total", n))
        D.upload_csr_device(rowptr, col, val)
        start_state(D, cfg, n, 0, cfg["seed"], cfg.get("n_users_total", n), dev)
        D.iterate(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        D.iterate(10)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        tm = D.mean_timing(10)
        th, be = D.get_state_device("THETA_E", dev), D.get_state_device("BETA_E", dev)
        if ref is None:
            ref = (th.clone(), be.clone())
        same = bool(torch.equal(th, ref[0]) and torch.equal(be, ref[1]))
        del th, be
        D.close()
        torch.cuda.empty_cache()
        r = {"HPF_OVERLAP": mode, "cus": c or None, "ms_per_iteration": round(ms, 4), "bit_identical_to_mode_0": same,
             "kernels_ms": {k: round(v, 4) for k, v in tm.items() if k.endswith("_ms")}}
        out["runs"].append(r)
        print(json.dumps(r), file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
