#!/usr/bin/env python
"""What XCD-local placement of the user-major phi pass could win, measured
without building it.

Workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md) and the user pass gives
four consecutive segments (users) to a workgroup.  Variant "xcd" of C2 re-labels
every nonzero's item so that workgroup b only touches items with
item % 8 == b % 8: each XCD's L2 then sees one eighth of W_beta (10 MB at C2,
its popular rows resident) -- the BEST case an XCD-aware bucketing could reach,
with none of its costs (no 8 partial rows per user, no combine).  Variant
"base" is plain C2; "m2000" shrinks the item matrix into one L2 (the latency
floor of the kernel).  Same degree sequences and popularity law in all three.

  python tools/xcd_locality_probe.py [base|xcd|m2000] [steps]

Run under `rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum` for the L2 hit rates.
"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    import torch
    from hgaprec_amd import synth
    from hgaprec_amd.capi import Hpf
    variant = sys.argv[1] if len(sys.argv) > 1 else "base"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    cfg = dict(synth.CONFIGS["C2"])
    n, m, K = cfg["n"], cfg["m"], cfg["K"]
    dev = torch.device("cuda", 0)
    rowptr, col, val = synth.generate_device(n, m, cfg["nnz"], cfg["alpha_u"], cfg["alpha_i"], seed=cfg["seed"], device=dev)
    if variant == "m2000":
        m = 2000
        col = (col.to(torch.int64) % m).to(torch.int32)
    D = Hpf(n, m, K, hier=True)
    if variant == "xcd":
        # segment index of a user: users in order, rows longer than 512 take several segments
        deg = rowptr[1:] - rowptr[:-1]
        nseg = torch.clamp((deg + 511) // 512, min=1)
        first_seg = torch.cumsum(nseg, 0) - nseg
        xcd_of_user = (first_seg // 4) % 8
        u = torch.repeat_interleave(torch.arange(n, device=dev), deg)
        c64 = col.to(torch.int64)
        col = ((c64 // 8) * 8 + xcd_of_user[u]).clamp(max=m - 1).to(torch.int32)
        del u, c64
    D.upload_csr_device(rowptr, col, val)
    st = synth.initial_state_device(n, K, 1, dev)
    D.set_state_device("THETA_E", st["E"]); D.set_state_device("THETA_ELOG", st["Elog"])
    st = synth.initial_state_device(m, K, 2, dev)
    D.set_state_device("BETA_E", st["E"]); D.set_state_device("BETA_ELOG", st["Elog"])
    D.set_state_device("XI_E", synth.initial_state_device(n, K, 3, dev, prior_v=K)["E"])
    D.set_state_device("ETA_E", synth.initial_state_device(m, K, 4, dev, prior_v=K)["E"])
    del st
    D.iterate(3)
    D.iterate(steps)
    D.synchronize()
    tm = D.mean_timing(steps)
    print(json.dumps({"variant": variant, "items": m, "nnz": int(rowptr[-1]),
                      **{k: round(v, 4) for k, v in tm.items() if k.endswith("_ms")}, "work": D.work_info()}))
    D.close()


if __name__ == "__main__":
    main()
