#!/usr/bin/env python
"""Strong-scaling dry run on ONE GPU: split a config's users into N shards
(hgaprec_amd.dist.partition_users), run every shard's local half in turn,
sum the exchange buffers on the host (stand-in for the all-reduce) and run the
replicated half.  Reports per-shard device time (load balance of the nnz-based
partition) and checks the result against the unsharded run."""
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    import torch
    from hgaprec_amd import synth
    from hgaprec_amd.capi import Hpf
    from hgaprec_amd.dist import partition_users, shard_csr
    name = sys.argv[1] if len(sys.argv) > 1 else "C2"
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    cfg = dict(synth.CONFIGS[name])
    dev = torch.device("cuda", 0)
    n, m, K = cfg["n"], cfg["m"], cfg["K"]
    rowptr, col, val = synth.generate(n, m, cfg["nnz"], cfg["alpha_u"], cfg["alpha_i"], seed=cfg["seed"], device=dev,
                                      binary=cfg["binary"])
    st = {k: synth.initial_state(r, K, s, dev, prior_v=p) for k, r, s, p in
          (("theta", n, 1, None), ("beta", m, 2, None), ("xi", n, 3, K), ("eta", m, 4, K))}

    def make(a, b, nr, r):
        D = Hpf(b - a, m, K, hier=True, n_ranks=nr, rank=r, n_users_total=n, binary=cfg["binary"])
        rp, c, v = shard_csr(rowptr, col, val, a, b)
        D.upload_csr(rp, c, v)
        D.set_state("THETA_E", st["theta"]["E"][a:b]); D.set_state("THETA_ELOG", st["theta"]["Elog"][a:b])
        D.set_state("BETA_E", st["beta"]["E"]); D.set_state("BETA_ELOG", st["beta"]["Elog"])
        D.set_state("XI_E", st["xi"]["E"][a:b]); D.set_state("ETA_E", st["eta"]["E"])
        return D

    full = make(0, n, 1, 0)
    full.iterate(3); full.synchronize()
    t_full = full.mean_timing(2)
    ref_beta = full.get_state("BETA_E")
    full.close()
    parts = partition_users(rowptr, N)
    shards = [make(a, b, N, r) for r, (a, b) in enumerate(parts)]
    local_ms = np.zeros((3, N)); glob_ms = np.zeros((3, N))
    for it in range(3):
        tot = None
        for r, S in enumerate(shards):
            S.iterate_local(); S.synchronize()
            x = S.exchange_read()
            tot = x if tot is None else tot + x
        for r, S in enumerate(shards):
            S.exchange_write(tot)
            S.iterate_global(); S.synchronize()
            t = S.last_timing()
            local_ms[it, r] = t["phi_user_ms"] + t["combine_user_ms"] + t["phi_item_ms"] + t["combine_item_ms"] + t["sweep_user_ms"]
            glob_ms[it, r] = t_full["sweep_item_ms"]     # the replicated half costs what it costs unsharded
            # (this handle's own sweep_item interval also spans the host-side exchange above)
    err = float(np.max(np.abs(shards[0].get_state("BETA_E") - ref_beta) / ref_beta))
    loc = local_ms[1:].mean(0); glo = glob_ms[1:].mean(0)
    out = {
        "config": name, "shards": N, "users_per_shard": [b - a for a, b in parts],
        "nnz_per_shard": [int(rowptr[b] - rowptr[a]) for a, b in parts],
        "local_ms_per_shard": [round(float(v), 3) for v in loc],
        "replicated_item_sweep_ms": round(float(glo.mean()), 3),
        "single_gpu_iteration_ms": round(t_full["iteration_ms"], 3),
        "ideal_ms": round(t_full["iteration_ms"] / N, 3),
        "slowest_shard_compute_ms": round(float((loc + glo).max()), 3),
        "compute_only_speedup": round(t_full["iteration_ms"] / float((loc + glo).max()), 2),
        "exchange_MB": round(shards[0].exchange_count() * 8 / 1e6, 1),
        "max_rel_diff_beta_vs_unsharded": err,
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
