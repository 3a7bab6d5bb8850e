#!/usr/bin/env python
"""Strong-scaling dry run on ONE GPU: cut a config's users into N shards exactly as
`bench.py --gpus N` does (degrees -> partition_users on the planned nnz prefix -> each
shard generated on its own from the counter-hash generator), keep all N handles resident,
run every shard's local half in turn, sum the exchange buffers (stand-in for the
all-reduce) and run the replicated half.  Reports per-shard nonzeros and device time
(the load balance of the nnz-based partition), the compute-only speed-up against the
unsharded run, and checks the sharded result against the unsharded one.

  python tools/emulate_shards.py C3 8
"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    import torch
    from hgaprec_amd import synth
    from hgaprec_amd.capi import Hpf
    from hgaprec_amd.dist import partition_users
    name = sys.argv[1] if len(sys.argv) > 1 else "C2"
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    iters = 3
    cfg = dict(synth.CONFIGS[name])
    dev = torch.device("cuda", 0)
    n, m, K = cfg["n"], cfg["m"], cfg["K"]
    deg = synth.degrees(n, m, cfg["nnz"], cfg["alpha_u"], cfg["seed"], dev)
    planned = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(deg, 0, out=planned[1:])

    def make(a, b, nr, r):
        rp, c, v = synth.generate_device(n, m, cfg["nnz"], cfg["alpha_u"], cfg["alpha_i"], seed=cfg["seed"], device=dev,
                                         binary=cfg["binary"], user_range=(a, b), deg=deg)
        D = Hpf(b - a, m, K, hier=True, n_ranks=nr, rank=r, n_users_total=n, binary=cfg["binary"])
        x = torch.zeros(D.exchange_count(), dtype=torch.float64, device=dev)
        D.bind_exchange_buffer(x.data_ptr(), x.numel())
        D.upload_csr_device(rp, c, v)
        nnz = int(rp[-1])
        del rp, c, v
        st = synth.initial_state_device(b - a, K, 1, dev, row0=a)
        D.set_state_device("THETA_E", st["E"]); D.set_state_device("THETA_ELOG", st["Elog"])
        st = synth.initial_state_device(m, K, 2, dev)
        D.set_state_device("BETA_E", st["E"]); D.set_state_device("BETA_ELOG", st["Elog"])
        D.set_state_device("XI_E", synth.initial_state_device(b - a, K, 3, dev, prior_v=K, row0=a)["E"])
        D.set_state_device("ETA_E", synth.initial_state_device(m, K, 4, dev, prior_v=K)["E"])
        del st
        torch.cuda.empty_cache()
        return D, x, nnz

    full, _, nnz_full = make(0, n, 1, 0)
    full.iterate(iters); full.synchronize()
    t_full = full.mean_timing(iters - 1)
    ref_beta = full.get_state_device("BETA_E", dev)
    full.close()
    torch.cuda.empty_cache()

    parts = partition_users(planned.cpu().numpy(), N)
    shards = [make(a, b, N, r) for r, (a, b) in enumerate(parts)]
    loc = torch.zeros(iters, N); swi = torch.zeros(iters, N)
    for it in range(iters):
        for S, _, _ in shards:                   # one after the other: every handle has its own stream
            S.iterate_local()
            S.synchronize()
        tot = sum(x for _, x, _ in shards)
        for S, x, _ in shards:
            x.copy_(tot)
        torch.cuda.synchronize()
        for r, (S, _, _) in enumerate(shards):
            S.iterate_global(); S.synchronize()
            t = S.last_timing()
            loc[it, r] = t["phi_user_ms"] + t["combine_user_ms"] + t["phi_item_ms"] + t["combine_item_ms"] + t["sweep_user_ms"]
            swi[it, r] = t["sweep_item_ms"]
    got = shards[0][0].get_state_device("BETA_E", dev)
    err = float(((got - ref_beta).abs() / ref_beta).max())
    lo, sw = loc[1:].mean(0), swi[1:].mean(0)
    per = (lo + sw)
    out = {
        "config": name, "shards": N, "nnz_total": nnz_full,
        "users_per_shard": [b - a for a, b in parts],
        "nnz_per_shard": [s[2] for s in shards],
        "nnz_imbalance_max_over_mean": round(max(s[2] for s in shards) / (sum(s[2] for s in shards) / N), 5),
        "local_ms_per_shard": [round(float(v), 3) for v in lo],
        "replicated_item_sweep_ms_per_shard": [round(float(v), 3) for v in sw],
        "single_gpu_iteration_ms": round(t_full["iteration_ms"], 3),
        "ideal_ms": round(t_full["iteration_ms"] / N, 3),
        "slowest_shard_compute_ms": round(float(per.max()), 3),
        "compute_only_speedup": round(t_full["iteration_ms"] / float(per.max()), 2),
        "exchange_MB": round(shards[0][0].exchange_count() * 8 / 1e6, 1),
        "max_rel_diff_beta_vs_unsharded": err,
        "note": "shards run one after another on ONE GPU: compute and load balance only, no xGMI traffic",
    }
    print(json.dumps(out))
    for S, _, _ in shards:
        S.close()


if __name__ == "__main__":
    main()
