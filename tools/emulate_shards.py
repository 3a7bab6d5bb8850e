#!/usr/bin/env python
"""Strong-scaling dry run on ONE GPU: cut a config's users into N shards exactly as
`bench.py --gpus N` does (degrees -> partition_users on the planned nnz prefix -> each
shard generated on its own from the counter-hash generator) and run them.

  python tools/emulate_shards.py C3 8                 all N handles resident, exchange buffers summed in
                                                      place of the all-reduce, result checked against the
                                                      unsharded run (C3, C4: they fit)
  python tools/emulate_shards.py C5 8 --sequential    one shard at a time (C5: a shard is 14 GB, the eight
                                                      and the whole do not fit together): each shard runs on
                                                      its own sums -- load balance and per-shard time only --
                                                      and the whole matrix is timed on its own afterwards

Reports per-shard nonzeros, tiles and device time (the load balance of the nnz-based
partition), the compute-only speed-up against the unsharded run, and how much of an
iteration may be EXPOSED communication if N GPUs are to reach 6x (8 GPUs) -- an
expectation from one-GPU measurements, not a scaling curve: nothing here crosses xGMI.
"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    import torch
    from bench import start_state
    from hgaprec_amd import synth
    from hgaprec_amd.capi import Hpf
    from hgaprec_amd.dist import partition_users
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    sequential = "--sequential" in sys.argv
    name = args[0] if args else "C2"
    N = int(args[1]) if len(args) > 1 else 8
    iters = 3
    cfg = dict(synth.CONFIGS[name])
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    n, m, K = cfg["n"], cfg["m"], cfg["K"]
    deg = synth.degrees(n, m, cfg["nnz"], cfg["alpha_u"], cfg["seed"], dev)
    planned = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(deg, 0, out=planned[1:])

    def make(a, b, nr, r, bind=True):
        rp, c, v = synth.generate_device(n, m, cfg["nnz"], cfg["alpha_u"], cfg["alpha_i"], seed=cfg["seed"], device=dev,
                                         binary=cfg["binary"], user_range=(a, b), deg=deg)
        torch.cuda.empty_cache()
        D = Hpf(b - a, m, K, hier=cfg["hier"], bias=cfg["bias"], n_ranks=nr, rank=r, n_users_total=n, binary=cfg["binary"],
                stream=stream.cuda_stream)
        x = None
        if bind:
            x = torch.zeros(D.exchange_count(), dtype=torch.float64, device=dev)
            D.bind_exchange_buffer(x.data_ptr(), x.numel())
        D.upload_csr_device(rp, c, v)
        nnz = int(rp[-1])
        del rp, c, v
        start_state(D, cfg, b - a, a, cfg["seed"], n, dev)
        return D, x, nnz

    def times(t):
        return (t["phi_user_ms"] + t["combine_user_ms"] + t["phi_item_ms"] + t["combine_item_ms"] + t["sweep_user_ms"],
                t["sweep_item_ms"], t["phi_user_ms"] + t["combine_user_ms"] + t["sweep_user_ms"])

    parts = partition_users(planned.cpu().numpy(), N)
    per_shard = []
    err = None
    if not sequential:
        full, _, nnz_full = make(0, n, 1, 0)
        full.iterate(iters); full.synchronize()
        t_full = full.mean_timing(iters - 1)
        ref_beta = full.get_state_device("BETA_E", dev)
        full.close()
        torch.cuda.empty_cache()
        shards = [make(a, b, N, r) for r, (a, b) in enumerate(parts)]
        loc = torch.zeros(iters, N); swi = torch.zeros(iters, N); cov = torch.zeros(iters, N)
        for it in range(iters):
            for S, _, _ in shards:                   # one after the other: every handle has its own events
                S.iterate_local()
                S.synchronize()
            tot = sum(x for _, x, _ in shards)
            for S, x, _ in shards:
                x.copy_(tot)
            torch.cuda.synchronize()
            for r, (S, _, _) in enumerate(shards):
                S.iterate_global(); S.synchronize()
                loc[it, r], swi[it, r], cov[it, r] = times(S.last_timing())
        got = shards[0][0].get_state_device("BETA_E", dev)
        err = float(((got - ref_beta).abs() / ref_beta).max())
        for r, (S, _, nz) in enumerate(shards):
            wi = S.work_info()
            per_shard.append({"users": parts[r][1] - parts[r][0], "nnz": nz, "tiles_user": wi["tiles_user"], "tiles_item": wi["tiles_item"],
                              "local_ms": round(float(loc[1:, r].mean()), 3), "item_sweep_ms": round(float(swi[1:, r].mean()), 3),
                              "user_half_ms": round(float(cov[1:, r].mean()), 3)})
        exch_mb = shards[0][0].exchange_count() * 8 / 1e6
        for S, _, _ in shards:
            S.close()
        single_ms = t_full["iteration_ms"]
    else:
        exch_mb = None
        for r, (a, b) in enumerate(parts):
            S, _, nz = make(a, b, N, r, bind=False)
            for _ in range(iters):                   # the shard on its own sums: what 1 of N ranks computes, minus the exchange
                S.iterate_local()
                S.iterate_global()
            S.synchronize()
            t = S.mean_timing(iters - 1)
            wi = S.work_info()
            lo, sw, cv = times(t)
            per_shard.append({"users": b - a, "nnz": nz, "tiles_user": wi["tiles_user"], "tiles_item": wi["tiles_item"],
                              "local_ms": round(lo, 3), "item_sweep_ms": round(sw, 3), "user_half_ms": round(cv, 3)})
            exch_mb = S.exchange_count() * 8 / 1e6
            S.close()
            torch.cuda.empty_cache()
            print(f"[shard {r}] {per_shard[-1]}", file=sys.stderr, flush=True)
        full, _, nnz_full = make(0, n, 1, 0, bind=False)
        full.iterate(iters); full.synchronize()
        single_ms = full.mean_timing(iters - 1)["iteration_ms"]
        full.close()
    comp = [p["local_ms"] + p["item_sweep_ms"] for p in per_shard]
    nnzs = [p["nnz"] for p in per_shard]
    out = {
        "config": name, "shards": N, "nnz_total": nnz_full, "mode": "sequential (each shard on its own sums)" if sequential else "resident (exchange buffers summed)",
        "per_shard": per_shard,
        "nnz_imbalance_max_over_mean": round(max(nnzs) / (sum(nnzs) / N), 5),
        "time_imbalance_max_over_mean": round(max(comp) / (sum(comp) / N), 4),
        "single_gpu_iteration_ms": round(single_ms, 3),
        "ideal_ms": round(single_ms / N, 3),
        "slowest_shard_compute_ms": round(max(comp), 3),
        "compute_only_speedup": round(single_ms / max(comp), 2),
        "exchange_MB": round(exch_mb, 1),
        # what may be exposed of the all-reduce (and of anything else that is not compute) per iteration if N GPUs are to be
        # `target` times faster than one: single / target - slowest shard; the big all-reduce travels under the user half
        "budget_ms_exposed_for": {f"{t}x": round(single_ms / t - max(comp), 3) for t in (6, 7)} if N >= 8 else
                                 {f"{0.75 * N:g}x": round(single_ms / (0.75 * N) - max(comp), 3)},
        "user_half_ms_that_can_cover_the_item_allreduce": round(min(p["user_half_ms"] for p in per_shard), 3),
        "max_rel_diff_beta_vs_unsharded": err,
        "note": "shards run one after another on ONE GPU: compute and load balance only, no xGMI traffic -- an expectation, not a measured scaling curve",
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
