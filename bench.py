#!/usr/bin/env python
"""bench.py -- rating-nonzeros/sec per CAVI iteration (steps A-F) on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full CAVI iteration (phi passes + row sweeps; report steps
excluded) over one synthetic ratings matrix that is resident in HBM before the
timed region starts.  N=1 runs BASELINE config C2 (1M x 100K, 5e7 nnz, K=100,
-hier).  N>1 is weak scaling: every rank owns a C2-sized shard of users (its
own 5e7 nonzeros), items are shared, and the item-side shape sums plus
sum_u E[theta_u] go through ONE RCCL all-reduce per iteration.
Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 achievable


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(cfg, rowptr, col, val, target_nnz=4_000_000):
    """time the CPU oracle (single thread, fp64 restatement of the reference)
    on a bounded contiguous slice of the same workload's users"""
    from oracle import orc
    n = rowptr.shape[0] - 1
    s = int(np.searchsorted(rowptr, min(target_nnz, rowptr[-1])))
    s = max(1, min(s, n))
    nz = int(rowptr[s])
    M = orc.Model(s, cfg["m"], cfg["K"], cfg["hier"], cfg["bias"], cfg["binary"])
    M.set_csr(rowptr[: s + 1].copy(), col[:nz].copy(), None if val is None else val[:nz].copy())
    t0 = time.perf_counter()
    M.initialize(0)
    t_init = time.perf_counter() - t0
    t0 = time.perf_counter()
    M.iterate(1)
    dt = time.perf_counter() - t0
    log(f"[cpu_baseline] slice users={s} nnz={nz} init={t_init:.1f}s iterate={dt:.2f}s")
    # context only (not the contract's cpu_baseline): the same restatement with
    # step A spread over every host core (OpenMP over users, atomics on item rows)
    t0 = time.perf_counter()
    M.iterate_all_cores()
    dt_all = time.perf_counter() - t0
    log(f"[cpu_baseline] all cores ({orc.omp_threads()} threads): iterate={dt_all:.2f}s")
    return {
        "value": nz / dt, "unit": "rating-nonzeros/s", "cores": 1, "kind": "port",
        "sample": f"1 CAVI iteration of oracle/liborc.so (single thread) on the first {s} users "
                  f"({nz} nonzeros) of the same matrix, all {cfg['m']} items, K={cfg['K']}",
        "seconds": dt,
        "all_cores": {"value": nz / dt_all, "cores": orc.omp_threads(), "seconds": dt_all,
                      "note": "same slice; step A under OpenMP (atomics on item rows), expectations parallel over rows"},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink n, m, nnz (debug only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--n", type=int, default=0, help="override users (experiments)")
    ap.add_argument("--m", type=int, default=0, help="override items (experiments)")
    ap.add_argument("--nnz", type=int, default=0, help="override nonzeros (experiments)")
    ap.add_argument("--K", type=int, default=0, help="override factors (experiments)")
    ap.add_argument("--w32", action="store_true",
                    help="EXPERIMENTAL storage mode: W kept in fp32 (arithmetic/accumulators fp64); "
                         "drifts out of the 1e-4 contract after ~30 iterations -- NOT the headline configuration")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend for N>1 (nccl = RCCL; gloo only for the "
                         "single-GPU smoke test of the N>1 code path)")
    ap.add_argument("--same-device", action="store_true",
                    help="debug: put every rank on cuda:0 (use with --backend gloo)")
    args = ap.parse_args()

    # RCCL prints a version banner through C stdio on stdout (flushed at process
    # exit when stdout is a pipe).  Keep the real stdout for the ONE JSON line:
    # everything else that lands on fd 1 is sent to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from hgaprec_amd import synth
    from hgaprec_amd.capi import Hpf

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        # started as plain `python bench.py --gpus N`: become the launcher (one
        # rank per GPU); rank 0's JSON line goes to our real stdout
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, stdout=json_fd))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # HPF_BENCH_FORCE_DIST=1: take the N>1 code path (process group, bound exchange
    # tensor, all_reduce per step) even with one rank -- lets a 1-GPU box exercise
    # the RCCL calls and the stream ordering
    force_dist = os.environ.get("HPF_BENCH_FORCE_DIST") == "1"
    use_dist = world > 1 or force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    # one explicit (non-default) HIP stream carries the kernels AND orders the
    # all-reduce: torch.distributed synchronises the collective with the
    # *current* torch stream, so the library must launch on that same stream
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)

    cfg = dict(synth.CONFIGS[args.config])
    if args.scale != 1.0:
        for k in ("n", "m", "nnz"):
            cfg[k] = max(64, int(cfg[k] * args.scale))
    for k in ("n", "m", "nnz", "K"):
        if getattr(args, k):
            cfg[k] = getattr(args, k)
    custom = args.scale != 1.0 or args.w32 or any(getattr(args, k) for k in ("n", "m", "nnz", "K"))
    n_loc, m, K = cfg["n"], cfg["m"], cfg["K"]

    # ---- synthetic shard (generated on the GPU, handed over as host CSR)
    t0 = time.perf_counter()
    rowptr, col, val = synth.generate(n_loc, m, cfg["nnz"], cfg["alpha_u"], cfg["alpha_i"],
                                      seed=cfg["seed"] + 1000 * rank, device=dev,
                                      binary=cfg["binary"], item_seed=cfg["seed"])
    nnz_loc = int(rowptr[-1])
    torch.cuda.synchronize()
    log(f"[rank {rank}] generated {n_loc} x {m}, nnz={nnz_loc} in {time.perf_counter() - t0:.1f}s")

    D = Hpf(n_loc, m, K, hier=cfg["hier"], bias=cfg["bias"], binary=cfg["binary"],
            device=local_rank, stream=stream.cuda_stream, n_ranks=2 if (force_dist and world == 1) else world,
            rank=rank, n_users_total=n_loc * world, w_storage=1 if args.w32 else 0)
    xbuf = None
    if use_dist:
        xbuf = torch.zeros(D.exchange_count(), dtype=torch.float64, device=dev)
        D.bind_exchange_buffer(xbuf.data_ptr(), xbuf.numel())
        ld_x = xbuf.numel() // (m + 1)                        # [m x ld | ld]
        x_items, x_tail = xbuf[: m * ld_x], xbuf[m * ld_x:]
    t0 = time.perf_counter()
    D.upload_csr(rowptr, col, val)
    t_upload = time.perf_counter() - t0

    # ---- bench-mode initial state (counter RNG; the parity path uses MT19937)
    t0 = time.perf_counter()
    st = synth.initial_state(n_loc, K, cfg["seed"] + 17 + 1000 * rank, dev)
    D.set_state("THETA_SHAPE", st["shape"]); D.set_state("THETA_E", st["E"]); D.set_state("THETA_ELOG", st["Elog"])
    st = synth.initial_state(m, K, cfg["seed"] + 29, dev)
    D.set_state("BETA_SHAPE", st["shape"]); D.set_state("BETA_E", st["E"]); D.set_state("BETA_ELOG", st["Elog"])
    if cfg["hier"]:
        st = synth.initial_state(n_loc, K, cfg["seed"] + 31 + 1000 * rank, dev, prior_v=K)
        D.set_state("XI_E", st["E"])
        st = synth.initial_state(m, K, cfg["seed"] + 37, dev, prior_v=K)
        D.set_state("ETA_E", st["E"])
    if cfg["bias"]:
        st = synth.initial_state(n_loc, K, cfg["seed"] + 41 + 1000 * rank, dev, prior_v=m)
        D.set_state("UBIAS_E", st["E"]); D.set_state("UBIAS_ELOG", st["Elog"]); D.set_state("UBIAS_SHAPE", st["shape"])
        st = synth.initial_state(m, K, cfg["seed"] + 43, dev, prior_v=n_loc * world)
        D.set_state("IBIAS_E", st["E"]); D.set_state("IBIAS_ELOG", st["Elog"]); D.set_state("IBIAS_SHAPE", st["shape"])
    del st
    torch.cuda.empty_cache()
    log(f"[rank {rank}] upload {t_upload:.1f}s, state {time.perf_counter() - t0:.1f}s")

    def step():
        if not use_dist:
            D.iterate(1)
        else:
            # the item shape sums (m*ld doubles) are final after the item-major phi
            # pass, which runs first: their all-reduce runs on RCCL's stream while
            # the user-major pass and the user sweep run on ours; sum_u E[theta]
            # (ld doubles) follows in a second, tiny one
            D.iterate_local_items()
            w = dist.all_reduce(x_items, async_op=True)
            D.iterate_local_users()
            dist.all_reduce(x_tail)
            w.wait()
            D.iterate_global()

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        nn = torch.tensor([nnz_loc], dtype=torch.float64, device=dev)
        dist.all_reduce(nn)
        nnz_total = int(nn.item())
    else:
        nnz_total = nnz_loc

    replica_check = None
    if use_dist:
        # every rank must hold bit-identical item-side state after the same
        # all-reduced sums: a cheap end-of-run guard against an ordering race
        be = D.get_state("BETA_E")
        cs = torch.tensor([float(be.sum()), float(np.abs(be).max())], dtype=torch.float64, device=dev)
        hi, lo = cs.clone(), cs.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        replica_check = "ok" if bool(torch.equal(hi, lo)) and bool(torch.isfinite(hi).all()) else "MISMATCH"
        del be

    # the timed iterations did the work: every nonzero's phi sums to max(y, 1), so
    # the shape rows of this rank's users must hold exactly that mass (+ priors)
    self_check = None
    if n_loc * K <= 200_000_000:
        ts = D.get_state("THETA_SHAPE")
        got = float((ts - 0.3).sum())
        if cfg["bias"]:
            got += float((D.get_state("UBIAS_SHAPE") - 0.3).sum())
        want_k = float(nnz_loc) if val is None else float(np.maximum(val, 1).astype(np.float64).sum())
        del ts
        if cfg["bias"]:
            # the item-bias slot's share went to the items: bound instead of equality
            self_check = {"user_side_mass_fraction": got / want_k, "ok": bool(0.0 < got <= want_k * (1 + 1e-9))}
        else:
            self_check = {"mass_rel_err": abs(got - want_k) / want_k, "ok": bool(abs(got - want_k) / want_k < 1e-9)}
    tm = D.mean_timing(min(args.steps, 64))
    ab = D.algorithmic_bytes()
    copy_gbs = None
    if rank == 0:
        # context for the roofline: what a plain device-to-device copy reaches on
        # this GPU right now (read + write bytes / time), outside the timed region
        src = torch.empty(1 << 27, dtype=torch.float64, device=dev)        # 1 GiB
        dst = torch.empty_like(src)
        src.fill_(1.0)
        dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dst.copy_(src)
        e1.record()
        e1.synchronize()
        copy_gbs = 5 * 2 * src.numel() * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst
    if rank == 0:
        # dominant kernel of the iteration and its HBM roofline position
        kern = "phi_item" if tm["phi_item_ms"] >= tm["phi_user_ms"] else "phi_user"
        kms = tm[kern + "_ms"]
        kname, kbytes = f"phi_pass_kernel ({kern} pass)", ab[kern]
        if kms == 0:
            # launch-bound workload: hpf_iterate replayed the iteration as one
            # hipGraph, so only the whole iteration is timed
            kms = tm["iteration_ms"]
            kname, kbytes = "whole iteration (hipGraph replay)", ab["phi_user"] + ab["phi_item"] + ab["rows"]
        achieved = kbytes / (kms * 1e-3) / 1e9
        traffic = None
        tf = ROOT / "profiles" / "traffic.json"
        if tf.exists() and not custom:
            try:
                traffic = json.loads(tf.read_text()).get(f"{args.config}:{kern}")
            except Exception:
                traffic = None
        out = {
            "metric": "rating-nonzeros/sec per CAVI iter (K=%d)" % K,
            "value": nnz_total * args.steps / dt,
            "unit": "rating-nonzeros/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64 arithmetic, W stored f32 (opt-in mode)" if args.w32 else "f64", "data": "synthetic",
            "config": {
                "workload": f"{args.config}: synthetic power-law ratings, {n_loc} users x {m} items "
                            f"and {nnz_loc} nonzeros per GPU, K={K}, "
                            + " ".join(f for f, on in (("-hier", cfg["hier"]), ("-bias", cfg["bias"]),
                                                       ("-binary-data", cfg["binary"])) if on),
                "users_per_gpu": n_loc, "items": m, "nnz_per_gpu": nnz_loc, "nnz_total": nnz_total,
                "K": K, "sharding": "users (contiguous ranges); items replicated; 1 all-reduce/iter"
                if world > 1 else "single GPU",
            },
            "roofline": {
                "bound": "hbm", "kernel": kname,
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": kbytes, "avg_launch_ms": kms,
                "hbm_copy_measured_GBps": copy_gbs,
            },
            "kernels_ms": {k: round(v, 4) for k, v in tm.items() if k.endswith("_ms")},
            "replica_check": replica_check, "self_check": self_check,
            "iteration_algorithmic_GBps": (ab["phi_user"] + ab["phi_item"] + ab["rows"]) / (dt / args.steps) / 1e9,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, rowptr, col, val)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    D.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
