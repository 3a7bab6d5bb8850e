#!/usr/bin/env python
"""bench.py -- rating-nonzeros/sec per CAVI iteration (steps A-F) on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full CAVI iteration (phi passes + row sweeps; report steps
excluded) over one synthetic ratings matrix that is resident in HBM before the
timed region starts (generated on the GPU, handed over with
hpf_upload_csr_device / hpf_set_state_device: no PCIe inside or before `value`).

  N = 1   BASELINE config C2 (1M x 100K, 5e7 nnz, K=100, -hier) on one GPU.
  N > 1   BASELINE config C3 (10M x 1M, 1e9 nnz, K=100, -hier), STRONG scaling:
          the same matrix whatever N is; every rank generates only its own
          nnz-balanced user range (partition_users), items are replicated, and
          the item-side shape sums (800 MB) + sum_u E[theta_u] go through one
          RCCL all-reduce per iteration, started right after the item-major
          phi pass so that it travels underneath the user-major half (and a tiny
          second one for sum_u E[theta]); --single-allreduce fuses both into one
          call after the user half, as BASELINE.json words it (no overlap).
          Rank 0 then times the WHOLE matrix on its own GPU in the same process
          (outside the timed region): `speedup_vs_1gpu_same_workload` needs no
          stored number; `rccl` says what the communicator saw.
  --weak  the round-1 mode: every rank owns a C2-sized shard (weak scaling).

Rank 0 prints one JSON line.  Beside the contract's fields (round 4):

  roofline        the dominant kernel against the HBM peak.  `frac` is the MEMORY-SIDE fraction:
                  bytes that crossed from the fabric into the L2s per launch -- read from the PMC
                  counters IN THIS RUN (two `rocprofv3 --kernel-trace --pmc` passes, FETCH_SIZE and
                  WRITE_SIZE, over `bench.py --lean --steps 3` spawned after the timed region, with
                  the guide's x 2 and its calibration on a kernel of known bytes in the same pass) --
                  / the launch time of the timed region / 8 TB/s.  The contract's ALGORITHMIC bytes
                  (every gathered row once per nonzero) / time stay beside it as `algorithmic_GBps`:
                  cache-inclusive, may exceed the peak where tiles are served by the XCDs' L2s.
  ms_per_step_median_hipevent   the median of the hipEvent-timed iterations beside the
                  wall-clock mean `value` is computed from
  self_check      mass conservation AND the values of a sample of owner rows (one more iteration,
                  recomputed in fp64 from the exported Elog arrays: tests/rowcheck.py)
  other_configs   C4 whole, what one of 8 GPUs holds of C3 and of C5: every BASELINE shape timed in
                  the driver's own run (never part of `value`)
  cpu_baseline    the CPU oracle on a slice of the same matrix, 1 thread (and all cores, for context)
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GBS = 6300.0    # float4 copy measured in the same guide (79 % of spec)
REFERENCE_PROBE = {            # the reference binary itself, survey-time probe
    "value": 0.325e6, "unit": "rating-nonzeros/s", "cores": 1, "K": 100,
    "what": "premgopalan/hgaprec vb_hier, ML-1M-shaped input, Xeon 2.1 GHz (SURVEY.md section 6; not this host)",
}


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
        "reference_probe": REFERENCE_PROBE,
        "all_cores": {"value": nz / dt_all, "cores": orc.omp_threads(), "seconds": dt_all,
                      "note": "same slice; step A under OpenMP (atomics on item rows), expectations parallel over rows"},
    }


def parse_counter_csvs(directory, ctr):
    """rocprofv3 --pmc output under `directory` -> ({side: [counter value per launch of that phi pass]},
    [counter values of materialize_es_kernel launches]); side 0 = user-major, 1 = item-major: the last template
    argument of the pass kernels (phi_pass_kernel<..., SIDE>, phi_pass_packed_kernel<..., SIDE>)"""
    import csv
    import glob
    import re
    vals = {0: [], 1: []}
    calv = []
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != ctr:
                continue
            name = row["Kernel_Name"]
            if "phi_pass" in name:
                mt = re.search(r",\s*(\d)>\(", name)
                if mt:
                    vals[int(mt.group(1))].append(float(row["Counter_Value"]))
            elif "materialize_es_kernel" in name:
                calv.append(float(row["Counter_Value"]))
    return vals, calv


def clean_profiler_env(env):
    """the environment for a rocprofv3 run of our own, when this process may itself be running under one
    (`rocprofv3 --kernel-trace --stats -- python bench.py`): the outer tool's settings (ROCPROF_*, ROCP_*,
    ROCPROFILER_*) and its libraries in LD_PRELOAD are not handed down, so that the inner passes start from
    what a shell would give them"""
    out = {k: v for k, v in env.items() if not k.startswith(("ROCPROF_", "ROCP_", "ROCPROFILER_"))}
    pre = [x for x in out.get("LD_PRELOAD", "").split(":") if x and "rocprofiler" not in x and "rocprof" not in os.path.basename(x)]
    if pre:
        out["LD_PRELOAD"] = ":".join(pre)
    else:
        out.pop("LD_PRELOAD", None)
    return out


# What the machine gives whole-row gathers (tools/mixed_gather.hip, profiles/r06/mixed_gather.json; tools/gather_ceiling.hip before it):
# 33 TB/s when every row is in the XCD's own L2, 7.1 TB/s when every row comes over the fabric (Infinity Cache or HBM alike) -- and
# for a stream of which a share f hits, NOT the faster of the two side by side but their serial sum, 1 / (f / 33 + (1 - f) / 7.1)
# (measured: at f = 0.73 15.0 TB/s, the serial model 16.2, side by side would be 26): hits and misses go through one pipeline.
L2_GATHER_CEILING_GBS = 33000.0
FABRIC_GATHER_CEILING_GBS = 7100.0


def mixed_stream_ceiling(hit):
    """GB/s the memory system gives row gathers of which a share `hit` is served by the XCD's L2 (serial model above)"""
    hit = min(max(hit, 0.0), 1.0)
    return 1.0 / (hit / L2_GATHER_CEILING_GBS + (1.0 - hit) / FABRIC_GATHER_CEILING_GBS)


def l2_side(per_sd, ms):
    """L2-side figures of one pass from its per-launch counters: hit rate, requests x 128 B / time, and that against the
    mixed-stream ceiling for this hit rate.  {} without the counters."""
    if not per_sd or "TCC_REQ_sum" not in per_sd or ms <= 0:
        return {}
    hit, miss = per_sd.get("TCC_HIT_sum", 0.0), per_sd.get("TCC_MISS_sum", 0.0)
    if hit + miss <= 0:
        return {}
    e = {"l2_hit_rate": hit / (hit + miss), "l2_side_GBps": per_sd["TCC_REQ_sum"] * 128.0 / (ms * 1e-3) / 1e9}
    e["mixed_stream_ceiling_GBps"] = mixed_stream_ceiling(e["l2_hit_rate"])
    e["frac_of_mixed_stream_ceiling"] = e["l2_side_GBps"] / e["mixed_stream_ceiling_GBps"]
    return e


def pmc_passes(argv_workload, n_rows_big, ld, tmo=300, local_rank=0, steps=3):
    """HBM-side traffic of the phi passes, measured IN THIS RUN: two `rocprofv3 --kernel-trace --pmc`
    passes (FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X guide section "rocprofv3 PMC
    slots") over `bench.py --lean --steps 3 --warmup 1` of the same workload, spawned after the
    timed region while this process is idle.  Per launch: bytes = 2 x FETCH_SIZE x 1024 / cal +
    WRITE_SIZE x 1024, the x 2 being the guide's gfx950 correction and `cal` its calibration in the
    same pass on materialize_es_kernel, which reads n_rows_big x ld doubles by construction.
    The L2's own counters ride along (round 6: the TCC block's four slots take FETCH_SIZE + TCC_HIT_sum in one pass and
    WRITE_SIZE + TCC_REQ_sum + TCC_MISS_sum in the other, checked on gfx950); a pass that refuses the group is repeated bare.
    -> ({side: {"fetch_KiB", "write_KiB", "launches"}}, cal dict, note) or (None, None, why)"""
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="hpf_pmc_", dir="/tmp")
    env = clean_profiler_env(dict(os.environ, TMPDIR="/tmp"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "HPF_BENCH_FORCE_DIST",
              "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "NCCL_DEBUG", "NCCL_DEBUG_SUBSYS", "NCCL_DEBUG_FILE"):
        env.pop(k, None)
    env["LOCAL_RANK"] = str(local_rank)          # the child runs on the GPU of the rank that spawned it
    per = {0: {}, 1: {}}
    cal = {}
    t0 = time.perf_counter()
    try:
        for gi, group in enumerate((("FETCH_SIZE", "TCC_HIT_sum"), ("WRITE_SIZE", "TCC_REQ_sum", "TCC_MISS_sum"))):
            for attempt in (group, group[:1]):
                d = os.path.join(tmp, f"g{gi}_{len(attempt)}")
                cmd = [exe, "--kernel-trace", "--pmc", *attempt, "--output-format", "csv", "-d", d, "-o", "p", "--",
                       sys.executable, str(ROOT / "bench.py"), "--lean", "--steps", str(steps), "--warmup", "1"] + argv_workload
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=tmo)
                if r.returncode == 0:
                    break
                if len(attempt) == 1:
                    return None, None, f"rocprofv3 --pmc {attempt[0]} failed (rc {r.returncode}): {r.stderr.decode(errors='replace')[-300:]}"
            for ctr in attempt:
                vals, calv = parse_counter_csvs(d, ctr)
                for sd in (0, 1):
                    v = vals[sd][1:] if len(vals[sd]) > 1 else vals[sd]        # the first launch is the warm-up iteration
                    if v:
                        per[sd][ctr] = sum(v) / len(v)
                        per[sd]["launches"] = len(v)
                if calv and ctr in ("FETCH_SIZE", "WRITE_SIZE"):
                    cal[ctr] = max(calv) * 1024.0                                # the larger side's launch
        if not all("FETCH_SIZE" in per[sd] and "WRITE_SIZE" in per[sd] for sd in (0, 1)):
            return None, None, "rocprofv3 ran but the phi kernels were not in its counter CSV"
        known_read = float(n_rows_big) * ld * 8
        c = {"fetch_x2_over_known_read": 2.0 * cal["FETCH_SIZE"] / known_read if cal.get("FETCH_SIZE") else None,
             "write_over_known_write": cal["WRITE_SIZE"] / (2.0 * known_read) if cal.get("WRITE_SIZE") else None,
             "kernel": "materialize_es_kernel: reads rows x ld doubles, writes twice that, by construction",
             "seconds": round(time.perf_counter() - t0, 1)}
        return per, c, f"this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two passes) over bench.py --lean --steps {steps} --warmup 1"
    except Exception as ex:
        return None, None, f"in-run PMC passes failed: {ex}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_child_workload(args, cname, user_range, split):
    """the workload arguments of the one-rank child the PMC passes profile: the same configuration and overrides as this
    run; `user_range` = (a, b) when this rank holds a range of a larger matrix (rank 0 of a strong-scaling run: the child
    generates exactly that range); `split`: this run drives the iteration through hpf_iterate_local_* / _global (any
    distributed run), so the child does too"""
    wl = ["--config", cname]
    for k in ("n", "m", "nnz", "K"):
        if getattr(args, k):
            wl += [f"--{k}", str(getattr(args, k))]
    if args.scale != 1.0:
        wl += ["--scale", str(args.scale)]
    if args.w48:
        wl.append("--w48")
    if args.w32:
        wl.append("--w32")
    if user_range:
        wl += ["--user-range", str(user_range[0]), str(user_range[1])]
    if split or args.split_iteration:
        wl.append("--split-iteration")
    return wl


def kernels_sha():
    h = hashlib.sha256()
    for f in ("hpf_kernels.hpp", "hpf_build.hpp"):    # the kernels and the work lists they walk: what the traffic was measured on
        h.update((ROOT / "hgaprec_amd" / "csrc" / f).read_bytes())
    return h.hexdigest()[:16]


def measured_traffic(config, kern):
    """HBM-side bytes per launch from the PMC passes kept in profiles/traffic.json
    (FETCH_SIZE x 2 + WRITE_SIZE, see profiles/README.md).  Only quoted when the
    file was measured on THIS kernel source (sha of hpf_kernels.hpp + hpf_build.hpp)."""
    tf = ROOT / "profiles" / "traffic.json"
    if not tf.exists():
        return None, "no profiles/traffic.json"
    try:
        d = json.loads(tf.read_text())
    except Exception:
        return None, "profiles/traffic.json unreadable"
    if d.get("kernels_sha") != kernels_sha():
        return None, f"stale: measured on kernels_sha {d.get('kernels_sha')}, this build is {kernels_sha()}"
    return d.get(f"{config}:{kern}"), f"PMC passes of {d.get('measured', '?')}"


def start_state(H, cfg, rows, first_row, seed_users, n_total, dev):
    """bench-mode initial state (counter hash; the parity path uses MT19937), a function of the
    GLOBAL row: shards of one problem start from one state"""
    import torch
    from hgaprec_amd import synth
    m, K = cfg["m"], cfg["K"]

    def put(names, st):
        for w, k in names:
            H.set_state_device(w, st[k])
    put((("THETA_SHAPE", "shape"), ("THETA_E", "E"), ("THETA_ELOG", "Elog")),
        synth.initial_state_device(rows, K, seed_users + 17, dev, row0=first_row))
    put((("BETA_SHAPE", "shape"), ("BETA_E", "E"), ("BETA_ELOG", "Elog")),
        synth.initial_state_device(m, K, cfg["seed"] + 29, dev))
    if cfg["hier"]:
        put((("XI_E", "E"),), synth.initial_state_device(rows, K, seed_users + 31, dev, prior_v=K, row0=first_row))
        put((("ETA_E", "E"),), synth.initial_state_device(m, K, cfg["seed"] + 37, dev, prior_v=K))
    if cfg["bias"]:
        put((("UBIAS_E", "E"), ("UBIAS_ELOG", "Elog"), ("UBIAS_SHAPE", "shape")),
            synth.initial_state_device(rows, K, seed_users + 41, dev, prior_v=m, row0=first_row))
        put((("IBIAS_E", "E"), ("IBIAS_ELOG", "Elog"), ("IBIAS_SHAPE", "shape")),
            synth.initial_state_device(m, K, cfg["seed"] + 43, dev, prior_v=n_total))
    torch.cuda.empty_cache()


def mass_check(D, cfg, nnz_loc, val, dev):
    """every nonzero's phi sums to max(y, 1): the shape rows of this handle's users must hold exactly
    that mass (+ priors).  Blind to WHICH rows were gathered -- that is what sampled_rows is for."""
    import torch
    ts = D.get_state_device("THETA_SHAPE", dev)
    got = float((ts - 0.3).sum())
    del ts
    if cfg["bias"]:
        got += float((D.get_state_device("UBIAS_SHAPE", dev) - 0.3).sum())
    want_k = float(nnz_loc) if val is None else float(torch.clamp(val, min=1).to(torch.float64).sum())
    if cfg["bias"]:
        # the item-bias slot's share went to the items: bound instead of equality
        return {"user_side_mass_fraction": got / want_k, "ok": bool(0.0 < got <= want_k * (1 + 1e-9))}
    return {"mass_rel_err": abs(got - want_k) / want_k, "ok": bool(abs(got - want_k) / want_k < 1e-9)}


def sampled_rows(D, cfg, rowptr, col, val, n_users, n_items):
    """value check of the handle just timed: ONE more iteration, the phi sums of a sample of owner
    rows recomputed in plain fp64 from the exported Elog arrays (tests/rowcheck.py; outside `value`)"""
    try:
        from tests import rowcheck
        r = rowcheck.check_handle(D, rowptr, col, val, bias=cfg["bias"], n_users=n_users, n_items=n_items, seed=7)
        r["what"] = ("one more iteration; raw phi sums of the sampled user and item rows (random, last, heaviest, rows cut "
                     "into several segments, rows at the heavy/light bar and at tile boundaries) recomputed in fp64 from "
                     "the exported Elog arrays vs shape - 0.3 from the device, max relative error; bar 1e-9")
        return r
    except Exception as ex:
        return {"error": str(ex), "ok": False}


def side_config(name, over, dev, local_rank, stream, steps=5, warmup=2, tiling=0, check_rows=True):
    """one of the OTHER BASELINE shapes on this GPU, outside `value`: generate, hand over, time a few
    iterations with the library's hipEvents, mass + sampled-row checks.  -> dict
    tiling = 1: hpf_config.tiling = 1, row-major work lists only (the all-HBM item pass behind roofline.hbm_only)"""
    import torch
    from hgaprec_amd import synth
    from hgaprec_amd.capi import Hpf
    cfg = dict(synth.CONFIGS[name])
    cfg.update(over)
    n, m, K = cfg["n"], cfg["m"], cfg["K"]
    t0 = time.perf_counter()
    try:
        need = (n + m) * (K + 8) * 8 * 5 + cfg["nnz"] * 40          # state + CSR/CSC/tiled copies + the sort's temporaries
        free_b, _ = torch.cuda.mem_get_info(dev)
        if need > 0.8 * free_b:
            return {"skipped": f"needs ~{need / 1e9:.0f} GB, {free_b / 1e9:.0f} GB free"}
        rowptr, col, val = synth.generate_device(n, m, cfg["nnz"], cfg["alpha_u"], cfg["alpha_i"], seed=cfg["seed"],
                                                 device=dev, binary=cfg["binary"])
        nnz = int(rowptr[-1])
        torch.cuda.empty_cache()
        D = Hpf(n, m, K, hier=cfg["hier"], bias=cfg["bias"], binary=cfg["binary"], device=local_rank,
                stream=stream.cuda_stream, n_users_total=over.get("n_users_total", n), tiling=tiling)
        D.upload_csr_device(rowptr, col, val)
        start_state(D, cfg, n, 0, cfg["seed"], over.get("n_users_total", n), dev)
        D.iterate(warmup)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        D.iterate(steps)
        torch.cuda.synchronize()
        wall_ms = (time.perf_counter() - t1) / steps * 1e3
        tm = D.mean_timing(steps)
        its = D.iteration_times(steps)
        wi = D.work_info()
        ab = D.algorithmic_bytes()
        out = {
            "workload": f"{n} users x {m} items, {nnz} nonzeros, K={K}" + (", -bias" if cfg["bias"] else "")
                        + (", -binary-data" if cfg["binary"] else "") + over.get("_what", ""),
            "value": nnz / (wall_ms * 1e-3), "ms_per_step": wall_ms,
            "ms_per_step_median_hipevent": float(np.median(its)) if its.size else None,
            "kernels_ms": {k: round(v, 4) for k, v in tm.items() if k.endswith("_ms")},
            "tiles": {"user": wi["tiles_user"], "item": wi["tiles_item"]}, "w_layout": wi["w_layout"], "ld": wi["ld"],
            "algorithmic_bytes": ab,
            "self_check": mass_check(D, cfg, nnz, val, dev),
        }
        if check_rows:
            out["self_check"]["sampled_rows"] = sampled_rows(D, cfg, rowptr, col, val, 24, 8)
            out["self_check"]["ok"] = bool(out["self_check"]["ok"] and out["self_check"]["sampled_rows"].get("ok"))
        D.close()
        del rowptr, col, val
        torch.cuda.empty_cache()
        out["seconds"] = round(time.perf_counter() - t0, 1)
        return out
    except Exception as ex:
        torch.cuda.empty_cache()
        return {"error": str(ex), "seconds": round(time.perf_counter() - t0, 1)}


def side_roofline(out, wl, n_big, local_rank, budget_s):
    """`roofline` of one of the other BASELINE shapes, from PMC passes of this run over a one-rank child that
    runs the same shape: per pass the memory-side fraction (bytes crossing the fabric into the L2s / time / HBM peak), the L2
    hit rate, and -- where most of the algorithmic bytes never cross the fabric -- the L2-SIDE figure: requests the L2s served x
    128 B / time against what the machine gives random whole-row gathers out of L2.  `binding` says which ceiling the pass is
    nearer to.  Outside `value`; `budget_s` bounds the passes (they are skipped, and say so, when it is spent)."""
    if budget_s <= 5:
        return {"skipped": "the time budget of the per-shape counter passes is spent"}
    t0 = time.perf_counter()
    per, cal, note = pmc_passes(wl, n_big, out["ld"], tmo=max(30, int(budget_s)), local_rank=local_rank, steps=2)
    if not per:
        return {"skipped": note, "seconds": round(time.perf_counter() - t0, 1)}
    fcal = cal.get("fetch_x2_over_known_read") or 1.0
    fcal = fcal if 0.9 < fcal < 1.25 else 1.0
    km, ab = out["kernels_ms"], out["algorithmic_bytes"]
    blk = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic_source": note, "fetch_calibration_applied": fcal,
           "gather_ceilings_GBps": {"all_hits": L2_GATHER_CEILING_GBS, "all_misses": FABRIC_GATHER_CEILING_GBS,
                                    "model": "hit share f: 1 / (f / all_hits + (1 - f) / all_misses) -- tools/mixed_gather.hip"},
           "seconds": None, "per_kernel": {}}
    for sd, nm in ((1, "phi_item"), (0, "phi_user")):
        ms = km.get(nm + "_ms") or 0.0
        if ms <= 0 or "FETCH_SIZE" not in per[sd]:
            continue
        traffic = int(2.0 * per[sd]["FETCH_SIZE"] * 1024.0 / fcal + per[sd]["WRITE_SIZE"] * 1024.0)
        e = {"ms": round(ms, 4), "traffic": traffic, "achieved": traffic / (ms * 1e-3) / 1e9, "algorithmic_bytes": ab[nm],
             "algorithmic_GBps": ab[nm] / (ms * 1e-3) / 1e9, "traffic_over_algorithmic": traffic / ab[nm]}
        e["frac"] = e["achieved"] / HBM_PEAK_GBS
        e.update(l2_side(per[sd], ms))
        if "l2_hit_rate" in e:
            # which ceiling the pass sits under: most of its rows crossing the fabric -> the HBM-side fraction; most of them served
            # by the L2s -> the memory system's rate for a stream with this hit share
            e["binding"] = "fabric (hbm-side frac)" if e["traffic_over_algorithmic"] >= 0.5 else "mixed L2 / fabric stream (frac_of_mixed_stream_ceiling)"
        blk["per_kernel"][nm] = e
    dom = max(blk["per_kernel"], key=lambda k: blk["per_kernel"][k]["ms"], default=None)
    if dom:
        d = blk["per_kernel"][dom]
        blk.update({"kernel": f"{dom} pass", "achieved": d["achieved"], "frac": d["frac"], "traffic": d["traffic"], "avg_launch_ms": d["ms"],
                    "frac_basis": "memory-side traffic (PMC counters of this run) / launch time / peak; per_kernel.*."
                                  "frac_of_mixed_stream_ceiling where the pass lives on its L2 hits (traffic_over_algorithmic < 0.5)"})
    blk["seconds"] = round(time.perf_counter() - t0, 1)
    return blk


# what 8 GPUs hold of C3 and C5 (the size of one nnz-balanced shard, all items), and C4 whole: every
# BASELINE shape gets a number timed in the driver's own run (never part of `value`)
OTHER_CONFIGS = (
    ("C4", "C4", {}),
    ("C3_shard_of_8", "C3", {"n": 1_250_000, "nnz": 125_000_000, "n_users_total": 10_000_000,
                             "_what": " (what one of 8 GPUs holds of C3: 1/8 of the users, all items)"}),
    ("C5_shard_of_8", "C5", {"n": 6_250_000, "nnz": 625_000_000, "n_users_total": 50_000_000,
                             "_what": " (the size of what one of 8 GPUs holds of C5: 1/8 of the users, all items)"}),
)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default=None, help="C1..C5; default C2 on one GPU, C3 on several")
    ap.add_argument("--weak", action="store_true",
                    help="N > 1: weak scaling -- every rank owns its own C2-sized user shard")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink n, m, nnz (debug only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lean", action="store_true",
                    help="the timed region and the mass check only: no gather-only probe, copy rate, 48-bit block, "
                         "other configs, PMC passes or CPU baseline (what the in-run rocprofv3 passes execute)")
    ap.add_argument("--no-pmc", action="store_true", help="do not spawn the in-run rocprofv3 --pmc passes")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the C4 / C3-shard / C5-shard blocks")
    ap.add_argument("--n", type=int, default=0, help="override users (experiments)")
    ap.add_argument("--m", type=int, default=0, help="override items (experiments)")
    ap.add_argument("--nnz", type=int, default=0, help="override nonzeros (experiments)")
    ap.add_argument("--K", type=int, default=0, help="override factors (experiments)")
    ap.add_argument("--host-handover", action="store_true",
                    help="also time hpf_upload_csr / hpf_set_state from host buffers (PCIe-inclusive set-up; never `value`)")
    ap.add_argument("--w32", action="store_true",
                    help="EXPERIMENTAL storage mode: W kept in fp32 (arithmetic/accumulators fp64); "
                         "drifts out of the 1e-4 contract after ~30 iterations -- NOT the headline configuration")
    ap.add_argument("--w48", action="store_true",
                    help="OPT-IN storage mode: W kept in 48 bits per element (top 48 bits of the fp64 value; "
                         "arithmetic/accumulators fp64), 25-30 %% shorter rows -- NOT the headline configuration")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend for N>1 (nccl = RCCL; gloo only for the "
                         "single-GPU smoke test of the N>1 code path)")
    ap.add_argument("--comm", choices=("torch", "library"), default="torch",
                    help="N > 1: who runs the exchange.  torch = torch.distributed.all_reduce on the bound exchange tensor (default, "
                         "what the SCALE runs time); library = hpf_comm_init + hpf_iterate: the library's own dlopen'ed RCCL calls on "
                         "its own communication stream -- the path `hgaprec -ngpus N -comm rccl` ships (needs one GPU per rank)")
    ap.add_argument("--single-allreduce", action="store_true",
                    help="N > 1: ONE all-reduce of [m x ld | ld] after the user half instead of the overlapped pair")
    ap.add_argument("--no-1gpu-reference", action="store_true",
                    help="N > 1: skip rank 0's same-run timing of the whole matrix on one GPU")
    ap.add_argument("--user-range", type=int, nargs=2, default=None, metavar=("A", "B"),
                    help="one rank only: generate exactly users [A, B) of the configured matrix (what a rank of a "
                         "strong-scaling run holds) -- the in-run PMC passes of an N > 1 run profile rank 0's shard with it")
    ap.add_argument("--split-iteration", action="store_true",
                    help="one rank only: run the iteration through the calls a rank of a sharded run makes (hpf_iterate_local_items, "
                         "_users, hpf_iterate_global on a handle created with n_ranks = 2; nothing is reduced) instead of hpf_iterate -- "
                         "the same kernels on the same lists as rank 0 of an N > 1 run, never a hipGraph replay")
    ap.add_argument("--same-device", action="store_true",
                    help="debug: put every rank on cuda:0 (use with --backend gloo)")
    args = ap.parse_args()
    if args.lean:
        args.no_pmc = args.no_other_configs = args.no_cpu_baseline = True

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
    from hgaprec_amd.dist import partition_users

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
    rccl_log = None
    if use_dist and args.backend == "nccl" and rank == 0:
        # what RCCL itself reports (ranks, channels, transports, algorithm / protocol of the
        # big all-reduce) goes to a file that the `rccl` block of the JSON line quotes
        rccl_log = f"/tmp/hpf_rccl_{os.getpid()}.log"
        if os.environ.get("NCCL_DEBUG", "INFO") != "INFO":
            log(f"[rank 0] NCCL_DEBUG={os.environ['NCCL_DEBUG']} overridden with INFO for the rccl block")
        os.environ["NCCL_DEBUG"] = "INFO"
        os.environ["NCCL_DEBUG_SUBSYS"] = "INIT,TUNING"
        os.environ["NCCL_DEBUG_FILE"] = rccl_log
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # rank 0 works alone for a while after the timed region (same-run one-GPU reference, PMC passes over its own
        # shard, CPU baseline) while the others wait at the last barrier: a generous collective time-out
        from datetime import timedelta
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=timedelta(minutes=60))
        else:
            dist.init_process_group(args.backend, timeout=timedelta(minutes=60))
    # one explicit (non-default) HIP stream carries the kernels AND orders the
    # all-reduce: torch.distributed synchronises the collective with the
    # *current* torch stream, so the library must launch on that same stream
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)

    strong = world > 1 and not args.weak
    if (args.user_range or args.split_iteration) and world > 1:
        raise SystemExit("--user-range / --split-iteration are for a single rank")
    cname = args.config or ("C3" if strong else "C2")
    cfg = dict(synth.CONFIGS[cname])
    if args.scale != 1.0:
        for k in ("n", "m", "nnz"):
            cfg[k] = max(64, int(cfg[k] * args.scale))
    for k in ("n", "m", "nnz", "K"):
        if getattr(args, k):
            cfg[k] = getattr(args, k)
    custom = args.scale != 1.0 or args.w32 or args.w48 or any(getattr(args, k) for k in ("n", "m", "nnz", "K"))
    m, K = cfg["m"], cfg["K"]

    # ---- synthetic shard, generated on the GPU and left there
    t0 = time.perf_counter()
    if args.user_range:
        # one rank's shard of a strong-scaling run, on its own: the same range of the same matrix (the PMC child of an
        # N > 1 run: rank 0's item pass walks exactly this work list)
        n_total = cfg["n"]
        ua, ub = args.user_range
        if not (0 <= ua < ub <= n_total):
            raise SystemExit(f"--user-range {ua} {ub}: outside [0, {n_total}]")
        deg = synth.degrees(n_total, m, cfg["nnz"], cfg["alpha_u"], cfg["seed"], dev)
        rowptr, col, val = synth.generate_device(n_total, m, cfg["nnz"], cfg["alpha_u"], cfg["alpha_i"],
                                                 seed=cfg["seed"], device=dev, binary=cfg["binary"],
                                                 user_range=(ua, ub), deg=deg)
        del deg
        row0, state_seed_shift = ua, 0
        custom = True
    elif strong:
        # ONE matrix G(seed, n, m, nnz) for every N: the generator is a pure function of
        # (seed, user, draw), so each rank builds exactly its user range of it
        n_total = cfg["n"]
        deg = synth.degrees(n_total, m, cfg["nnz"], cfg["alpha_u"], cfg["seed"], dev)
        planned = torch.zeros(n_total + 1, dtype=torch.int64, device=dev)
        torch.cumsum(deg, 0, out=planned[1:])
        ua, ub = partition_users(planned.cpu().numpy(), world)[rank]
        del planned
        rowptr, col, val = synth.generate_device(n_total, m, cfg["nnz"], cfg["alpha_u"], cfg["alpha_i"],
                                                 seed=cfg["seed"], device=dev, binary=cfg["binary"],
                                                 user_range=(ua, ub), deg=deg)
        # every rank must have cut the same matrix the same way
        import zlib
        cdf_sig = zlib.crc32(synth.item_cdf(m, cfg["alpha_i"]).numpy().tobytes())      # the item popularity table, bit for bit
        sig = torch.stack([deg.sum(), (deg * torch.arange(1, n_total + 1, device=dev)).sum() % 1_000_000_007,
                           torch.tensor(cdf_sig, device=dev)]).to(torch.float64)
        del deg
        hi_, lo_ = sig.clone(), sig.clone()
        dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
        if not torch.equal(hi_, lo_):
            raise SystemExit("ranks disagree on the synthetic matrix: generator is not device-independent")
        row0, state_seed_shift = ua, 0
    else:
        # one C2-sized matrix per rank (weak scaling; N = 1 is plain C2)
        n_total = cfg["n"] * world
        ua, ub = 0, cfg["n"]
        rowptr, col, val = synth.generate_device(cfg["n"], m, cfg["nnz"], cfg["alpha_u"], cfg["alpha_i"],
                                                 seed=cfg["seed"] + 1000 * rank, device=dev,
                                                 binary=cfg["binary"], item_seed=cfg["seed"])
        row0, state_seed_shift = 0, 1000 * rank
    n_loc = ub - ua
    nnz_loc = int(rowptr[-1])
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    log(f"[rank {rank}] generated users [{ua}, {ub}) x {m} items, nnz={nnz_loc} in {t_gen:.1f}s")
    torch.cuda.empty_cache()             # the generator's temporaries go back to the driver: the library allocates with hipMalloc

    lib_comm = use_dist and args.comm == "library"
    if lib_comm and args.same_device and world > 1:
        raise SystemExit("--comm library: RCCL wants one GPU per rank (--same-device is for --backend gloo --comm torch)")
    # one rank taking the distributed path: n_ranks = 2 so that the handle runs the pieces a rank of several runs -- except with the
    # library's communicator, whose size is the handle's n_ranks: a real ONE-rank communicator then (hpf_iterate goes through the
    # library's collectives whenever a communicator is there)
    D = Hpf(n_loc, m, K, hier=cfg["hier"], bias=cfg["bias"], binary=cfg["binary"],
            device=local_rank, stream=stream.cuda_stream,
            n_ranks=2 if ((force_dist or args.split_iteration) and world == 1 and not lib_comm) else world,
            rank=rank, n_users_total=n_total, w_storage=1 if args.w32 else 2 if args.w48 else 0)
    xbuf = None
    if use_dist:
        xbuf = torch.zeros(D.exchange_count(), dtype=torch.float64, device=dev)
        D.bind_exchange_buffer(xbuf.data_ptr(), xbuf.numel())
        ld_x = xbuf.numel() // (m + 1)                        # [m x ld | ld]
        x_items, x_tail = xbuf[: m * ld_x], xbuf[m * ld_x:]
    t0 = time.perf_counter()
    D.upload_csr_device(rowptr, col, val)
    D.synchronize()
    t_upload = time.perf_counter() - t0

    t0 = time.perf_counter()
    ss = cfg["seed"] + state_seed_shift
    start_state(D, cfg, n_loc, row0, ss, n_total, dev)
    t_state = time.perf_counter() - t0
    log(f"[rank {rank}] device hand-over: csr {t_upload:.2f}s, state {t_state:.2f}s")

    if lib_comm:
        # the 128-byte RCCL id from rank 0 over the process group (any backend), then the collective hpf_comm_init
        box = [Hpf.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        D.comm_init(bytes(box[0]))

    def step(single=args.single_allreduce):
        if args.split_iteration and not use_dist:
            D.iterate_local_items()
            D.iterate_local_users()
            D.iterate_global()
        elif not use_dist:
            D.iterate(1)
        elif lib_comm:
            # what `hgaprec -ngpus N -comm rccl [-single-allreduce]` runs (hgaprec_main.cpp Driver::iterate): hpf_iterate = item pass,
            # hpf_allreduce_items_begin on the library's communication stream, user half, hpf_allreduce_exchange (the tail + the
            # wait for both), item sweep; or the pieces around ONE hpf_allreduce_exchange on the kernels' stream
            if single:
                D.iterate_local_items()
                D.iterate_local_users()
                D.allreduce_exchange()
                D.iterate_global()
            else:
                D.iterate(1)
        else:
            # the item shape sums (m*ld doubles) are final after the item-major phi
            # pass, which runs first: their all-reduce runs on RCCL's stream while
            # the user-major pass and the user sweep run on ours; sum_u E[theta]
            # (ld doubles) follows in a second, tiny one.  The item update then
            # consumes the NEW theta sums, like hgaprec.cc:1380-1386.
            D.iterate_local_items()
            if single:
                D.iterate_local_users()
                dist.all_reduce(xbuf)
            else:
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
    tm = D.mean_timing(min(args.steps, 64))
    it_ms = D.iteration_times(min(args.steps, 64))          # hipEvents on the handle's stream, one figure per timed iteration
    per_rank = None
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        nn = torch.tensor([nnz_loc], dtype=torch.float64, device=dev)
        dist.all_reduce(nn)
        nnz_total = int(nn.item())
        keys = ("phi_item_ms", "combine_item_ms", "phi_user_ms", "combine_user_ms", "sweep_user_ms",
                "exchange_wait_ms", "sweep_item_ms", "iteration_ms")
        mine = torch.tensor([float(nnz_loc), float(n_loc)] + [tm[k] for k in keys], dtype=torch.float64, device=dev)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [dict(rank=r, nnz=int(v[0]), users=int(v[1]), **{k: round(float(x), 3) for k, x in zip(keys, v[2:])})
                    for r, v in enumerate(allr)]
    else:
        nnz_total = nnz_loc

    modes_ms = None
    if use_dist:
        # both exchange orders in the same run, outside `value` (the first hardware run decides the default with data): a few more
        # iterations each way, max over ranks of the wall time per iteration, and what the stream had to wait for
        modes_ms = {}
        k2 = max(2, min(args.steps, 5))
        for nm, sg in (("pair_overlapped", False), ("single_fused", True)):
            step(sg)
            fence()
            t1 = time.perf_counter()
            for _ in range(k2):
                step(sg)
            fence()
            tt = torch.tensor([(time.perf_counter() - t1) / k2 * 1e3, D.mean_timing(k2)["exchange_wait_ms"]], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            modes_ms[nm] = {"ms_per_step": round(float(tt[0]), 4), "exposed_allreduce_ms_max": round(float(tt[1]), 4)}
        modes_ms["timed_region_used"] = "single_fused" if args.single_allreduce else "pair_overlapped"
        modes_ms["iterations_each"] = k2

    replica_check = None
    if use_dist:
        # every rank must hold bit-identical item-side state after the same
        # all-reduced sums: a cheap end-of-run guard against an ordering race
        be = D.get_state_device("BETA_E", dev)
        cs = torch.stack([be.sum(), be.abs().max()])
        hi, lo = cs.clone(), cs.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        replica_check = "ok" if bool(torch.equal(hi, lo)) and bool(torch.isfinite(hi).all()) else "MISMATCH"
        beta_checksum = [float(cs[0]), float(cs[1])]                  # sum and max of BETA_E after the last iteration (comparing runs)
        del be

    rccl = None
    if use_dist:
        # what the communicator really was: ranks RCCL saw, its version, and
        # the big all-reduce on its own -- same tensor size, nothing else on the GPU
        scratch = torch.zeros_like(x_items)
        dist.all_reduce(scratch)
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        e0.record()
        for _ in range(reps):
            dist.all_reduce(scratch)
        e1.record()
        e1.synchronize()
        ar_ms = torch.tensor([e0.elapsed_time(e1) / reps], dtype=torch.float64, device=dev)
        dist.all_reduce(ar_ms, op=dist.ReduceOp.MAX)
        ar_ms = float(ar_ms.item())
        nbytes = scratch.numel() * 8
        del scratch
        ver = None
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            pass
        rccl = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "version": ver,
                "item_allreduce_bytes": nbytes, "item_allreduce_alone_ms": round(ar_ms, 3),
                # ring bus bandwidth, 2 (N-1)/N x bytes / time (the per-link figure xGMI is judged by)
                "bus_GBps": round(2 * (world - 1) / max(world, 1) * nbytes / (ar_ms * 1e-3) / 1e9, 1) if world > 1 else None,
                "mode": "one fused all-reduce after the user half" if args.single_allreduce
                        else "item sums all-reduced underneath the user half + a tail of ld doubles",
                # who issued the collectives of the TIMED region: torch.distributed on torch's stream, or the library's own
                # dlopen'ed RCCL calls on its communication stream (hpf_comm_init + hpf_iterate: what hgaprec -comm rccl runs)
                "path": "library" if lib_comm else "torch"}
        if rccl_log and rank == 0:
            try:
                txt = Path(rccl_log).read_text(errors="replace").splitlines()
                pick = lambda pat: [ln.split("NCCL INFO", 1)[-1].strip() for ln in txt if pat in ln]
                rccl["log"] = {
                    "init": (pick("Init COMPLETE") or pick("comm 0x"))[:2],
                    "channels": pick("Connected all rings")[:1] + pick("Channel 00")[:2],
                    # the line RCCL's tuner prints for the largest message it saw
                    "allreduce_tuning": sorted(set(pick("AllReduce:")), key=len)[-2:],
                    "lines": len(txt)}
            except Exception as ex:
                rccl["log"] = {"error": str(ex)}

    # the timed iterations did the work: mass conservation on the handle that was timed ...
    self_check = mass_check(D, cfg, nnz_loc, val, dev)
    if args.lean and m > n_loc:
        # the PMC passes calibrate FETCH_SIZE / WRITE_SIZE on materialize_es_kernel over the LARGER side (known bytes):
        # the mass check above exported the user side only
        del_me = D.get_state_device("BETA_SHAPE", dev)
        del del_me
    ab = D.algorithmic_bytes()
    wi = D.work_info()
    # the ceiling of each phi pass's ACCESS PATTERN on this GPU, measured now: the same work list,
    # index stream and rows with the arithmetic taken out (hpf_gather_only).  Outside the timed region.
    gather_only = None
    if not args.lean:
        try:
            gather_only = {"phi_user": D.gather_only_ms(0, 3), "phi_item": D.gather_only_ms(1, 3)}
        except Exception as ex:
            gather_only = {"error": str(ex)}
    # ... and the VALUES of a sample of its rows (one more iteration, outside the timed region): a gather of
    # the wrong row conserves mass; this does not pass it
    if world == 1 and not force_dist and not args.lean:
        self_check["sampled_rows"] = sampled_rows(D, cfg, rowptr, col, val, 48, 16)
        self_check["ok"] = bool(self_check["ok"] and self_check["sampled_rows"].get("ok"))

    handover = {"generate_s": round(t_gen, 3), "upload_csr_device_s": round(t_upload, 3),
                "set_state_device_s": round(t_state, 3)}
    free_b, total_b = torch.cuda.mem_get_info(dev)
    second_handle_bytes = (n_loc + m) * (K + 2) * 8 * 4 + nnz_loc * 12          # a whole second handle beside the first
    if args.host_handover and rank == 0 and second_handle_bytes > 0.8 * free_b:
        handover["host_handover_skipped"] = "a second resident handle does not fit beside the first"
    elif args.host_handover and rank == 0:
        # PCIe-inclusive set-up for a caller that holds host buffers (never part of `value`)
        rp_h, col_h = rowptr.cpu().numpy(), col.cpu().numpy().view(np.uint32)
        val_h = None if val is None else val.cpu().numpy()
        D2 = Hpf(n_loc, m, K, hier=cfg["hier"], bias=cfg["bias"], binary=cfg["binary"], device=local_rank)
        t0 = time.perf_counter()
        D2.upload_csr(rp_h, col_h, val_h)
        handover["upload_csr_host_s"] = round(time.perf_counter() - t0, 3)
        a = np.random.default_rng(1).random((n_loc, K))
        t0 = time.perf_counter()
        D2.set_state("THETA_E", a)
        handover["set_state_host_GBps"] = round(a.nbytes / (time.perf_counter() - t0) / 1e9, 2)
        t0 = time.perf_counter()
        D2.get_state("THETA_E")
        handover["get_state_host_GBps"] = round(a.nbytes / (time.perf_counter() - t0) / 1e9, 2)     # into a fresh array: first touch of its pages included
        t0 = time.perf_counter()
        D2.get_state("THETA_E", out=a)
        handover["get_state_host_touched_GBps"] = round(a.nbytes / (time.perf_counter() - t0) / 1e9, 2)   # into memory that has been written before
        # the same into / out of page-locked memory (hpf_host_alloc, C-ABI v6): the DMA's own target, no staging copy
        from hgaprec_amd.capi import pinned_empty
        pin = pinned_empty((n_loc, K))
        pin[...] = a
        t0 = time.perf_counter()
        D2.set_state("THETA_E", pin)
        handover["set_state_pinned_GBps"] = round(a.nbytes / (time.perf_counter() - t0) / 1e9, 2)
        t0 = time.perf_counter()
        D2.get_state("THETA_E", out=pin)
        handover["get_state_pinned_GBps"] = round(a.nbytes / (time.perf_counter() - t0) / 1e9, 2)
        del pin
        D2.close()
        del rp_h, col_h, val_h, a

    copy_gbs = None
    if rank == 0 and not args.lean:
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

    out = None
    if rank == 0:
        flags = " ".join(f for f, on in (("-hier", cfg["hier"]), ("-bias", cfg["bias"]),
                                         ("-binary-data", cfg["binary"])) if on)
        if strong:
            workload = (f"{cname}: synthetic power-law ratings, {n_total} users x {m} items, {nnz_total} nonzeros "
                        f"in total, K={K}, {flags}; users sharded over {world} GPUs by nonzeros (strong scaling)")
        else:
            workload = (f"{cname}: synthetic power-law ratings, {n_loc} users x {m} items and {nnz_loc} nonzeros "
                        f"per GPU, K={K}, {flags}")
        out = {
            "metric": "rating-nonzeros/sec per CAVI iter (K=%d)" % K,
            "value": nnz_total * args.steps / dt,
            "unit": "rating-nonzeros/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            # the median of the hipEvent-timed iterations, beside the wall-clock mean
            # `value` is computed from (this rank's handle; N > 1: rank 0's)
            "ms_per_step_median_hipevent": float(np.median(it_ms)) if it_ms.size else None,
            "ms_per_step_hipevent_min_max": [float(it_ms.min()), float(it_ms.max())] if it_ms.size else None,
            "higher_is_better": True, "scaling": "strong" if strong else ("weak" if world > 1 else "none"), "vs_baseline": None,
            "dtype": "f64 arithmetic, W stored f32 (opt-in mode)" if args.w32 else
                     "f64 arithmetic, W stored in 48 bits (opt-in mode)" if args.w48 else "f64", "data": "synthetic",
            "config": {
                "workload": workload,
                "users_total": n_total, "users_per_gpu": n_loc, "items": m, "nnz_per_gpu": nnz_loc,
                "nnz_total": nnz_total, "K": K,
                "sharding": ("users (contiguous ranges balanced by nonzeros); items replicated; the item shape sums "
                             f"({m * ld_x * 8 / 1e6:.0f} MB) all-reduced once per iteration underneath the user half")
                if use_dist else "single GPU",
            },
            "kernels_ms": {k: round(v, 4) for k, v in tm.items() if k.endswith("_ms")},
            "work": {k: wi[k] for k in ("user_segments", "item_segments", "user_long_rows", "item_long_rows",
                                        "item_huge_rows", "phi_G", "phi_R", "phi_V", "sweep_G", "sweep_R", "ld", "w_layout",
                                        "tiles_user", "tiles_item", "tile_rows_user", "tile_rows_item",
                                        "heavy_min_nnz_user", "heavy_min_nnz_item", "w_fallbacks", "notes")},
            "handover": handover,
            "replica_check": replica_check, "self_check": self_check,
            "beta_e_checksum": beta_checksum if use_dist else None,
            # (B_phi + B_rows of SURVEY.md 8d) / step time.  NOT an HBM figure: the user
            # pass's share is served by caches, so this can sit above what HBM delivers.
            "iteration_cache_inclusive_algorithmic_GBps":
                (ab["phi_user"] + ab["phi_item"] + ab["rows"]) / (dt / args.steps) / 1e9,
        }
        if per_rank is not None:
            comp = [r["phi_item_ms"] + r["combine_item_ms"] + r["phi_user_ms"] + r["combine_user_ms"]
                    + r["sweep_user_ms"] + r["sweep_item_ms"] for r in per_rank]
            out["per_rank"] = per_rank
            out["exposed_allreduce_ms"] = {"max": max(r["exchange_wait_ms"] for r in per_rank),
                                           "mean": sum(r["exchange_wait_ms"] for r in per_rank) / world}
            out["compute_ms"] = {"max": max(comp), "min": min(comp)}
            out["allreduce_modes_ms"] = modes_ms
        if rccl is not None:
            out["rccl"] = rccl
        if (world == 1 and not custom and not force_dist and not args.lean
                and (n_loc + m) * wi["ld"] * 8 * 5 + nnz_loc * 12 < 0.4 * free_b):
            # context, never `value`: the same workload with W stored in 48 bits (hpf_config.w_storage = 2,
            # opt-in): the passes are bound by bytes per gathered row, this is what shorter rows buy
            try:
                D2 = Hpf(n_loc, m, K, hier=cfg["hier"], bias=cfg["bias"], binary=cfg["binary"], device=local_rank,
                         stream=stream.cuda_stream, w_storage=2)
                D2.upload_csr_device(rowptr, col, val)
                start_state(D2, cfg, n_loc, row0, ss, n_total, dev)
                D2.iterate(2)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                D2.iterate(5)
                torch.cuda.synchronize()
                ms48 = (time.perf_counter() - t0) / 5 * 1e3
                tm48 = D2.mean_timing(5)
                w2 = D2.work_info()
                D2.close()
                out["w48_opt_in"] = {
                    "what": "same workload, W stored as the top 48 bits of its fp64 value (w_storage = 2; arithmetic fp64); "
                            "NOT the headline configuration: drift vs the fp64 oracle ~3e-11 after 5 sweeps, ~1e-6 after 150 "
                            "(tests/w32_error_growth.py; contract 1e-4)",
                    "value": nnz_loc / (ms48 * 1e-3), "ms_per_step": ms48,
                    "kernels_ms": {k: round(v, 4) for k, v in tm48.items() if k.endswith("_ms")},
                    "row_bytes": w2["phi_G"] * w2["phi_R"] * 16,
                    "default_row_bytes": wi["phi_G"] * wi["phi_R"] * (16 if wi["w_layout"] else 8 * wi["phi_V"])}
            except Exception as ex:
                out["w48_opt_in"] = {"error": str(ex)}
    cpu_slice = None
    cpu_note = ""
    if rank == 0 and not args.no_cpu_baseline and n_loc > 0 and nnz_loc > 0:
        # N = 1: ~4 M nonzeros of C2 (10 s of one core).  Strong scaling (C3 / C5): a contiguous 1 % USER slice of the whole
        # matrix, taken from the front of rank 0's range, capped at 10 M nonzeros (~25 s at K = 100)
        target = 4_000_000
        if strong:
            target = min(10_000_000, max(1, int(rowptr[min(n_loc, max(1, n_total // 100))].item())))
            cpu_note = (f"; a contiguous ~1 % user slice of the whole {n_total}-user matrix from the front of rank 0's range "
                        f"[{ua}, {ub}), all {m} items replicated as on every rank (their sweep is part of the time)")
        s_users = int(torch.searchsorted(rowptr, torch.tensor(target, device=dev)).item()) + 1
        s_users = min(s_users, n_loc)
        nz = int(rowptr[s_users])
        cpu_slice = (rowptr[: s_users + 1].cpu().numpy(), col[:nz].cpu().numpy().view(np.uint32),
                     None if val is None else val[:nz].cpu().numpy())
    D.close()
    one_gpu = None
    if rank == 0 and strong and not args.no_1gpu_reference:
        # the SAME matrix, whole, on this rank's GPU, timed by the same clock in the same
        # process: the strong-scaling ratio then rests on nothing stored
        try:
            del rowptr, col, val
            xbuf = x_items = x_tail = None
            torch.cuda.empty_cache()
            need = (n_total + m) * wi["ld"] * 8 * 4 + cfg["nnz"] * 12
            free_b, _ = torch.cuda.mem_get_info(dev)
            if need > 0.85 * free_b:
                raise RuntimeError(f"whole matrix needs ~{need / 1e9:.0f} GB, {free_b / 1e9:.0f} GB free")
            rp1, c1, v1 = synth.generate_device(n_total, m, cfg["nnz"], cfg["alpha_u"], cfg["alpha_i"],
                                                seed=cfg["seed"], device=dev, binary=cfg["binary"])
            D1 = Hpf(n_total, m, K, hier=cfg["hier"], bias=cfg["bias"], binary=cfg["binary"], device=local_rank,
                     stream=stream.cuda_stream, w_storage=1 if args.w32 else 2 if args.w48 else 0)
            D1.upload_csr_device(rp1, c1, v1)
            nnz1 = int(rp1[-1])
            del rp1, c1, v1
            start_state(D1, cfg, n_total, 0, cfg["seed"], n_total, dev)
            steps1 = max(2, min(args.steps, 5))
            D1.iterate(2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            D1.iterate(steps1)
            torch.cuda.synchronize()
            ms1 = (time.perf_counter() - t0) / steps1 * 1e3
            tm1 = D1.mean_timing(steps1)
            D1.close()
            one_gpu = {
                "value": ms1 / out["ms_per_step"], "one_gpu_ms_per_step": ms1, "one_gpu_nnz": nnz1,
                "one_gpu_kernels_ms": {k: round(v, 3) for k, v in tm1.items() if k.endswith("_ms")},
                "source": f"same run: rank 0 timed {steps1} iterations of the whole matrix on its own GPU after the "
                          "timed region (the other ranks wait)"}
        except Exception as ex:
            one_gpu = {"value": None, "source": f"same-run reference failed: {ex}"}
        out["speedup_vs_1gpu_same_workload"] = one_gpu
    elif rank == 0:
        del rowptr, col, val
        torch.cuda.empty_cache()

    hbm_only_run = None
    if rank == 0:
        # ---- the other BASELINE shapes on this GPU (outside `value`; a few seconds each)
        if world == 1 and not custom and not force_dist and not args.no_other_configs and cname == "C2":
            out["other_configs"] = {"note": "outside `value`: other BASELINE shapes timed in this same run on this GPU "
                                            "(5 iterations after 2 warm-up; wall clock and the library's hipEvents)"}
            pmc_budget = 0.0 if args.no_pmc else float(os.environ.get("HPF_BENCH_SIDE_PMC_SECONDS", "150"))
            for label, base, over in OTHER_CONFIGS:
                oc = side_config(base, dict(over), dev, local_rank, stream)
                out["other_configs"][label] = oc
                log(f"[other_configs] {label}: {oc.get('ms_per_step')} ms/step ({oc.get('seconds')} s)")
                if oc.get("kernels_ms") and not args.no_pmc:
                    # the same shape in a one-rank child under the counters (sizes as overrides of the base config)
                    wl = ["--config", base] + [x for k in ("n", "nnz") if k in over for x in (f"--{k}", str(over[k]))]
                    c2 = dict(synth.CONFIGS[base]); c2.update(over)
                    oc["roofline"] = side_roofline(oc, wl, max(c2["n"], c2["m"]), local_rank, pmc_budget)
                    pmc_budget -= oc["roofline"].get("seconds") or 0.0
                    log(f"[other_configs] {label} roofline: {json.dumps(oc['roofline'].get('per_kernel', oc['roofline']))[:400]}")
            # the dominant kernel where NOTHING it gathers can be cache-resident, measured in this run (round 5; a stored
            # figure until round 4): whole C3 on this GPU with row-major work lists (hpf_config.tiling = 1) -- its item pass
            # gathers 10^9 rows out of an 8 GB user matrix, so its algorithmic bytes ARE its HBM bytes
            hb = side_config("C3", {"_what": " (whole C3, row-major work lists: the all-HBM item pass)"}, dev, local_rank, stream,
                             steps=3, warmup=1, tiling=1, check_rows=False)
            if hb.get("kernels_ms", {}).get("phi_item_ms"):
                hb_gbs = hb["algorithmic_bytes"]["phi_item"] / (hb["kernels_ms"]["phi_item_ms"] * 1e-3) / 1e9
                hbm_only_run = {"GBps": hb_gbs, "phi_item_ms": hb["kernels_ms"]["phi_item_ms"], "ms_per_step": hb["ms_per_step"],
                                "mass_ok": hb["self_check"]["ok"], "seconds": hb.get("seconds"),
                                "what": "whole C3 (10^7 users x 10^6 items, 10^9 nonzeros, K=100) on this GPU, row-major work lists "
                                        "(hpf_config.tiling = 1), 3 iterations: algorithmic bytes of the item pass / its hipEvent time"}
            else:
                hbm_only_run = {"skipped": hb.get("skipped") or hb.get("error")}
            log(f"[hbm_only] {hbm_only_run}")

        # ---- roofline of the dominant kernel.  `frac` is the MEMORY-SIDE fraction: bytes that crossed from
        # the fabric into the XCDs' L2s per launch (PMC counters read in this run) / launch time / HBM peak.
        kern = "phi_item" if tm["phi_item_ms"] >= tm["phi_user_ms"] else "phi_user"
        kms = tm[kern + "_ms"]
        kname = {0: "phi_pass_kernel", 2: "phi_pass_packed_kernel<codec_f48>", 3: "phi_pass_packed_kernel<codec_p59>",
                 4: "phi_pass_packed_kernel<codec_f64>"}.get(wi["w_layout"], "phi pass")
        kname, kbytes = f"{kname} ({kern} pass)", ab[kern]
        graph = kms == 0
        if graph:
            # launch-bound workload: hpf_iterate replayed the iteration as one
            # hipGraph, so only the whole iteration is timed
            kms = tm["iteration_ms"]
            kname, kbytes = "whole iteration (hipGraph replay)", ab["phi_user"] + ab["phi_item"] + ab["rows"]
        alg_gbs = kbytes / (kms * 1e-3) / 1e9
        traffic = None
        pmc = None
        traffic_note = "hipGraph replay: no per-kernel times" if graph else "--lean / --no-pmc"
        if not args.no_pmc and not graph:
            # N > 1 (and HPF_BENCH_FORCE_DIST): the child is ONE rank that generates exactly rank 0's user range of the
            # same matrix and runs whole iterations on it -- the item pass walks the work list rank 0 just timed; the
            # other ranks wait at the last barrier, rank 0's GPU is otherwise idle
            wl = pmc_child_workload(args, cname, (ua, ub) if (strong or args.user_range) else None, use_dist)
            per, cal, traffic_note = pmc_passes(wl, max(n_loc, m), wi["ld"], local_rank=local_rank)
            if strong and per:
                traffic_note += f" --user-range {ua} {ub} (rank 0's shard, one rank on rank 0's GPU)"
            if per:
                fcal = cal.get("fetch_x2_over_known_read") or 1.0
                fcal = fcal if 0.9 < fcal < 1.25 else 1.0                 # a calibration outside that band is not one
                tr = {}
                for sd, nm in ((0, "phi_user"), (1, "phi_item")):
                    tr[nm] = int(2.0 * per[sd]["FETCH_SIZE"] * 1024.0 / fcal + per[sd]["WRITE_SIZE"] * 1024.0)
                traffic = tr[kern]
                l2s = {nm: l2_side(per[sd], tm[nm + "_ms"]) for sd, nm in ((0, "phi_user"), (1, "phi_item"))}
                pmc = {"per_launch_bytes": tr, "l2_side": l2s, "FETCH_SIZE_KiB": {"phi_user": per[0]["FETCH_SIZE"], "phi_item": per[1]["FETCH_SIZE"]},
                       "WRITE_SIZE_KiB": {"phi_user": per[0]["WRITE_SIZE"], "phi_item": per[1]["WRITE_SIZE"]},
                       "launches_averaged": per[1].get("launches"), "calibration": cal, "fetch_calibration_applied": fcal,
                       "formula": "bytes = 2 x FETCH_SIZE x 1024 / fetch_calibration + WRITE_SIZE x 1024 (gfx950: FETCH_SIZE counts 128-B "
                                  "requests at 64 B; Infinity-Cache hits are included: no DRAM-side counter exists)"}
        if traffic is None and not custom and world == 1 and not force_dist:
            stored, note2 = measured_traffic(cname, kern)            # labelled fallback: a stored profile of the same kernel source
            if stored:
                traffic, traffic_note = stored, f"STORED, not this run ({traffic_note}): {note2}"
        mem_gbs = traffic / (kms * 1e-3) / 1e9 if traffic else None
        # bytes the sweeps really move: read the raw sums, write W (16 B per element);
        # SURVEY.md's formula credits 32 (it assumes shape, rate, E and Elog all materialised)
        Kp = K + (1 if cfg["bias"] else 0)
        rows_moved = (n_loc + m) * Kp * 16

        def gbs(b, ms):
            return round(b / (ms * 1e-3) / 1e9, 1) if ms > 0 else None

        per_kernel = None if graph else {
            "phi_item": {"algorithmic_bytes": ab["phi_item"], "ms": round(tm["phi_item_ms"], 4),
                         "gather_only_ms": (gather_only or {}).get("phi_item"),
                         "algorithmic_GBps": gbs(ab["phi_item"], tm["phi_item_ms"]),
                         "memory_side_bytes": pmc["per_launch_bytes"]["phi_item"] if pmc else None,
                         "memory_side_GBps": gbs(pmc["per_launch_bytes"]["phi_item"], tm["phi_item_ms"]) if pmc else None,
                         "note": "gathers rows of the user matrix (>> Infinity Cache): L2-miss fills from HBM"},
            "phi_user": {"algorithmic_bytes": ab["phi_user"], "ms": round(tm["phi_user_ms"], 4),
                         "gather_only_ms": (gather_only or {}).get("phi_user"),
                         "algorithmic_GBps": gbs(ab["phi_user"], tm["phi_user_ms"]),
                         "memory_side_bytes": pmc["per_launch_bytes"]["phi_user"] if pmc else None,
                         "memory_side_GBps": gbs(pmc["per_launch_bytes"]["phi_user"], tm["phi_user_ms"]) if pmc else None,
                         "note": "its gathers of item rows are largely served by L2 / Infinity Cache: the algorithmic figure is "
                                 "CACHE-INCLUSIVE and may exceed the HBM peak; the memory-side one counts Infinity-Cache hits too"},
            "sweeps": {"bytes_moved": rows_moved, "survey_formula_bytes": ab["rows"],
                       "ms": round(tm["sweep_user_ms"] + tm["sweep_item_ms"], 4),
                       "GBps_moved": gbs(rows_moved, tm["sweep_user_ms"] + tm["sweep_item_ms"]),
                       "note": "fp64-VALU-bound (digamma + exp per element), not HBM-bound"},
        }
        hbm_only, hbm_only_note = measured_traffic("C3", "phi_item_hbm_only_GBps")
        if hbm_only_run and hbm_only_run.get("GBps"):
            hbm_only, hbm_only_note = hbm_only_run["GBps"], "this run: " + hbm_only_run["what"]
        out["roofline"] = {
            "bound": "hbm", "kernel": kname,
            # memory side: what crossed the fabric into the L2s per launch / launch time.  <= peak by construction
            # of the machine; when the counters are unavailable the ALGORITHMIC rate stands in and says so
            "achieved": mem_gbs if mem_gbs else alg_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": (mem_gbs if mem_gbs else alg_gbs) / HBM_PEAK_GBS,
            "frac_basis": "memory-side traffic (PMC counters) / launch time / peak" if mem_gbs else
                          "ALGORITHMIC bytes / launch time / peak (no counters in this run): cache-inclusive, may exceed 1",
            "traffic": traffic, "traffic_source": traffic_note,
            "avg_launch_ms": kms,
            "frac_of_achievable": (mem_gbs / HBM_ACHIEVABLE_GBS) if mem_gbs else None,
            "frac_of_measured_copy": (mem_gbs / copy_gbs) if mem_gbs and copy_gbs else None,
            "hbm_copy_measured_GBps": copy_gbs,
            # the contract's ALGORITHMIC bytes (every gathered row counted once per nonzero, at its stored
            # size) / launch time.  Cache-inclusive: the tiled share of those rows is served by the XCDs' L2s and never
            # crosses the fabric, so this may exceed the HBM peak -- it is a throughput figure, not a roofline fraction
            "algorithmic_bytes_per_launch": kbytes, "algorithmic_GBps": alg_gbs, "algorithmic_over_peak": alg_gbs / HBM_PEAK_GBS,
            "traffic_over_algorithmic": (traffic / kbytes) if traffic else None,
            "pmc": pmc,
            # the same kernel where nothing it gathers can be cache-resident (whole C3, row-major): a stored profile
            "hbm_only_frac": (hbm_only / HBM_PEAK_GBS) if hbm_only else None,
            "hbm_only_source": hbm_only_note if hbm_only else None,
            "hbm_only_run": hbm_only_run,
            # the same pass with the arithmetic taken out: what the memory system needs for its gathers alone
            # the L2's side of the dominant kernel (same counter passes): hit rate, requests x 128 B / time, and that against what the
            # memory system gives a stream of row gathers with this hit share (mixed_stream_ceiling)
            "l2_side": (pmc or {}).get("l2_side", {}).get(kern),
            "gather_only_ms": (gather_only or {}).get(kern),
            "frac_of_gather_only": (gather_only[kern] / kms) if gather_only and kern in gather_only and kms > 0 else None,
            "w_rows": {0: "plain fp64", 2: "48-bit (opt-in, lossy)", 3: "59-bit packed fp64 (lossless)",
                       4: "plain fp64 in 16-byte pieces (w_storage = 3, or after a fallback from the packed rows)"}.get(wi["w_layout"]),
            "w_fallbacks": wi["w_fallbacks"],
            "tiles": {"phi_item": wi["tiles_item"], "phi_user": wi["tiles_user"],
                      "note": "0 = row-major pass; >0 = heavy rows regrouped by that many tiles of the gathered rows"},
            "per_kernel": per_kernel,
        }
        if cpu_slice is not None:
            try:
                out["cpu_baseline"] = cpu_baseline(cfg, *cpu_slice, target_nnz=int(cpu_slice[0][-1]))
                out["cpu_baseline"]["sample"] += cpu_note
            except Exception as ex:                       # the line must not be lost to its context block
                out["cpu_baseline"] = {"value": None, "unit": "rating-nonzeros/s", "cores": 1, "kind": "port", "error": str(ex)}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
