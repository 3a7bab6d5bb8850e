#!/bin/bash
# Phi-pass time against the size of the matrix the gathers read (run ON the GPU box):
#   bash tools/size_sweep.sh <outdir>
# C2's 5e7 nonzeros and K=100 with n or m overridden, so that the gathered matrix
# (768 B per row) sits in one XCD's L2 (4 MiB), in the Infinity Cache (256 MiB) or in HBM.
# Row-major passes (HPF_TILE=0): this is the ceiling the gathers of an untiled pass meet, the
# reason for tiling; the last column repeats the run with the library's own choice of tiles.
OUT=${1:-gpurun_out/size_sweep}
mkdir -p $OUT
B="python bench.py --config C2 --steps 6 --warmup 2 --no-cpu-baseline"
run() { name=$1; shift
  HPF_EXPERIMENTAL=1 HPF_TILE=0 $B "$@" > $OUT/$name.json 2> $OUT/$name.log
  $B "$@" > $OUT/${name}_tiled.json 2> $OUT/${name}_tiled.log
  python - "$OUT/$name.json" "$name" "$OUT/${name}_tiled.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); t = json.load(open(sys.argv[3]))
    k, w, kt, wt = d["kernels_ms"], d["work"], t["kernels_ms"], t["work"]
    print(f"{sys.argv[2]:>14}: user {k['phi_user_ms']:.3f}  item {k['phi_item_ms']:.3f}  sweeps {k['sweep_user_ms']:.3f}+{k['sweep_item_ms']:.3f}  iter {k['iteration_ms']:.3f}  nnz {d['config']['nnz_total']}  segs u/i {w['user_segments']}/{w['item_segments']}"
          f"  | tiles u/i {wt['tiles_user']}/{wt['tiles_item']}: user {kt['phi_user_ms']:.3f}  item {kt['phi_item_ms']:.3f}  iter {kt['iteration_ms']:.3f}")
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
run c2
run n100k --n 100000
run n250k --n 250000
run n500k --n 500000
run m20k --m 20000
run m4k --m 4000
run m1m --m 1000000
