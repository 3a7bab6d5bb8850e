#!/bin/bash
# EXPERIMENT (round 5): with one wave per workgroup on the tiled sides, is the heavy-row bar (HPF_TILE_RUN, default 16) or the
# tile size (HPF_TILE_BYTES, default 4 MiB) still where round 3 put it?  ms per kernel from bench.py --lean.
OUT=gpurun_out/${1:-r05h}; mkdir -p $OUT
run() { # label, env..., -- args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env HPF_EXPERIMENTAL=1 "${envs[@]}" timeout 300 python bench.py --lean --steps 6 --warmup 2 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels_ms']
print(json.dumps({'label':'$label','env':'${envs[*]}','ms':round(d['ms_per_step'],3),'phi_user':k['phi_user_ms'],'phi_item':k['phi_item_ms'],'comb_u':k['combine_user_ms'],'comb_i':k['combine_item_ms'],'tiles':[d['work']['tiles_user'],d['work']['tiles_item']],'ok':d['self_check']['ok']}))" >> $OUT/tile_probe.jsonl
}
for cfg in "c2 --config C2" "c4 --config C4" "c5s --config C5 --user-range 0 6250000" "c3s --config C3 --user-range 0 1250000"; do
  set -- $cfg; label=$1; shift
  run $label HPF_NOTHING=1 -- "$@"
  for r in 6 8 12 24; do run $label HPF_TILE_RUN=$r -- "$@"; done
  for b in 2097152 3145728 6291456 8388608; do run $label HPF_TILE_BYTES=$b -- "$@"; done
  run $label HPF_TILE_RUN=8 HPF_TILE_BYTES=3145728 -- "$@"
  run $label HPF_NOTHING=1 -- "$@"
done
cat $OUT/tile_probe.jsonl
