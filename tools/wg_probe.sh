#!/bin/bash
# EXPERIMENT (round 5): threads per workgroup of the packed phi passes (HPF_PHI_WG) x segments per chunk (HPF_TILE_CHUNK)
# on the four shapes; ms per kernel from bench.py --lean.  Usage: bash tools/wg_probe.sh [tag]
OUT=gpurun_out/${1:-r05f}; mkdir -p $OUT
run() { # label, env..., -- args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env HPF_EXPERIMENTAL=1 "${envs[@]}" timeout 300 python bench.py --lean --steps 6 --warmup 2 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels_ms']
print(json.dumps({'label':'$label','env':'${envs[*]}','ms':round(d['ms_per_step'],3),'phi_user':k['phi_user_ms'],'phi_item':k['phi_item_ms'],'comb_u':k['combine_user_ms'],'comb_i':k['combine_item_ms'],'tiles':[d['work']['tiles_user'],d['work']['tiles_item']],'ok':d['self_check']['ok']}))" >> $OUT/wg_probe.jsonl
}
for cfg in "c2 --config C2" "c4 --config C4" "c5s --config C5 --user-range 0 6250000" "c3s --config C3 --user-range 0 1250000"; do
  set -- $cfg; label=$1; shift
  for rep in 1 2; do
    run $label HPF_PHI_WG=256 -- "$@"
    run $label HPF_PHI_WG=64 HPF_TILE_CHUNK=1 -- "$@"
    run $label HPF_PHI_WG=64 HPF_TILE_CHUNK=2 -- "$@"
    run $label HPF_PHI_WG=64 HPF_TILE_CHUNK=3 -- "$@"
    run $label HPF_PHI_WG=128 HPF_TILE_CHUNK=2 -- "$@"
    run $label HPF_PHI_WG=128 HPF_TILE_CHUNK=4 -- "$@"
  done
done
cat $OUT/wg_probe.jsonl
