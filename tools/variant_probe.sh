#!/bin/bash
# EXPERIMENT (round 6): builds of the packed phi pass on the four shapes -- last round's library beside this one's, then
# waves per SIMD (HPF_PHI_WAVES), the owner's factors in LDS (HPF_PHI_OWN_LDS) and rows read by half their lanes
# (HPF_PHI_X2); ms per kernel from bench.py --lean.  Usage: bash tools/variant_probe.sh [tag] [reps]
# FOR THE RECORD (profiles/r06/experiments.md 2): the knobs exist only with profiles/r06/pass_builds_prototype.diff applied, and
# hgaprec_amd/libhpf_hip_r05.so is round 5's hpf_capi.hip (git show f06a9f0:...) built beside this one.
OUT=gpurun_out/${1:-r06}; mkdir -p $OUT
REPS=${2:-2}
run() { # label, env..., -- args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env HPF_EXPERIMENTAL=1 "${envs[@]}" timeout 300 python bench.py --lean --steps 6 --warmup 2 "$@" 2>$OUT/variant_probe.err | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
    k=d['kernels_ms']
    print(json.dumps({'label':'$label','env':'${envs[*]}','ms':round(d['ms_per_step'],3),'phi_user':k['phi_user_ms'],'phi_item':k['phi_item_ms'],'comb_u':k['combine_user_ms'],'comb_i':k['combine_item_ms'],'sweep_u':k['sweep_user_ms'],'sweep_i':k['sweep_item_ms'],'tiles':[d['work']['tiles_user'],d['work']['tiles_item']],'ok':d['self_check']['ok']}))
except Exception as ex:
    print(json.dumps({'label':'$label','env':'${envs[*]}','error':str(ex)}))" >> $OUT/variant_probe.jsonl
  tail -1 $OUT/variant_probe.jsonl
}
for cfg in "c4 --config C4" "c5s --config C5 --user-range 0 6250000" "c2 --config C2" "c3s --config C3 --user-range 0 1250000"; do
  set -- $cfg; label=$1; shift
  for rep in $(seq $REPS); do
    [ -f hgaprec_amd/libhpf_hip_r05.so ] && run $label HPF_LIB=libhpf_hip_r05.so -- "$@"
    run $label HPF_PHI_WAVES=3 -- "$@"
    run $label HPF_PHI_WAVES=4 -- "$@"
    run $label HPF_PHI_WAVES=4 HPF_PHI_OWN_LDS=1 -- "$@"
    run $label HPF_PHI_WAVES=3 HPF_PHI_OWN_LDS=1 -- "$@"
    if [ $label = c4 ] || [ $label = c2 ] || [ $label = c3s ]; then
      run $label HPF_PHI_X2=1 -- "$@"
      run $label HPF_PHI_X2=1 HPF_PHI_OWN_LDS=1 -- "$@"
    fi
  done
done
