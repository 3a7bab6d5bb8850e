#!/bin/bash
# second GPU call of round 6
OUT=gpurun_out/r06; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "three_pieces or bound_heldout or graph_replay or rccl or overlapped or call_order" > $OUT/pytest_run2.log 2>&1
tail -5 $OUT/pytest_run2.log
run() { # label, env..., -- args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env HPF_EXPERIMENTAL=1 "${envs[@]}" timeout 300 python bench.py --lean --steps 6 --warmup 2 "$@" 2>$OUT/occ_probe.err | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
    k=d['kernels_ms']
    print(json.dumps({'label':'$label','env':'${envs[*]}','ms':round(d['ms_per_step'],3),'phi_user':k['phi_user_ms'],'phi_item':k['phi_item_ms'],'comb_u':k['combine_user_ms'],'comb_i':k['combine_item_ms'],'sweep_u':k['sweep_user_ms'],'sweep_i':k['sweep_item_ms'],'it_ms':k['iteration_ms'],'tiles':[d['work']['tiles_user'],d['work']['tiles_item']],'ok':d['self_check']['ok']}))
except Exception as ex:
    print(json.dumps({'label':'$label','env':'${envs[*]}','error':str(ex)}))" >> $OUT/occ_probe.jsonl
  tail -1 $OUT/occ_probe.jsonl
}
for cfg in "c4 --config C4" "c5s --config C5 --user-range 0 6250000" "c2 --config C2"; do
  set -- $cfg; label=$1; shift
  for rep in 1 2; do
    run $label HPF_LIB=libhpf_hip_r05.so -- "$@"
    run $label HPF_PHI_LDS_PAD=0 -- "$@"
    run $label HPF_PHI_LDS_PAD=13000 -- "$@"
    run $label HPF_PHI_LDS_PAD=20000 -- "$@"
    run $label HPF_PHI_LDS_PAD=40000 -- "$@"
  done
done
# an eighth of C4 through the calls a rank makes: eager pieces against the three graph replays
for rep in 1 2 3; do
  run c4_8th HPF_GRAPH=0 -- --config C4 --user-range 0 60970 --split-iteration
  run c4_8th HPF_GRAPH=1 -- --config C4 --user-range 0 60970 --split-iteration
done
run c1_split HPF_GRAPH=0 -- --config C1 --split-iteration
run c1_split HPF_GRAPH=1 -- --config C1 --split-iteration
# report step: held-out sets bound once
python tools/bench_report_step.py C2 1250000 > $OUT/report_step_c2.json 2> $OUT/report_step_c2.log; cat $OUT/report_step_c2.json
# which counter groups fit one pass beside FETCH_SIZE / WRITE_SIZE?
for grp in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_REQ_sum TCC_MISS_sum" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  rm -rf /tmp/pmc_try
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_try -o p -- python bench.py --lean --steps 2 --warmup 1 --scale 0.05 > /dev/null 2> $OUT/pmc_try.log
  echo "pmc group [$grp] rc=$? rows=$(cat /tmp/pmc_try/*/*counter_collection.csv 2>/dev/null | grep -c phi_pass)" | tee -a $OUT/pmc_groups.txt
done
# the distributed path on one rank: torch's collectives and the library's
HPF_BENCH_FORCE_DIST=1 MASTER_PORT=29611 python bench.py --steps 3 --warmup 1 --scale 0.02 --no-cpu-baseline --no-pmc > $OUT/bench_force_dist_torch.json 2> $OUT/bench_force_dist_torch.log; tail -c 600 $OUT/bench_force_dist_torch.json
HPF_BENCH_FORCE_DIST=1 MASTER_PORT=29612 python bench.py --steps 3 --warmup 1 --scale 0.02 --no-cpu-baseline --no-pmc --comm library > $OUT/bench_force_dist_library.json 2> $OUT/bench_force_dist_library.log; tail -c 600 $OUT/bench_force_dist_library.json; tail -3 $OUT/bench_force_dist_library.log
