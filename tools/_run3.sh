#!/bin/bash
OUT=gpurun_out/r06; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "three_pieces or bound_heldout or graph_replay or every_build" > $OUT/pytest_run3a.log 2>&1; tail -3 $OUT/pytest_run3a.log
python -m pytest tests/test_gpu_multi.py -q -m gpu -k "eight_ranks or library_comm or takes_the_distributed" > $OUT/pytest_run3b.log 2>&1; tail -3 $OUT/pytest_run3b.log
python -m pytest tests/test_gpu_cli.py -q -m gpu -k "eight_process or two_process" > $OUT/pytest_run3c.log 2>&1; tail -3 $OUT/pytest_run3c.log
tools/mixed_gather > $OUT/mixed_gather.json 2> $OUT/mixed_gather.log; cat $OUT/mixed_gather.json
python tools/bench_report_step.py C2 1250000 > $OUT/report_step_c2.json 2> $OUT/report_step_c2.log; cat $OUT/report_step_c2.json
for grp in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_REQ_sum TCC_MISS_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  rm -rf /tmp/pmc_try
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_try -o p -- python bench.py --lean --steps 2 --warmup 1 --scale 0.05 > /dev/null 2> $OUT/pmc_try.log
  echo "pmc group [$grp] rc=$? files=$(find /tmp/pmc_try -name '*counter_collection.csv' | wc -l) rows=$(find /tmp/pmc_try -name '*counter_collection.csv' -exec cat {} + | grep -c phi_pass) ctrs=$(find /tmp/pmc_try -name '*counter_collection.csv' -exec cat {} + | grep phi_pass | awk -F, '{print $(NF-3)}' | sort | uniq -c | tr '\n' ' ')" | tee -a $OUT/pmc_groups.txt
done
( time python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.log ) 2> $OUT/bench_full.time; cat $OUT/bench_full.time; python - <<'P'
import json
d=json.loads(open('gpurun_out/r06/bench_full.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['scaling'])
for k,v in d['other_configs'].items():
    if isinstance(v,dict): print(k, v.get('ms_per_step'), json.dumps(v.get('roofline'))[:1500])
P
