#!/bin/bash
# Profiling recipe of a round, run ON the GPU box from the repo root:
#   bash tools/profile_round.sh r03
# Writes under gpurun_out/<tag>/ ; tools/summarize_profiles.py turns that into profiles/<tag>/.
# Counters are collected in their own passes (--kernel-trace --pmc only), as the
# MI355X guide prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass.
set -u
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
BENCH="python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
# 1. per-kernel time, same command as the bench line next to it
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o c2 -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.log
# 2. counters, one pass each
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"; do
  name=$(echo $pmc | tr ' ' '_' | tr 'A-Z' 'a-z' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pmc_$name -o c2 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_$name.log
done
# 3. the plain bench line (with the CPU baseline)
python bench.py --host-handover > $OUT/bench.json 2> $OUT/bench.log
# 4. XCD-locality probe: times, then L2 hit rate and fabric bytes of base vs xcd
for v in base xcd m2000; do python tools/xcd_locality_probe.py $v 10 > $OUT/xcd_$v.json 2> $OUT/xcd_$v.log; done
for v in base xcd; do
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/xcd_pmc_tcc_$v -o p -- python tools/xcd_locality_probe.py $v 3 > /dev/null 2> $OUT/xcd_pmc_tcc_$v.log
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/xcd_pmc_fetch_$v -o p -- python tools/xcd_locality_probe.py $v 3 > /dev/null 2> $OUT/xcd_pmc_fetch_$v.log
done
# 4b. the tiled pass: where workgroups run, the same bench line with the tiles switched
# off, and the L2 hit rate of the item pass either way; l2probe = a pass whose gathered rows fit one L2
if [ -x /opt/rocm/bin/hipcc ]; then
  [ -x tools/xcc_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o tools/xcc_probe tools/xcc_probe.hip > /dev/null 2>&1
  [ -x tools/xcd_fabric_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/xcd_fabric_probe tools/xcd_fabric_probe.hip > /dev/null 2>&1
fi
tools/xcc_probe 200000 2000 > $OUT/xcc_probe.json 2> $OUT/xcc_probe.log
tools/xcc_probe 50000 20000 >> $OUT/xcc_probe.json 2>> $OUT/xcc_probe.log
tools/xcd_fabric_probe > $OUT/xcd_fabric_probe.json 2> $OUT/xcd_fabric_probe.log
HPF_EXPERIMENTAL=1 HPF_TILE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_c2_untiled.json 2> $OUT/bench_c2_untiled.log
HPF_EXPERIMENTAL=1 HPF_TILE=0 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc_untiled -o c2 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_tcc_untiled.log
HPF_EXPERIMENTAL=1 HPF_TILE=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_untiled -o c2 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch_untiled.log
bash tools/l2probe.sh > $OUT/l2probe.txt 2>&1
# 5. phi-pass time against the size of the gathered matrix (L2 / Infinity Cache / HBM)
bash tools/size_sweep.sh $OUT/size > $OUT/size_sweep.txt 2>&1
# 5b. what the machine gives a kernel that does nothing but random whole-row gathers
if [ -x /opt/rocm/bin/hipcc ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/gather_ceiling tools/gather_ceiling.hip > /dev/null 2>&1
  tools/gather_ceiling > $OUT/gather_ceiling.json 2> $OUT/gather_ceiling.log
fi
# 6. the other configs on one GPU; C1 and C4 WITH their CPU baseline
python bench.py --config C1 --steps 20 --warmup 5 > $OUT/bench_c1.json 2> $OUT/bench_c1.log
python bench.py --config C4 --steps 5 --warmup 2 > $OUT/bench_c4.json 2> $OUT/bench_c4.log
python bench.py --config C3 --steps 5 --warmup 2 --no-cpu-baseline --host-handover > $OUT/bench_c3_full_1gpu.json 2> $OUT/bench_c3_full_1gpu.log
HPF_EXPERIMENTAL=1 HPF_TILE=0 python bench.py --config C3 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_c3_full_1gpu_untiled.json 2> $OUT/bench_c3_full_1gpu_untiled.log
HPF_EXPERIMENTAL=1 HPF_TILE=0 python bench.py --config C4 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_c4_untiled.json 2> $OUT/bench_c4_untiled.log
python bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_c5_full_1gpu.json 2> $OUT/bench_c5_full_1gpu.log
# 7. what one of 8 GPUs would hold of C3 (1.25M users x ALL 1M items, 1.25e8 nnz): the compute side of the 8-GPU estimate
python bench.py --config C3 --n 1250000 --nnz 125000000 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_c3_shard_like_1gpu.json 2> $OUT/bench_c3_shard_like_1gpu.log
# 8. report-step operations at C2 (held-out LL, ELBO, ranking evaluation)
python tools/bench_report_step.py C2 > $OUT/report_step_c2.json 2> $OUT/report_step_c2.log
ls -la $OUT
