#!/bin/bash
# Profiling recipe of a round, run ON the GPU box from the repo root:
#   bash tools/profile_round.sh r04
# Writes under gpurun_out/<tag>/ ; tools/summarize_profiles.py turns that into profiles/<tag>/ (step 10 runs it on the
# box and leaves the result under gpurun_out/<tag>/profiles_<tag>/: copy that into profiles/<tag>/ and commit).
# Counters are collected in their own passes (--kernel-trace --pmc only), as the
# MI355X guide prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass.
set -u
TAG=${1:-r04}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
LEAN="python bench.py --lean --steps 3 --warmup 1"
# 1. per-kernel time: the bench's timed region under the tracer (no side blocks: their kernels would mix in)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o c2 -- python bench.py --lean --steps 10 --warmup 3 > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.log
# 2. counters, one pass each (the same command bench.py itself spawns for FETCH_SIZE / WRITE_SIZE)
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"; do
  name=$(echo $pmc | tr ' ' '_' | tr 'A-Z' 'a-z' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pmc_$name -o c2 -- $LEAN > /dev/null 2> $OUT/pmc_$name.log
done
# 3. the plain bench line: in-run PMC passes, other_configs, CPU baseline, host hand-over
python bench.py --host-handover > $OUT/bench.json 2> $OUT/bench.log
# 4. the same line with the tiles switched off, and the L2 hit rate / fabric bytes of the item pass either way
HPF_EXPERIMENTAL=1 HPF_TILE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $OUT/bench_c2_untiled.json 2> $OUT/bench_c2_untiled.log
HPF_EXPERIMENTAL=1 HPF_TILE=0 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc_untiled -o c2 -- $LEAN > /dev/null 2> $OUT/pmc_tcc_untiled.log
# 5. the other configs on one GPU, each with its own in-run PMC passes; C1 and C4 WITH their CPU baseline
python bench.py --config C1 --steps 20 --warmup 5 > $OUT/bench_c1.json 2> $OUT/bench_c1.log
python bench.py --config C4 --steps 5 --warmup 2 > $OUT/bench_c4.json 2> $OUT/bench_c4.log
python bench.py --config C3 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_c3_full_1gpu.json 2> $OUT/bench_c3_full_1gpu.log
HPF_EXPERIMENTAL=1 HPF_TILE=0 python bench.py --config C3 --steps 5 --warmup 2 --no-cpu-baseline --no-pmc > $OUT/bench_c3_full_1gpu_untiled.json 2> $OUT/bench_c3_full_1gpu_untiled.log
HPF_EXPERIMENTAL=1 HPF_TILE=0 python bench.py --config C4 --steps 5 --warmup 2 --no-cpu-baseline --no-pmc > $OUT/bench_c4_untiled.json 2> $OUT/bench_c4_untiled.log
python bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_c5_full_1gpu.json 2> $OUT/bench_c5_full_1gpu.log
# 6. what 8 GPUs would hold (compute and load balance only: nothing crosses xGMI)
for n in 2 4 8; do python tools/emulate_shards.py C3 $n > $OUT/emulate_c3_$n.json 2> $OUT/emulate_c3_$n.log; done
for n in 2 4 8; do python tools/emulate_shards.py C4 $n > $OUT/emulate_c4_$n.json 2> $OUT/emulate_c4_$n.log; done
python tools/emulate_shards.py C5 8 --sequential > $OUT/emulate_c5_8.json 2> $OUT/emulate_c5_8.log
# 7. the sweep on its own (ways of writing W) and what the machine gives random whole-row gathers
if [ -x /opt/rocm/bin/hipcc ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/sweep_probe tools/sweep_probe.hip > /dev/null 2>&1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/gather_ceiling tools/gather_ceiling.hip > /dev/null 2>&1
fi
tools/sweep_probe > $OUT/sweep_probe.json 2> $OUT/sweep_probe.log
tools/gather_ceiling > $OUT/gather_ceiling.json 2> $OUT/gather_ceiling.log
# 7a. (round 6) the same gathers with a share served by the XCD's L2 and the rest over the fabric: the ceiling of a tiled pass
if [ -x /opt/rocm/bin/hipcc ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/mixed_gather tools/mixed_gather.hip > /dev/null 2>&1
fi
tools/mixed_gather > $OUT/mixed_gather.json 2> $OUT/mixed_gather.log
# 7b. the sweep at the C3-shard shape: blocks x rows held ahead; sweep_probe_pipe2 = the same source built with -DHPF_SWEEP_PIPE=2
if [ -x /opt/rocm/bin/hipcc ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DHPF_SWEEP_PIPE=2 -o tools/sweep_probe_pipe2 tools/sweep_probe.hip > /dev/null 2>&1
fi
: > $OUT/sweep_probe_c3shard.jsonl
for b in 1024 2048 4096 8192; do
  tools/sweep_probe 1250000 $b >> $OUT/sweep_probe_c3shard.jsonl 2>> $OUT/sweep_probe.log
  [ -x tools/sweep_probe_pipe2 ] && tools/sweep_probe_pipe2 1250000 $b >> $OUT/sweep_probe_c3shard.jsonl 2>> $OUT/sweep_probe.log
done
# 8. report-step operations at C2 (held-out LL, ELBO, ranking evaluation)
python tools/bench_report_step.py C2 1250000 > $OUT/report_step_c2.json 2> $OUT/report_step_c2.log
# 9. counters for the OTHER shapes: C4 whole, and what one of 8 GPUs holds of C3 and of C5 (the first
#    1/8 of the users of the real matrix, all items) -- the same passes as step 2, per shape, under $OUT/cfg_<label>/
cfgprof() {
  label=$1; shift
  D=$OUT/cfg_$label
  mkdir -p $D
  L="python bench.py --lean --steps 3 --warmup 1 $*"
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o k -- $L > $D/bench_lean.json 2> $D/bench_lean.log
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "TCC_REQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"; do
    name=$(echo $pmc | tr ' ' '_' | tr 'A-Z' 'a-z' | cut -c1-40)
    rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $D/pmc_$name -o k -- $L > /dev/null 2> $D/pmc_$name.log
  done
  # the pass with its arithmetic taken out, in the same shape (roofline.gather_only_ms of the full line)
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-pmc $* > $D/bench.json 2> $D/bench.log
}
cfgprof c4 --config C4
cfgprof c3_shard --config C3 --user-range 0 1250000
cfgprof c5_shard --config C5 --user-range 0 6250000
# 10. gpurun copies at most 64 MiB back: the summaries are made HERE (profiles/<tag>/ of this copy of the repo, then copied
#     to $OUT/profiles_<tag>/ beside the JSON lines and the logs) and the raw rocprofv3 trees are dropped
python tools/summarize_profiles.py $TAG > $OUT/summarize.log 2>&1
mkdir -p $OUT/profiles_$TAG
cp -r profiles/$TAG/. $OUT/profiles_$TAG/
cp profiles/traffic.json profiles/c3_1gpu_reference.json $OUT/profiles_$TAG/ 2>/dev/null
rm -rf $OUT/stats $OUT/pmc_*/ $OUT/cfg_*/stats $OUT/cfg_*/pmc_*/
du -sh $OUT
ls -la $OUT
