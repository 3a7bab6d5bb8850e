timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8
python bench.py 2>/dev/null | grep '^{' > gpurun_out/bench_tiled.json; python -c "
import json; d=json.load(open('gpurun_out/bench_tiled.json')); r=d['roofline']
print(d['ms_per_step'], d['value']); print({k:r[k] for k in ('bound','kernel','achieved','frac','traffic','gather_only_ms','frac_of_gather_only','tiles')}); print(d['kernels_ms']); print(d['handover']); print(d['cpu_baseline'])"
