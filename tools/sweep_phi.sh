python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed" 
HPF_PHI_DEPTH=2 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed"
for d in 1 2; do for cfg in "8,7,2" "16,4,2"; do
  HPF_PHI_DEPTH=$d HPF_PHI_CFG=$cfg python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('C2 depth $d cfg $cfg', round(d['value']/1e9,3), d['kernels_ms'])"
done; done
for d in 1 2; do for cfg in "16,7,2" "32,4,2"; do
  HPF_PHI_DEPTH=$d HPF_PHI_CFG=$cfg python bench.py --config C4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('C4 depth $d cfg $cfg', round(d['value']/1e9,3), d['kernels_ms'])"
done; done
