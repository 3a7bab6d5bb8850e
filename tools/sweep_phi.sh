for cfg in "16,7" "32,4" "64,2" "16,8" "32,5"; do
  HPF_SWEEP_CFG=$cfg python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('sweep $cfg', round(d['value']/1e9,3), d['kernels_ms'])"
done
