for cfg in "8,4,4" "16,2,4" "8,7,2" "16,4,2" "32,1,4" "4,7,4"; do
  HPF_PHI_CFG=$cfg python bench.py --w32 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('w32 $cfg', round(d['value']/1e9,3), d['kernels_ms'])"
done
