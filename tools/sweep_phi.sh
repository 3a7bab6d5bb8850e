for pb in 16384 65536 1048576; do
  HPF_PHI_BLOCKS=$pb python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('blocks $pb', round(d['value']/1e9,3), d['kernels_ms'])"
done
for sm in 128 256 512; do
  HPF_SEG_MAX=$sm python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('segmax $sm', round(d['value']/1e9,3), d['kernels_ms'])"
done
HPF_PHI_CFG=8,7,2 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('8,7,2', round(d['value']/1e9,3), d['kernels_ms'])"
