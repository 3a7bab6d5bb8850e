for m in 2000 5000 20000 100000; do
  python bench.py --m $m --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('m=$m nnz', d['config']['nnz_total'], round(d['value']/1e9,3), d['kernels_ms'])"
done
