export TMPDIR=/tmp
for cfg in "HPF_TILE=0" "HPF_TILE=2"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  env HPF_EXPERIMENTAL=1 $cfg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$tag -o p -- python bench.py --n 1000000 --m 2000 --nnz 50000000 --steps 10 --warmup 3 --no-cpu-baseline > /tmp/b_$tag.log 2>&1 < /dev/null
  f=$(find /tmp/st_$tag -name '*kernel_stats.csv' | head -1)
  echo "== $cfg"; python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'codec_p59' in r['Name'] or 'combine' in r['Name'] or 'row_sweep' in r['Name']:
        print('  %-80s calls %4s avg %.3f ms' % (r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e6))
PY
  env HPF_EXPERIMENTAL=1 $cfg timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/pm_$tag -o p -- python bench.py --n 1000000 --m 2000 --nnz 50000000 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 < /dev/null
  f=$(find /tmp/pm_$tag -name '*counter_collection.csv' | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
hit=collections.Counter(); miss=collections.Counter(); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name']
    if 'codec_p59' not in k: continue
    k = 'side1' if ', 1>' in k else 'side0'
    v=float(r['Counter_Value'])
    if r['Counter_Name']=='TCC_HIT_sum': hit[k]+=v; n[k]+=1
    elif r['Counter_Name']=='TCC_MISS_sum': miss[k]+=v
for k in hit: print('  %s hit %.3g miss %.3g  hit rate %.3f' % (k, hit[k]/n[k], miss[k]/n[k], hit[k]/(hit[k]+miss[k])))
PY
done
