# how fast is a phi pass whose gathered matrix fits one XCD's L2?  (item pass of an n-user job)
OUTD=${OUT:-gpurun_out}/l2probe; mkdir -p $OUTD
export TMPDIR=/tmp
for n in 4096 16384; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$n -o p -- python bench.py --n $n --m 17770 --nnz 30000000 --steps 10 --warmup 3 --no-cpu-baseline > /tmp/b_$n.log 2>&1 < /dev/null
  echo "== n=$n rc=$?"; grep '^{' /tmp/b_$n.log | cut -c1-300
  f=$(find /tmp/p_$n -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then head -8 "$f" | cut -c1-200; cp "$f" $OUTD/stats_$n.csv; fi
done
