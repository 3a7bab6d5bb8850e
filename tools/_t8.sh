timeout 1200 python -m pytest tests/test_gpu_tiled.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -5
export TMPDIR=/tmp
for cfg in "HPF_TILE=0" "HPF_TILE=2" "HPF_TILE=2 HPF_TILE_BYTES=2097152" "HPF_TILE=2 HPF_TILE_CHUNK=2" "HPF_TILE=2 HPF_TILE_CHUNK=32"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  env HPF_EXPERIMENTAL=1 $cfg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$tag -o p -- python bench.py --n 1000000 --m 2000 --nnz 50000000 --steps 10 --warmup 3 --no-cpu-baseline > /tmp/b_$tag.log 2>&1 < /dev/null
  f=$(find /tmp/st_$tag -name '*kernel_stats.csv' | head -1)
  echo "== $cfg"; grep '^{' /tmp/b_$tag.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('  ms/step %.3f' % d['ms_per_step'], d['roofline'].get('gather_only_ms'))"
  grep "phi_pass_packed_kernel<hpf::codec_p59\|combine" "$f" | awk -F'","' '{printf "  %-75s calls %s avg_ns %s\n", substr($1,2,75), $2, $4}'
done
