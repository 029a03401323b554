cd /root/repo
for b in 128 192 256 384 512 1024; do
  echo "LSGPU_NE_BLOCKS=$b"
  LSGPU_NE_BLOCKS=$b timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-compute-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('  ', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'ne', round(d['roofline_ne']['avg_us'],1), 'knn', round(r['avg_main_us'],1))"
done
