cd /root/repo
for b in 512 1024 2048 4096; do
  echo "rowq_blocks=$b"
  LSGPU_ROWQ_BLOCKS=$b timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-compute-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'knn', round(r['avg_main_us'],1), round(r['avg_fallback_us'],1), 'strag', r['stragglers_per_launch'])"
done
