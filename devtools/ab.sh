cd /root/repo
for rep in 1 2; do
for lib in liblsgpu_prev.so liblsgpu_icp.so; do
  echo -n "$lib: "
  LSGPU_SO=/root/repo/laser_slam_amd/$lib timeout 200 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-compute-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'knn', round(r['avg_main_us'],1), round(r['avg_fallback_us'],1), 'ne', round(d['roofline_ne']['avg_us'],1))"
done; done
