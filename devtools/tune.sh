cd /root/repo
run() {
  echo -n "$* : "
  env "$@" timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-compute-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'knn', round(r['avg_main_us'],1), round(r['avg_fallback_us'],1))"
}
run A=1
run LSGPU_GAP=0.001
run LSGPU_GAP=0.003
run LSGPU_GAP=0.005
run LSGPU_BUDGET=64
run LSGPU_BUDGET=256
run LSGPU_WIDE_ITERS=2
run LSGPU_WIDE_ITERS=4
run LSGPU_ROUTE_R=0.01
run LSGPU_ROUTE_R=0.05
run LSGPU_Q_ELEV=0.3
run LSGPU_TILE_OCC_DUMMY=1
run A=2
