cd /root/repo
run() {
  echo "REFRESH=$1 GAP_MUL=$2 GAP_MAX=$3 GAP=$4"
  LSGPU_REFRESH=$1 LSGPU_GAP_MUL=$2 LSGPU_GAP_MAX=$3 LSGPU_GAP=$4 timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-compute-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('   ', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'knn', round(r['avg_main_us'],1), round(r['avg_fallback_us'],1), 'strag', round(r['stragglers_per_launch']), d['final_error_vs_truth']['trans_m'])"
}
run 0 0 0.03 0.002
run 1 0 0.03 0.002
run 2 0 0.03 0.002
run 0 4 0.03 0.002
run 1 4 0.03 0.002
run 2 4 0.03 0.002
run 1 2 0.03 0.002
run 1 6 0.03 0.002
run 1 4 0.01 0.002
run 1 4 0.06 0.002
run 1 0 0.03 0.004
run 1 0 0.03 0.008
