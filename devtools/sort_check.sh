cd /root/repo
for v in 16 8 4; do
  echo "LSGPU_SORT_ITEMS=$v"
  LSGPU_SORT_ITEMS=$v timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'e2e', {k:round(v['ms_per_scan'],2) for k,v in d['value_e2e'].items() if isinstance(v,dict)}, d['final_error_vs_truth'])"
done
