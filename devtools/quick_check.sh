cd /root/repo
timeout 900 python -m pytest tests -m gpu -q -x -k "not config4_sequence and not config3_full" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'knn', round(r['avg_main_us'],1), round(r['avg_fallback_us'],1), 'ne', round(d['roofline_ne']['avg_us'],1), 'sel', round(d['roofline_select']['avg_us'],1), 'e2e', {k:round(v['ms_per_scan'],2) for k,v in d['value_e2e'].items() if isinstance(v,dict)}, d['final_error_vs_truth'])"
