cd /root/repo
LSGPU_SPARSE_LANES=16 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "not config3_full" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
for cfg in "0 512" "8 512" "16 2048" "24 2048" "32 2048" "48 2048"; do
  set -- $cfg
  echo "sparse=$1 rowq_blocks=$2"
  LSGPU_SPARSE_LANES=$1 LSGPU_ROWQ_BLOCKS=$2 timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-compute-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'main', round(r['avg_main_us'],1), 'fb', round(r['avg_fallback_us'],1), 'strag', r['stragglers_per_launch'], d['final_error_vs_truth'])"
done
