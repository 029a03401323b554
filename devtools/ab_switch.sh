#!/bin/bash
# dev helper (on the GPU box): A/B of library switches -- whole-compute / resident-loop scans/s and per-iteration kNN times
# usage: ab_switch.sh "" LSGPU_NO_LAZY=1 "LSGPU_NO_REP=1 LSGPU_NO_LAZY=1" ...     (one variant per argument; "" = defaults)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
i=0
for v in "$@"; do
  i=$((i+1))
  echo "=== variant $i: [${v:-defaults}]"
  env $v timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value %.1f loop %.1f knn_us %.1f main %.1f fb %.1f ne %.1f sel %.1f filt_ms %.2f' % (d['value'], d['value_loop']['value'], d['roofline']['avg_launch_us'], d['roofline']['avg_main_us'], d['roofline']['avg_fallback_us'], d['roofline_ne']['avg_us'], d['roofline_select']['avg_us'], d['filters_and_grid_ms_per_step']))"
  env $v timeout 300 python devtools/iter_profile.py 16384 2>/dev/null | grep knn_main | sed 's/.*knn_main \([0-9.]*\) us fb \([0-9.]*\).*/\1+\2/' | tr '\n' ' '
  echo
done
