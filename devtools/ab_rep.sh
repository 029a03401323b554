#!/bin/bash
# dev helper (on the GPU box): A/B of the replicated evaluation -- resident-loop scans/s and per-iteration kNN times
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for v in rep norep; do
  if [ $v = norep ]; then export LSGPU_NO_REP=1; else unset LSGPU_NO_REP; fi
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', 'value %.1f loop %.1f knn_us %.1f main %.1f ne %.1f sel %.1f' % (d['value'], d['value_loop']['value'], d['roofline']['avg_launch_us'], d['roofline']['avg_main_us'], d['roofline_ne']['avg_us'], d['roofline_select']['avg_us']))"
  timeout 300 python devtools/iter_profile.py 16384 2>/dev/null | grep knn_main | sed 's/.*knn_main \([0-9.]*\) us fb \([0-9.]*\).*/\1+\2/' | tr '\n' ' ' > gpurun_out/ab_${v}_iters.txt
  echo "$v per-iteration knn_main us:"; cat gpurun_out/ab_${v}_iters.txt
done
