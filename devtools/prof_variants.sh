#!/bin/bash
# dev helper (GPU box): rocprofv3 kernel averages of the benchmark compute for several builds of the library (LSGPU_SO)
# usage: bash devtools/prof_variants.sh <tag> lib1.so lib2.so ...
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
tag=$1; shift
for so in "$@"; do
  name=$(basename $so .so)
  rm -rf gpurun_out/pv_${tag}_$name
  (cd /tmp && LSGPU_SO=$OLDPWD/$so timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/pv_${tag}_$name -- python $OLDPWD/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-compute-e2e > $OLDPWD/gpurun_out/pv_${tag}_$name.json 2> /dev/null)
  db=$(find gpurun_out/pv_${tag}_$name -name "*results.db" | head -1)
  echo "== $name  value $(python -c "import json;print(round(json.load(open('gpurun_out/pv_${tag}_$name.json'))['value'],1))")"
  [ -n "$db" ] && python profiles/summarize_rocpd.py $db | grep -E "k_knn_cone|k_ref_gather|k_ssn_select|k_ssn_emit|k_cone_gather|k_normal_eq" | cut -c1-150
  rm -rf gpurun_out/pv_${tag}_$name
done
