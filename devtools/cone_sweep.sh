#!/bin/bash
# dev helper (on the GPU box): k_knn_cone -- SQ counters of the benchmark align, rows x cols sweep of the direction index
# usage: cone_sweep.sh <tag> [so]
tag=${1:-sweep}; so=${2:-}
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
[ -n "$so" ] && export LSGPU_SO=$PWD/$so
rm -rf gpurun_out/sq_${tag}
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM -d $OLDPWD/gpurun_out/sq_${tag} --output-format csv -- python $OLDPWD/devtools/kernel_times.py > /dev/null 2> $OLDPWD/gpurun_out/${tag}_sq.err)
python devtools/pmc_summary.py $(find gpurun_out/sq_${tag} -name "*counter_collection.csv" | head -1) k_knn_ > gpurun_out/${tag}_knn_sq_counters.txt 2>> gpurun_out/${tag}_sq.err; cat gpurun_out/${tag}_knn_sq_counters.txt
rm -rf gpurun_out/sq_${tag}
run() {   # label, env...
  label=$1; shift
  env "$@" timeout 300 python devtools/iter_profile.py 16384 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" > gpurun_out/${tag}_iter_${label}.txt
  echo "=== $label: $(head -2 gpurun_out/${tag}_iter_${label}.txt | grep set_ref)"
  grep knn_main gpurun_out/${tag}_iter_${label}.txt | sed 's/.*knn_main \([0-9.]*\) us fb.*/\1/' | tr '\n' ' '; echo
}
for rc in "64 8192" "96 8192" "128 8192" "192 8192" "256 8192" "128 4096" "128 16384" "256 16384"; do
  set -- $rc
  run r$1_c$2 LSGPU_CONE_ROWS=$1 LSGPU_CONE_COLS=$2
done
