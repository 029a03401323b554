#!/bin/bash
# dev helper (on the GPU box): focused parity tests, then per-iteration kNN times of the benchmark align under the given
# environment variants:   cone_ab.sh <tag> "VAR=1 VAR2=2" "LSGPU_SO=devtools/x.so" ...   ("" = defaults)
tag=${1:-ab}; shift
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ -z "$NO_TESTS" ]; then
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "align or direction_index or radius_cap or submap_vs_scan or iteration_cap or golden or compute_matches or knn_exact" > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$?"; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/${tag}_tests.log | tail -15
fi
i=0
for v in "$@"; do
  i=$((i+1))
  vv=$(echo "$v" | sed "s#LSGPU_SO=#LSGPU_SO=$PWD/#")
  env $vv timeout 300 python devtools/iter_profile.py 16384 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" > gpurun_out/${tag}_iter_$i.txt
  echo "=== $i [${v:-defaults}]: $(grep set_ref gpurun_out/${tag}_iter_$i.txt)"
  grep knn_main gpurun_out/${tag}_iter_$i.txt | sed 's/.*knn_main \([0-9.]*\) us fb \([0-9.]*\) us.*/\1+\2/' | tr '\n' ' '; echo
done
