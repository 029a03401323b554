#!/bin/bash
# dev helper (on the GPU box): first look at k_knn_cone -- focused parity tests, then per-iteration kNN times of the
# benchmark align for the product build, LSGPU_NO_CONE and the build variants named on the command line
# usage: cone_check.sh <tag> [variant.so ...]
tag=${1:-cone}; shift
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "align or direction_index or radius_cap or submap_vs_scan or iteration_cap or golden or compute_matches or knn_exact" > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$?"; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/${tag}_tests.log | tail -25
run() {   # label, env...
  label=$1; shift
  echo "=== $label"
  env "$@" timeout 300 python devtools/iter_profile.py 16384 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" > gpurun_out/${tag}_iter_${label}.txt
  head -1 gpurun_out/${tag}_iter_${label}.txt
  grep knn_main gpurun_out/${tag}_iter_${label}.txt | sed 's/.*knn_main \([0-9.]*\) us fb \([0-9.]*\) us strag \([0-9]*\).*/\1+\2(\3)/' | tr '\n' ' '; echo
}
run product LSGPU_GAP=0.002
run nocone LSGPU_NO_CONE=1
for so in "$@"; do run $(basename $so .so) LSGPU_SO=$PWD/$so; done
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-compute-e2e > gpurun_out/${tag}_bench0.json 2> gpurun_out/${tag}_bench0.err; echo "bench0 rc=$?"; cut -c1-1200 gpurun_out/${tag}_bench0.json
