#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-r05g}
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "compute_matches_oracle_full or compute_clouds or reading_filter or reference_filter_is or icp_class_compute" > gpurun_out/${tag}_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/${tag}_tests.log
bash devtools/r05_tl.sh ${tag}
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-compute-e2e > gpurun_out/${tag}_bench0.json 2> gpurun_out/${tag}_bench0.err; echo "bench0 rc=$?"; cut -c1-700 gpurun_out/${tag}_bench0.json
