#!/bin/bash
# dev helper, run ON the GPU box: quick parity subset + per-iteration timing + cone phases.  bash devtools/r6_check.sh <tag> [pytest -k expr]
tag=${1:-r6}; kexpr=${2:-"direction or switch or trace or deterministic or reference_filter or cap_does or golden"}
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$kexpr" > gpurun_out/${tag}_subset.log 2>&1; echo "subset rc=$?"; tail -15 gpurun_out/${tag}_subset.log
timeout 300 python devtools/iter_profile.py > gpurun_out/${tag}_iter.txt 2>&1; grep -v "^$" gpurun_out/${tag}_iter.txt | tail -36
timeout 300 python devtools/cone_phases.py > gpurun_out/${tag}_phases.txt 2>&1; tail -34 gpurun_out/${tag}_phases.txt
