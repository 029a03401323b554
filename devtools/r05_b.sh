#!/bin/bash
# round 5, GPU run B: k_ssn_tree iterations -- bit-exact filter tests, phase stamps of workgroup 0, filter times
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-r05b}
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "reference_filter or filters_reproduce_golden or compute_matches_oracle_full or independent_known" > gpurun_out/${tag}_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/${tag}_tests.log
timeout 300 python devtools/filter_time.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tee gpurun_out/${tag}_filter_default.txt
timeout 300 python devtools/tree_phases.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tee gpurun_out/${tag}_tree_phases.txt
