#!/bin/bash
# dev helper (GPU box): the bit-exact filter tests (REPS="1 2 3": repeated, to catch a race), then devtools/filter_time.py
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
ulimit -c 0
for rep in ${REPS:-1}; do
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "reference_filter or filters_reproduce_golden or compute_matches_oracle_full or independent_known or compute_clouds" 2>&1 | grep "^E  \|passed\|failed" | cut -c1-400 | tail -12
done
LSGPU_GS_DEBUG=1 timeout 120 python devtools/filter_time.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -12
