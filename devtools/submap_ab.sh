#!/bin/bash
# dev helper (GPU box): bench.py's compute_variants F_submap3 / P_submap3 (scan 3 against scans 2, 1, 0; guess off by
# 10 cm / 0.5 deg; 1e-4 m / 1e-5 rad) under the given environment variants, per-iteration kNN times
cd "$(dirname "$0")/.." || exit 1
f() { grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|^chunks"; }
for v in "$@"; do
  echo "=== [${v:-defaults}]"
  for c in F P; do
    echo "$c: $(env $v CHAIN=$c TIGHT=1 GUESS_E=1 timeout 300 python devtools/track_iter.py 3 2>&1 | f | tr '\n' ' ')"
  done
done
