#!/bin/bash
# dev helper (GPU box): track_iter.py for the given scans under the given environment variants
#   track_iter.sh "17 19 8" "" "LSGPU_NO_CONE=1" ...
cd "$(dirname "$0")/.." || exit 1
scans=$1; shift
for s in $scans; do for v in "$@"; do
  echo "=== scan $s [${v:-defaults}]"
  env $v timeout 300 python devtools/track_iter.py $s 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"
done; done
