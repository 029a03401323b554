#!/bin/bash
# dev helper, ON the GPU box: rocprofv3 kernel averages of devtools/select_modes.py under the given environment.  bash devtools/prof_select.sh "ENV=1 ..." 
cd "$(dirname "$0")/.." || exit 1
R=$PWD; export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/ps; (cd /tmp && env $v rocprofv3 --kernel-trace --stats -d /tmp/ps -- python $R/devtools/select_modes.py > /tmp/ps.out 2>&1)
  echo "== $v: $(grep 'align ms' /tmp/ps.out | cut -c1-110)"
  python profiles/summarize_rocpd.py $(find /tmp/ps -name "*results.db" | head -1) | grep -E "normal_eq|k_hist|knn_cone<4, false>" | cut -c1-150
done
