#!/bin/bash
# dev helper (GPU box): rocprofv3 timeline of the last compute of devtools/step_workload.py (devtools/timeline.py)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-tl}
rm -rf gpurun_out/prof_${tag}
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_${tag} -- python $OLDPWD/devtools/step_workload.py 16384 4 > $OLDPWD/gpurun_out/${tag}_workload.txt 2> $OLDPWD/gpurun_out/${tag}_prof.err)
cat gpurun_out/${tag}_workload.txt | grep compute
db=$(find gpurun_out/prof_${tag} -name "*results.db" | head -1)
python devtools/timeline.py $db 5 > gpurun_out/${tag}_timeline.txt; head -3 gpurun_out/${tag}_timeline.txt; tail -2 gpurun_out/${tag}_timeline.txt
rm -rf gpurun_out/prof_${tag}
