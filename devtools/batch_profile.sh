#!/bin/bash
# dev helper (GPU box): rocprofv3 kernel trace of a few 200 k-point pairs through ONE handle (devtools/batch_bench.py):
# per-kernel statistics and the timeline of the last burst -- what a small pair pays per launch.   batch_profile.sh [tag] [pairs]
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-bp}; pairs=${2:-4}
rm -rf gpurun_out/prof_${tag}
(cd /tmp && LSGPU_BATCH_POOLS=1 timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_${tag} -- python $OLDPWD/devtools/batch_bench.py 3125 $pairs > $OLDPWD/gpurun_out/${tag}_workload.txt 2> $OLDPWD/gpurun_out/${tag}_prof.err)
grep pool gpurun_out/${tag}_workload.txt
db=$(find gpurun_out/prof_${tag} -name "*results.db" | head -1)
python profiles/summarize_rocpd.py $db > gpurun_out/${tag}_stats.txt
python devtools/timeline.py $db 5 > gpurun_out/${tag}_timeline.txt; head -3 gpurun_out/${tag}_timeline.txt; tail -2 gpurun_out/${tag}_timeline.txt
rm -rf gpurun_out/prof_${tag}
