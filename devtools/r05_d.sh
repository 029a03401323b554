#!/bin/bash
# round 5, GPU run D: the presorted upper levels (lsgpu_ssn_levels.hip.h): bit-exact filter tests, filter times per variant, profile
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-r05d}
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "reference_filter or filters_reproduce_golden or compute_matches_oracle_full or independent_known or compute_clouds" > gpurun_out/${tag}_tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/${tag}_tests.log
for v in default sortlevels root4096 root2048; do
  case $v in default) e="";; sortlevels) e="LSGPU_SSN_SORT_LEVELS=1";; root4096) e="LSGPU_SSN_ROOT=4096";; root2048) e="LSGPU_SSN_ROOT=2048";; esac
  env $e timeout 300 python devtools/filter_time.py > gpurun_out/${tag}_filter_$v.txt 2>&1; echo "== $v rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" gpurun_out/${tag}_filter_$v.txt | tail -6
done
rm -rf gpurun_out/prof_${tag}
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_${tag} -- python $OLDPWD/devtools/filter_time.py > /dev/null 2> $OLDPWD/gpurun_out/${tag}_prof.err)
db=$(find gpurun_out/prof_${tag} -name "*results.db" | head -1)
[ -n "$db" ] && python profiles/summarize_rocpd.py $db > gpurun_out/${tag}_filter.stats.txt && head -45 gpurun_out/${tag}_filter.stats.txt
rm -rf gpurun_out/prof_${tag}
