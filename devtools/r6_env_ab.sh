#!/bin/bash
# dev helper, run ON the GPU box: per-iteration kNN time of the benchmark align under several environments (one build)
# usage: bash devtools/r6_env_ab.sh <tag> "ENV1=.. ENV2=.." "..." ...   ("-" = default environment)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
tag=$1; shift
k=0
for envs in "$@"; do
  k=$((k+1)); [ "$envs" = "-" ] && envs=""
  env $envs timeout 300 python devtools/iter_profile.py > gpurun_out/${tag}_$k.txt 2>&1
  echo "== [$envs]"; grep -E "align ms|knn_main" gpurun_out/${tag}_$k.txt | awk '/align ms/{print} /knn_main/{n++; s+=$7; f+=$10; if(n<=3) a+=$7; else if (n<=16) b+=$7; else c+=$7; if(n<=4||n%4==0) printf "%s ",$7} END{printf "\n  launches %d main %.0f fb %.0f | it0-2 %.0f it3-15 %.0f it16-31 %.0f\n",n,s,f,a,b,c}'
done
