#!/bin/bash
# dev helper, run ON the GPU box: per-iteration kNN time of the benchmark align for several builds of the library
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
tag=$1; shift
for so in "$@"; do
  name=$(basename $so .so)
  LSGPU_SO=$PWD/$so timeout 300 python devtools/iter_profile.py > gpurun_out/${tag}_$name.txt 2>&1
  echo "== $name"; grep -E "align ms|knn_main" gpurun_out/${tag}_$name.txt | awk '/align ms/{print} /knn_main/{n++; s+=$7; if(n<=4||n%4==0) printf "%s ",$7} END{printf "\n  launches %d sum %.1f us avg %.1f us\n",n,s,s/n}'
done
