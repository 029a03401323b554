#!/bin/bash
# dev helper (GPU box): devtools/batch_bench.py (200 k-point pairs, 1 and 4 handles) under several environments.
# usage: bash devtools/batch_sweep.sh <tag> "ENV1=.. ENV2=.." "..." ...   ("-" = default environment)
tag=$1; shift
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
: > gpurun_out/${tag}_batch_sweep.txt
for v in "$@"; do
  [ "$v" = "-" ] && v=""
  echo "== $v" >> gpurun_out/${tag}_batch_sweep.txt
  env LSGPU_BATCH_POOLS=1,4 $v timeout 300 python devtools/batch_bench.py 3125 32 2>&1 | grep -E "^pool|unknown|rror" >> gpurun_out/${tag}_batch_sweep.txt
done
cat gpurun_out/${tag}_batch_sweep.txt
