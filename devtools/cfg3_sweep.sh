#!/bin/bash
# dev helper (GPU box): devtools/config4_shape.py (configs[3] on one GPU) under several environments; one line per variant.
# usage: bash devtools/cfg3_sweep.sh <tag> "ENV1=.. ENV2=.." "..." ...   ("-" = default environment)
tag=$1; shift
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for v in "$@"; do
  [ "$v" = "-" ] && v=""
  echo "== $v" >> gpurun_out/${tag}_cfg3_sweep.txt
  env $v timeout 300 python devtools/config4_shape.py 16384 - 2>&1 | grep -E "set_reference ms|err vs truth|unknown|rror" >> gpurun_out/${tag}_cfg3_sweep.txt
done
cat gpurun_out/${tag}_cfg3_sweep.txt
