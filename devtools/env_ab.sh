#!/bin/bash
# dev helper (on the GPU box): run one script under several environments:  env_ab.sh <script.py> "" "VAR=1" ...
script=$1; shift
cd "$(dirname "$0")/.." || exit 1
for v in "$@"; do
  echo "=== [${v:-defaults}]"
  env $v timeout 600 python $script 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"
done
