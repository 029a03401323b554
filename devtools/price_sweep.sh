#!/bin/bash
# dev helper (GPU box): the direction index's price check (stats.direction_index_heavy_share) and the align time on six
# workloads -- the benchmark pair, two scan-to-sub-map registrations of the track drive (a wall 0.8 m from the sensor in
# scan 18) with the odometry guess and with a guess 0.3 m off, bench.py's F_submap3 / P_submap3 -- per environment variant:
#   price_sweep.sh "" "LSGPU_CONE_HEAVY_STEPS=64 LSGPU_CONE_HEAVY_SHARE=2" ...
cd "$(dirname "$0")/.." || exit 1
f() { grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|^chunks\|^knn us"; }
for v in "$@"; do
  echo "=== [${v:-defaults}]"
  echo "pair: $(env $v timeout 300 python devtools/iter_profile.py 16384 2>&1 | grep set_ref | sed 's/set_reference ms [0-9.]* //')"
  for s in 18 8; do for g in "GUESS=0.3" ""; do
    echo "scan $s [$g]: $(env $v $g timeout 300 python devtools/track_iter.py $s 2>&1 | f | sed 's/compute ms \([0-9.]*\) filters [0-9.]* //' | tr '\n' ' ')"
  done; done
  for c in F P; do
    echo "submap3 $c: $(env $v CHAIN=$c TIGHT=1 GUESS_E=1 timeout 300 python devtools/track_iter.py 3 2>&1 | f | sed 's/compute ms \([0-9.]*\) filters [0-9.]* //' | tr '\n' ' ')"
  done
done
