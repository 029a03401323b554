#!/bin/bash
# dev helper: rebuild the product .so and the -DLSGPU_KNN_STATS variant (absolute paths; safe from any cwd)
set -e
R=/root/repo
make -C $R/laser_slam_amd/csrc 2>&1 | grep -E "error|warning|Error" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DLSGPU_KNN_STATS -DLSGPU_EXPERIMENTS -shared \
  -o $R/devtools/liblsgpu_stats.so $R/laser_slam_amd/csrc/lsgpu_icp.hip $R/laser_slam_amd/csrc/lsgpu_host_filters.cpp 2>&1 | grep -E "error" || true
# the measured-slower variants (k_knn_lane / k_knn_classify / k_knn_rows / sparse lanes / 4-wave tiles / XCD swizzle)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DLSGPU_EXPERIMENTS -shared \
  -o $R/devtools/liblsgpu_exp.so $R/laser_slam_amd/csrc/lsgpu_icp.hip $R/laser_slam_amd/csrc/lsgpu_host_filters.cpp 2>&1 | grep -E "error" || true
ls -la $R/devtools/liblsgpu_exp.so $R/laser_slam_amd/liblsgpu_icp.so $R/devtools/liblsgpu_stats.so
