// <laser_slam/parameters.hpp> as laser_slam_ros includes it (laser_slam/include/laser_slam/parameters.hpp:8-34): forwards to the
// GTSAM-typed overlay over the MI355X device path (integration/gtsam/laser_slam_gtsam_overlay.hpp, -DLSGPU_WITH_GTSAM=ON).
#pragma once
#include "../laser_slam_gtsam_overlay.hpp"
