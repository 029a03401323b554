// <laser_slam/common.hpp> as laser_slam_ros includes it (laser_slam/include/laser_slam/common.hpp:14-149, 263-269): forwards to the
// GTSAM-typed overlay over the MI355X device path (integration/gtsam/laser_slam_gtsam_overlay.hpp, -DLSGPU_WITH_GTSAM=ON).
#pragma once
#include "../laser_slam_gtsam_overlay.hpp"
