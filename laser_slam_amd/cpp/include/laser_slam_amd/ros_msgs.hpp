// ros_msgs.hpp -- the ROS message surface of the scan path, dependency free.
//
// LaserSlamWorker::scanCallback receives a sensor_msgs/PointCloud2, converts it with
// PointMatcher_ros::rosMsgToPointMatcherCloud<float> (laser_slam_ros/src/laser_slam_worker.cpp:125) and hands the
// DataPoints to LaserTrack; clouds go back out through lpmToPcl + pcl::toROSMsg
// (laser_slam_ros/include/laser_slam_ros/common.hpp:159-191).  ROS is not available here, so the message is the
// plain struct below with the field names and meaning of sensor_msgs/PointCloud2 and PointField; the conversions run
// on the device (lsgpu_cloud_from_pointcloud2 / lsgpu_cloud_to_pointxyz): the message's byte block is what crosses PCIe.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "laser_slam_amd/icp.hpp"

namespace laser_slam_amd {

struct PointField {  // sensor_msgs/PointField
  enum : uint8_t { INT8 = 1, UINT8 = 2, INT16 = 3, UINT16 = 4, INT32 = 5, UINT32 = 6, FLOAT32 = 7, FLOAT64 = 8 };
  std::string name;
  uint32_t offset = 0;
  uint8_t datatype = FLOAT32;
  uint32_t count = 1;
};

struct PointCloud2 {  // sensor_msgs/PointCloud2 (header reduced to what the path reads)
  uint64_t stamp_ns = 0;
  std::string frame_id;
  uint32_t height = 1, width = 0;
  std::vector<PointField> fields;
  bool is_bigendian = false;
  uint32_t point_step = 0, row_step = 0;
  std::vector<uint8_t> data;
  bool is_dense = true;
};

// PointMatcher_ros::rosMsgToPointMatcherCloud<float>: the x, y, z FLOAT32 fields become features (x,y,z,1 per point);
// records with a non-finite coordinate are dropped when the message says it is not dense.  The other fields would
// become descriptors upstream; the ICP path reads none of them.
inline DataPoints rosMsgToPointMatcherCloud(ICP& icp, const PointCloud2& msg) {
  int off[3] = {-1, -1, -1};
  for (const PointField& f : msg.fields) {
    const int k = f.name == "x" ? 0 : f.name == "y" ? 1 : f.name == "z" ? 2 : -1;
    if (k < 0) continue;
    if (f.datatype != PointField::FLOAT32 || f.count != 1) throw std::runtime_error("PointCloud2: x/y/z must be FLOAT32");
    off[k] = (int)f.offset;
  }
  if (off[0] < 0 || off[1] < 0 || off[2] < 0) throw std::runtime_error("PointCloud2: no x/y/z fields");
  const int64_t n = (int64_t)msg.width * (int64_t)msg.height;
  if ((uint64_t)n * msg.point_step > msg.data.size()) throw std::runtime_error("PointCloud2: data shorter than width * height * point_step");
  DataPoints out;
  out.features.resize((size_t)std::max<int64_t>(n, 1) * 4);
  int64_t m = 0;
  const int rc = lsgpu_cloud_from_pointcloud2(icp.handle(), msg.data.data(), n, (int)msg.point_step, off[0], off[1], off[2],
                                              msg.is_bigendian ? 1 : 0, msg.is_dense ? 0 : 1, out.features.data(), &m);
  if (rc != LSGPU_OK) throw DeviceError(std::string("lsgpu_cloud_from_pointcloud2: ") + lsgpu_strerror(rc) + " [" + lsgpu_last_error(icp.handle()) + "]");
  out.features.resize((size_t)m * 4);
  return out;
}

// lpmToPcl + pcl::toROSMsg<pcl::PointXYZ>: fields x@0 y@4 z@8 FLOAT32, point_step 16, one row.
inline PointCloud2 pointMatcherCloudToRosMsg(ICP& icp, const DataPoints& cloud, const std::string& frame_id, uint64_t stamp_ns) {
  PointCloud2 msg;
  msg.stamp_ns = stamp_ns;
  msg.frame_id = frame_id;
  msg.height = 1;
  msg.width = (uint32_t)cloud.getNbPoints();
  const char* names[3] = {"x", "y", "z"};
  for (int k = 0; k < 3; ++k) { PointField f; f.name = names[k]; f.offset = 4u * (uint32_t)k; msg.fields.push_back(f); }
  msg.point_step = 16;
  msg.row_step = 16 * msg.width;
  msg.data.resize((size_t)msg.row_step);
  const int rc = lsgpu_cloud_to_pointxyz(icp.handle(), cloud.features.data(), cloud.getNbPoints(), msg.data.data());
  if (rc != LSGPU_OK) throw DeviceError(std::string("lsgpu_cloud_to_pointxyz: ") + lsgpu_strerror(rc));
  return msg;
}

}  // namespace laser_slam_amd
