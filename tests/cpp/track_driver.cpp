// track_driver.cpp -- drives laser_slam_amd::LaserTrack like LaserSlamWorker::scanCallback does
// (laser_slam_ros/src/laser_slam_worker.cpp:133): one processPoseAndLaserScan per scan.
//   usage: track_driver <dir> <n_scans> <icp_yaml> <nscan_in_sub_map> [<scans_on_device>]
// <dir>/scan<i>.bin = float32 N x 4 (x,y,z,1), <dir>/poses.txt = one "t_ns qw qx qy qz px py pz" per scan
// (odometry pose measurements).  Prints one line per produced factor / ICP result.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>

#include "laser_slam_amd/laser_track.hpp"

using namespace laser_slam_amd;

static DataPoints readScan(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) { std::fprintf(stderr, "cannot open %s\n", path.c_str()); std::exit(2); }
  f.seekg(0, std::ios::end);
  const size_t bytes = (size_t)f.tellg();
  f.seekg(0);
  DataPoints d;
  d.features.resize(bytes / 4);
  f.read(reinterpret_cast<char*>(d.features.data()), (std::streamsize)bytes);
  return d;
}

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  const std::string dir = argv[1];
  const int n = std::atoi(argv[2]);
  LaserTrackParams p;
  p.icp_configuration_file = argv[3];
  {  // the input filter chain file lives next to the ICP chain file (an empty chain unless LSGPU_TEST_INPUT_FILTERS names one)
    const char* e = std::getenv("LSGPU_TEST_INPUT_FILTERS");
    const std::string yaml = argv[3];
    const size_t slash = yaml.find_last_of('/');
    p.icp_input_filters_file = e ? std::string(e) : (slash == std::string::npos ? std::string(".") : yaml.substr(0, slash)) + "/input_filters_none.yaml";
  }
  p.nscan_in_sub_map = std::atoi(argv[4]);
  if (argc > 5) p.scans_on_device = std::atoi(argv[5]);
  p.odometry_noise_model = {0.05, 0.05, 0.05, 0.01, 0.01, 0.01};
  p.icp_noise_model = {0.005, 0.005, 0.005, 0.0015, 0.0015, 0.0015};
  std::srand(4);
  try {
    LaserTrack track(p, 0u);
    std::ifstream poses(dir + "/poses.txt");
    for (int i = 0; i < n; ++i) {
      Pose pose;
      double q[4], t[3];
      long long tns;
      poses >> tns >> q[0] >> q[1] >> q[2] >> q[3] >> t[0] >> t[1] >> t[2];
      pose.time_ns = tns;
      pose.T_w = SE3({q[0], q[1], q[2], q[3]}, {t[0], t[1], t[2]});
      LaserScan scan;
      scan.time_ns = tns;
      scan.scan = readScan(dir + "/scan" + std::to_string(i) + ".bin");
      FactorList factors;
      Values values;
      bool is_prior = false;
      track.processPoseAndLaserScan(pose, scan, &factors, &values, &is_prior);
      std::printf("scan %d prior %d factors %zu values %zu numscans %zu\n", i, (int)is_prior, factors.size(),
                  values.size(), track.getNumScans());
      for (const Factor& f : factors) {
        const auto& qq = f.measurement.quaternion();
        const auto& pp = f.measurement.position();
        std::printf("factor %d keys %zu %zu q %.9f %.9f %.9f %.9f p %.9f %.9f %.9f\n", (int)f.type, f.key_a,
                    f.key_b, qq[0], qq[1], qq[2], qq[3], pp[0], pp[1], pp[2]);
      }
      if (i > 0) std::printf("icp_iterations %d converged %d scan_ms %.3f\n", track.lastIcpStats().iterations,
                             track.lastIcpStats().converged, track.getScanMatchingTimes().at(scan.time_ns));
    }
    DataPoints world, submap;
    track.getLocalCloudInWorldFrame(track.getMaxTime(), &world);
    track.buildSubMapAroundTime(track.getLaserScans()[1].time_ns, 1, &submap);
    std::printf("world_cloud %lld submap %lld\n", (long long)world.getNbPoints(), (long long)submap.getNbPoints());
  } catch (const std::exception& e) {
    std::printf("exception %s\n", e.what());
    return 1;
  }
  return 0;
}
