// track_driver.cpp -- drives laser_slam_amd::LaserTrack like LaserSlamWorker::scanCallback does
// (laser_slam_ros/src/laser_slam_worker.cpp:133): one processPoseAndLaserScan per scan.
//   usage: track_driver <dir> <n_scans> <icp_yaml> <nscan_in_sub_map> [<scans_on_device> [<shadow_scan> <oracle_threads>]]
// (built with -DLSGPU_TRACK_SHADOW -DLSGPU_TEST_SEAMS and linked with oracle/liblsoracle.so -- bench.py's value_track
// section: at scan <shadow_scan> the CPU oracle aligns the very clouds the device just aligned, from the same guess, and a
// "shadow" line reports its wall time and the difference of the two transforms)
// <dir>/scan<i>.bin = float32 N x 4 (x,y,z,1), <dir>/poses.txt = one "t_ns qw qx qy qz px py pz" per scan
// (odometry pose measurements).  Prints one line per produced factor / ICP result.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>

#include "laser_slam_amd/laser_track.hpp"
#ifdef LSGPU_TRACK_SHADOW
#include <chrono>
#include <cmath>
#include "../../oracle/icp_oracle.h"
#endif

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
  if (std::getenv("LSGPU_TRACK_NO_OVERLAP")) p.overlap_scan_copy = false;   // the scan's host copy in front of the registration, as upstream
  p.odometry_noise_model = {0.05, 0.05, 0.05, 0.01, 0.01, 0.01};
  p.icp_noise_model = {0.005, 0.005, 0.005, 0.0015, 0.0015, 0.0015};
  std::srand(4);
  try {
    LaserTrack track(p, 0u);
#ifdef LSGPU_TRACK_SHADOW
    const int shadow_scan = argc > 6 ? std::atoi(argv[6]) : -1;
    const int oracle_threads = argc > 7 ? std::atoi(argv[7]) : 1;
    int current_scan = -1;
    track.icp().setSeed(7);   // every compute() reseeds the filters' draw stream: device and oracle consume the same draws
    if (shadow_scan >= 0) track.icp().setComputeObserver([&](const ICP& self, const DataPoints& reading, const DataPoints& reference,
                                       const TransformationParameters& T_init, const TransformationParameters& T_dev) {
      if (current_scan != shadow_scan) return;
      lso_config c;
      lso_config_default(&c);
      c.reading_sampling_prob = self.readingSamplingProb(); c.surface_normal_knn = self.surfaceNormalKnn();
      c.surface_normal_ratio = self.surfaceNormalRatio(); c.trim_ratio = self.config().trim_ratio;
      c.max_iterations = self.config().max_iterations; c.min_diff_rot = self.config().min_diff_rot;
      c.min_diff_trans = self.config().min_diff_trans; c.smooth_length = self.config().smooth_length;
      c.accum_double = 1; c.num_threads = oracle_threads;
      TransformationParameters T = T_init;
      lso_stats st;
      const auto t0 = std::chrono::steady_clock::now();
      const int rc = lso_icp_compute_full(&c, reading.features.data(), reading.getNbPoints(), reference.features.data(),
                                          reference.getNbPoints(), T_init.data(), self.seed(), T.data(), &st);
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      // |translation difference| and the rotation angle of D = R_oracle^T R_device, from D's skew part (acos of the trace
      // loses half the digits near zero)
      double dt = 0, D[3][3];
      for (int r = 0; r < 3; ++r) dt += (double)(T[12 + r] - T_dev[12 + r]) * (double)(T[12 + r] - T_dev[12 + r]);
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
          D[a][b] = 0;
          for (int k = 0; k < 3; ++k) D[a][b] += (double)T[4 * a + k] * (double)T_dev[4 * b + k];   // column a of Ro . column b of Rd
        }
      const double sx = 0.5 * (D[2][1] - D[1][2]), sy = 0.5 * (D[0][2] - D[2][0]), sz = 0.5 * (D[1][0] - D[0][1]);
      const double dr = std::atan2(std::sqrt(sx * sx + sy * sy + sz * sz), 0.5 * (D[0][0] + D[1][1] + D[2][2] - 1.0));
      std::printf("shadow scan %d rc %d oracle_ms %.3f oracle_iterations %d device_iterations %d dt %.3e dr %.3e reading %lld reference %lld threads %d\n",
                  current_scan, rc, ms, st.iterations, self.lastStats().iterations, std::sqrt(dt), dr,
                  (long long)reading.getNbPoints(), (long long)reference.getNbPoints(), oracle_threads);
    });
#endif
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
#ifdef LSGPU_TRACK_SHADOW
      current_scan = i;
#endif
      track.processPoseAndLaserScan(pose, scan, &factors, &values, &is_prior);
      std::printf("scan %d prior %d factors %zu values %zu numscans %zu\n", i, (int)is_prior, factors.size(),
                  values.size(), track.getNumScans());
      for (const Factor& f : factors) {
        const auto& qq = f.measurement.quaternion();
        const auto& pp = f.measurement.position();
        std::printf("factor %d keys %zu %zu q %.9f %.9f %.9f %.9f p %.9f %.9f %.9f\n", (int)f.type, f.key_a,
                    f.key_b, qq[0], qq[1], qq[2], qq[3], pp[0], pp[1], pp[2]);
      }
      if (i > 0 && std::getenv("LSGPU_TRACK_STAGES"))
        std::printf("stages copy_ms %.3f upload_ms %.3f icp_ms %.3f device_filters_ms %.3f device_total_ms %.3f cone_launches %d cone_occupancy %.2f\n", track.lastStageTimes().copy_ms,
                    track.lastStageTimes().upload_ms, track.lastStageTimes().icp_ms, track.lastIcpStats().t_reserved[0], track.lastIcpStats().t_total_ms,
                    (int)track.lastIcpStats().direction_index_launches, (double)track.lastIcpStats().direction_index_occupancy);
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
