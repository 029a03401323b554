// slam_driver.cpp -- the per-scan loop of LaserSlamWorker::scanCallback
// (laser_slam_ros/src/laser_slam_worker.cpp:133-176) on the C++ mirror: LaserTrack::processPoseAndLaserScan,
// then IncrementalEstimator::registerPrior / estimate, then updateFromGTSAMValues; optionally one loop closure
// with the ICP step (incremental_estimator.cpp:89-115) at the end.
//   usage: slam_driver <dir> <n_scans> <icp_yaml> <nscan_in_sub_map> [<lc_index_a> <lc_index_b> <radius>]
// <dir>/scan<i>.bin = float32 N x 4, <dir>/poses.txt = "t_ns qw qx qy qz px py pz" per scan (odometry).
// Prints "pose <i> qw qx qy qz px py pz" for the final trajectory (before_lc / after_lc blocks).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>

#include "laser_slam_amd/incremental_estimator.hpp"

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

static void dump(const char* tag, LaserTrack& track, const std::vector<Time>& times) {
  std::printf("%s\n", tag);
  for (size_t i = 0; i < times.size(); ++i) {
    const SE3 T = track.evaluate(times[i]);
    std::printf("pose %zu %.12f %.12f %.12f %.12f %.9f %.9f %.9f\n", i, T.quaternion()[0], T.quaternion()[1],
                T.quaternion()[2], T.quaternion()[3], T.position()[0], T.position()[1], T.position()[2]);
  }
}

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  const std::string dir = argv[1];
  const int n = std::atoi(argv[2]);
  EstimatorParams ep;
  LaserTrackParams& p = ep.laser_track_params;
  p.icp_configuration_file = argv[3];
  {  // the input filter chain file lives next to the ICP chain file (an empty chain unless LSGPU_TEST_INPUT_FILTERS names one)
    const char* e = std::getenv("LSGPU_TEST_INPUT_FILTERS");
    const std::string yaml = argv[3];
    const size_t slash = yaml.find_last_of('/');
    p.icp_input_filters_file = e ? std::string(e) : (slash == std::string::npos ? std::string(".") : yaml.substr(0, slash)) + "/input_filters_none.yaml";
  }
  p.nscan_in_sub_map = std::atoi(argv[4]);
  p.odometry_noise_model = {0.5, 0.5, 0.5, 0.1, 0.1, 0.1};           // laser_slam_ros config_example.yaml scale
  p.icp_noise_model = {0.05, 0.05, 0.05, 0.015, 0.015, 0.015};
  ep.loop_closure_noise_model = {0.01, 0.01, 0.01, 0.003, 0.003, 0.003};
  ep.add_m_estimator_on_loop_closures = true;
  ep.do_icp_step_on_loop_closures = true;
  ep.loop_closures_sub_maps_radius = argc > 7 ? std::atoi(argv[7]) : 1;
  try {
    IncrementalEstimator est(ep, 1u);
    auto track = est.getLaserTrack(0);
    std::ifstream poses(dir + "/poses.txt");
    std::vector<Time> times;
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
      track->processPoseAndLaserScan(pose, scan, &factors, &values, &is_prior);
      const Values result = is_prior ? est.registerPrior(factors, values, 0u) : est.estimate(factors, values, tns);
      track->updateFromValues(result);
      times.push_back(tns);
    }
    std::printf("graph factors %zu error %.6g\n", est.graph().numFactors(), est.graph().error());
    dump("before_lc", *track, times);
    if (argc > 6) {
      RelativePose lc;  // a place recogniser reports: the clouds at a and b already overlap in the world frame
      lc.track_id_a = lc.track_id_b = 0;
      lc.time_a_ns = times.at((size_t)std::atoi(argv[5]));
      lc.time_b_ns = times.at((size_t)std::atoi(argv[6]));
      est.processLoopClosure(lc);
      const auto& st = est.lastLoopClosureIcpStats();
      std::printf("loop_closure iterations %d converged %d factors %zu error %.6g\n", st.iterations, st.converged,
                  est.graph().numFactors(), est.graph().error());
      const SE3& m = est.lastLoopClosure().T_a_b;
      std::printf("lc_measurement %.12f %.12f %.12f %.12f %.9f %.9f %.9f\n", m.quaternion()[0], m.quaternion()[1],
                  m.quaternion()[2], m.quaternion()[3], m.position()[0], m.position()[1], m.position()[2]);
      dump("after_lc", *track, times);
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "slam_driver: %s\n", e.what());
    return 1;
  }
  return 0;
}
