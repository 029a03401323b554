// laser_slam_gtsam_overlay.hpp -- seam B1 with the REAL types: namespace laser_slam as laser_slam_ros
// compiles against it (laser_slam_ros/src/laser_slam_worker.cpp:47-600), implemented ON TOP of the dependency-free,
// tested mirror in laser_slam_amd/cpp:
//   LaserTrack            every public member of laser_slam/include/laser_slam/laser_track.hpp:20-144
//   IncrementalEstimator  every public member of laser_slam/include/laser_slam/incremental_estimator.hpp:20-53,
//                         over the real gtsam::ISAM2 (incremental_estimator.cpp:12-61, 63-149, 151-163, 165-266, 268-291)
//   common.hpp            Pose / RelativePose / LaserScan / Trajectory / Covariance / Clock / correctTransformationMatrix /
//                         convertTransformationMatrixToSE3 (common.hpp:14-149, 263-269)
// The mirror does the bookkeeping and the device ICP; this overlay turns its plain-data factor records into the
// gtsam::ExpressionFactor<SE3> objects the reference emits (laser_track.cpp:431-458), its SE3 into
// kindr::minimal::QuatTransformation, its clouds into PointMatcher<float>::DataPoints, and keeps the gtsam::ISAM2 graph.
// The files integration/gtsam/laser_slam/*.hpp forward the reference's include names to this header.
//
// Built only with -DLSGPU_WITH_GTSAM=ON (integration/CMakeLists.txt).  GTSAM, minkindr, minkindr_gtsam and
// libpointmatcher are NOT installed in the image this repository is developed in: here the header is only PARSED
// (g++ -fsyntax-only against the declaration-only stand-ins of tests/cpp/mock/, tests/test_cpp_mirror.py::
// test_gtsam_overlay_parses_and_resolves_the_ros_worker_calls, together with a translation unit that makes every call
// laser_slam_worker.cpp makes).  That check proves the names and signatures resolve; it pins NO behaviour -- everything
// beneath the overlay (laser_slam_amd::LaserTrack / IncrementalEstimator / WorkerLinks / ICP) is what is tested.
// Out of scope: laser_slam/benchmarker.hpp.
#pragma once
#include <gtsam/nonlinear/Expression.h>
#include <gtsam/nonlinear/ExpressionFactor.h>
#include <gtsam/nonlinear/ISAM2.h>
#include <gtsam/nonlinear/Marginals.h>
#include <gtsam/nonlinear/NonlinearFactorGraph.h>
#include <gtsam/nonlinear/Values.h>
#include <kindr/minimal/quat-transformation.h>
#include <kindr/minimal/quat-transformation-gtsam.h>
#include <pointmatcher/PointMatcher.h>
#include <sys/time.h>

#include <algorithm>
#include <array>
#include <cstdint>
#include <fstream>
#include <stdexcept>
#include <ctime>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "laser_slam_amd/incremental_estimator.hpp"

namespace laser_slam {   // the reference's own namespace: laser_slam_ros sees the types it expects

typedef PointMatcher<float> PointMatcher;                                   // common.hpp:14
typedef typename PointMatcher::DataPoints DataPoints;                       // common.hpp:15
typedef kindr::minimal::QuatTransformationTemplate<double> SE3;             // common.hpp:17
typedef SE3::Rotation SO3;                                                  // common.hpp:18
typedef int64_t Time;                                                       // curves::Time (common.hpp:20)
typedef size_t Key;                                                         // common.hpp:85
struct Pose { SE3 T_w; Time time_ns; Key key; };                            // common.hpp:87-94
struct RelativePose { SE3 T_a_b; Time time_a_ns, time_b_ns; Key key_a, key_b; unsigned int track_id_a, track_id_b; };
struct LaserScan { DataPoints scan; Time time_ns; Key key; };               // common.hpp:113-120
typedef Eigen::MatrixXd Covariance;                                         // common.hpp:122
typedef std::map<Time, SE3> Trajectory;                                     // common.hpp:133
// parameters.hpp:8-34: same member names; the noise models are std::array<double, 6> (operator[] like the Eigen vector
// laser_slam_ros fills element by element, laser_slam_ros/common.hpp:103-135); two extra members with defaults
// (device, scans_on_device)
using LaserTrackParams = laser_slam_amd::LaserTrackParams;
using EstimatorParams = laser_slam_amd::EstimatorParams;

class Clock {                                                               // common.hpp:23-63
 public:
  Clock() { start(); }
  void start() { gettimeofday(&real_time_start_, NULL); cpu_start_ = clock(); }
  void takeTime() {
    struct timeval end;
    gettimeofday(&end, NULL);
    cpu_time_ms_ = double(clock() - cpu_start_) / CLOCKS_PER_SEC * 1000.0;
    real_time_ms_ = ((end.tv_sec - real_time_start_.tv_sec) * 1000.0 + (end.tv_usec - real_time_start_.tv_usec) * 0.001) + 0.5;
  }
  double getRealTime() { return real_time_ms_; }
  double getCPUTime() { return cpu_time_ms_; }
  double takeRealTime() { takeTime(); return getRealTime(); }

 private:
  struct timeval real_time_start_;
  double real_time_ms_ = 0.0, cpu_time_ms_ = 0.0;
  clock_t cpu_start_ = 0;
};

namespace overlay_detail {
namespace m = laser_slam_amd;
inline m::SE3 toMirror(const SE3& T) {
  const auto& q = T.getRotation();
  const auto& p = T.getPosition();
  return m::SE3({q.w(), q.x(), q.y(), q.z()}, {p[0], p[1], p[2]});
}
inline SE3 fromMirror(const m::SE3& T) {
  return SE3(SE3::Rotation(T.quaternion()[0], T.quaternion()[1], T.quaternion()[2], T.quaternion()[3]),
             SE3::Position(T.position()[0], T.position()[1], T.position()[2]));
}
// features are (dim+1) x N column major on both sides: a plain copy.  Descriptors: the scans laser_slam stores carry none
// (rosMsgToPointMatcherCloud of an x,y,z cloud; the normals of the ICP chain live inside icp_.compute) -- a cloud that
// does carry some is refused instead of silently thinned.
inline m::DataPoints toMirror(const DataPoints& c) {
  if (c.descriptors.cols() != 0 && c.descriptors.rows() != 0)
    throw std::runtime_error("laser_slam overlay: DataPoints with descriptors are not carried through the device path");
  m::DataPoints d;
  d.features.assign(c.features.data(), c.features.data() + c.features.size());
  return d;
}
inline DataPoints fromMirror(const m::DataPoints& d) {
  DataPoints::Labels labels;
  labels.push_back(DataPoints::Label("x", 1)); labels.push_back(DataPoints::Label("y", 1));
  labels.push_back(DataPoints::Label("z", 1)); labels.push_back(DataPoints::Label("pad", 1));
  DataPoints c(labels, DataPoints::Labels(), (size_t)d.getNbPoints());
  std::copy(d.features.begin(), d.features.end(), c.features.data());
  return c;
}
inline gtsam::noiseModel::Base::shared_ptr noiseOf(const std::array<double, 6>& sigmas, bool cauchy) {   // laser_track.cpp:37-64
  gtsam::Vector6 s;
  for (int i = 0; i < 6; ++i) s[i] = sigmas[(size_t)i];
  gtsam::noiseModel::Base::shared_ptr n = gtsam::noiseModel::Diagonal::Sigmas(s);
  if (cauchy) n = gtsam::noiseModel::Robust::Create(gtsam::noiseModel::mEstimator::Cauchy::Create(1), n);
  return n;
}
// one gtsam::ExpressionFactor<SE3> per mirror factor record (makeMeasurementFactor / makeRelativeMeasurementFactor,
// laser_track.cpp:421-458; the loop-closure factor of incremental_estimator.cpp:117-127)
inline gtsam::ExpressionFactor<SE3> factorOf(const m::Factor& f) {
  using gtsam::Expression;
  if (f.type == m::Factor::PRIOR)
    return gtsam::ExpressionFactor<SE3>(noiseOf(f.sigmas, f.cauchy), fromMirror(f.measurement), Expression<SE3>(f.key_b));
  const Expression<SE3> T_w_b(f.key_b);
  const Expression<SE3> T_w_a = f.fix_first_node ? Expression<SE3>(fromMirror(f.fixed_a)) : Expression<SE3>(f.key_a);
  return gtsam::ExpressionFactor<SE3>(noiseOf(f.sigmas, f.cauchy), fromMirror(f.measurement),
                                      kindr::minimal::compose(kindr::minimal::inverse(T_w_a), T_w_b));
}
inline m::Values toMirror(const gtsam::Values& v) {
  m::Values out;
  for (const auto& kv : v) out[(m::Key)kv.key] = toMirror(kv.value.template cast<SE3>());
  return out;
}
inline m::Pose toMirror(const Pose& p) { m::Pose o; o.T_w = toMirror(p.T_w); o.time_ns = p.time_ns; o.key = p.key; return o; }
inline Pose fromMirror(const m::Pose& p) { return Pose{fromMirror(p.T_w), p.time_ns, p.key}; }
inline m::RelativePose toMirror(const RelativePose& r) {
  m::RelativePose o;
  o.T_a_b = toMirror(r.T_a_b); o.time_a_ns = r.time_a_ns; o.time_b_ns = r.time_b_ns; o.key_a = r.key_a; o.key_b = r.key_b;
  o.track_id_a = r.track_id_a; o.track_id_b = r.track_id_b;
  return o;
}
}  // namespace overlay_detail

// common.hpp:136-149 / 263-269, on PointMatcher's own matrix type (the arithmetic is the mirror's, i.e. lsgpu_check_rigid /
// lsgpu_correct_rigid of include/lsgpu_icp.h)
static inline void correctTransformationMatrix(PointMatcher::TransformationParameters* transformation_matrix) {
  laser_slam_amd::TransformationParameters T;
  for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) T[(size_t)(4 * c + r)] = (*transformation_matrix)(r, c);
  laser_slam_amd::correctTransformationMatrix(&T);
  for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) (*transformation_matrix)(r, c) = T[(size_t)(4 * c + r)];
}
static inline SE3 convertTransformationMatrixToSE3(const PointMatcher::TransformationParameters& transformation_matrix) {
  float T[16];
  for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) T[4 * c + r] = transformation_matrix(r, c);
  return overlay_detail::fromMirror(laser_slam_amd::SE3::fromTransformationMatrix(T));
}

class LaserTrack {
 public:
  LaserTrack() {}
  explicit LaserTrack(const LaserTrackParams& parameters, unsigned int laser_track_id = 0u)
      : t_(new laser_slam_amd::LaserTrack(parameters, laser_track_id)) {}
  ~LaserTrack() {}

  void processPose(const Pose& pose) { track().processPose(overlay_detail::toMirror(pose)); }                // laser_track.hpp:27
  void processLaserScan(const LaserScan& scan) { track().processLaserScan(mirror(scan)); }                   // :30
  void processPoseAndLaserScan(const Pose& pose, const LaserScan& in_scan, gtsam::NonlinearFactorGraph* newFactors = NULL,
                               gtsam::Values* newValues = NULL, bool* is_prior = NULL) {                     // :33-36
    laser_slam_amd::FactorList f;
    laser_slam_amd::Values v;
    track().processPoseAndLaserScan(overlay_detail::toMirror(pose), mirror(in_scan), newFactors ? &f : nullptr,
                                    newValues ? &v : nullptr, is_prior);
    if (newFactors) for (const auto& r : f) newFactors->push_back(overlay_detail::factorOf(r));
    if (newValues) { newValues->clear(); for (const auto& kv : v) newValues->insert(kv.first, overlay_detail::fromMirror(kv.second)); }
  }

  void getLastPointCloud(DataPoints* out_point_cloud) const { check(out_point_cloud); }                      // :40 ("todo" upstream)
  void getPointCloudOfTimeInterval(const std::pair<Time, Time>&, DataPoints* out_point_cloud) const {       // :43-44 ("todo")
    check(out_point_cloud);
    *out_point_cloud = DataPoints();
  }
  void getLocalCloudInWorldFrame(const Time& timestamp, DataPoints* out_point_cloud) const {                 // :47
    check(out_point_cloud);
    laser_slam_amd::DataPoints d;
    track().getLocalCloudInWorldFrame(timestamp, &d);
    *out_point_cloud = overlay_detail::fromMirror(d);
  }
  // :50 -- a reference to PointMatcher-typed copies of the (filtered) scans the mirror holds; rebuilt lazily when scans
  // were added.  Only the get_laser_track service reads it (laser_slam_worker.cpp:264-281).
  const std::vector<LaserScan>& getLaserScans() const {
    std::lock_guard<std::recursive_mutex> lock(cache_mutex_);
    const auto& scans = track().getLaserScans();
    for (size_t i = laser_scans_cache_.size(); i < scans.size(); ++i)
      laser_scans_cache_.push_back(LaserScan{overlay_detail::fromMirror(scans[i].scan), scans[i].time_ns, scans[i].key});
    return laser_scans_cache_;
  }
  void getTrajectory(Trajectory* trajectory) const {                                                         // :53
    check(trajectory);
    laser_slam_amd::TrajectoryMap m;
    track().getTrajectory(&m);
    trajectory->clear();
    for (const auto& kv : m) trajectory->emplace(kv.first, overlay_detail::fromMirror(kv.second));
  }
  void getOdometryTrajectory(Trajectory* out_trajectory) const {                                             // :56
    check(out_trajectory);
    laser_slam_amd::TrajectoryMap m;
    track().getOdometryTrajectory(&m);
    out_trajectory->clear();
    for (const auto& kv : m) out_trajectory->emplace(kv.first, overlay_detail::fromMirror(kv.second));
  }
  void getCovariances(std::vector<Covariance>* out_covariances) const {                                      // :59
    check(out_covariances);
    std::lock_guard<std::recursive_mutex> lock(cache_mutex_);
    *out_covariances = covariances_;
  }
  Pose getCurrentPose() const { return overlay_detail::fromMirror(track().getCurrentPose()); }               // :62
  Pose getPreviousPose() const { return overlay_detail::fromMirror(track().getPreviousPose()); }             // :64
  Time getMinTime() const { return track().getMinTime(); }                                                   // :67
  Time getMaxTime() const { return track().getMaxTime(); }                                                   // :70
  void getLaserScansTimes(std::vector<Time>* out_times_ns) const { check(out_times_ns); track().getLaserScansTimes(out_times_ns); }  // :73

  // :76-98 -- the batch ("sliding window") graph builders.  Prior: the trajectory's first node at sigma 1e-7, as
  // DiscreteSE3Curve::addPriorFactors does; relative factors: makeRelativeMeasurementFactor with the caller's noise model,
  // first node fixed when it lies outside the window (laser_track.cpp:336-408).
  void appendPriorFactors(const Time& prior_time_ns, gtsam::NonlinearFactorGraph* graph) const {
    check(graph);
    std::array<double, 6> s; s.fill(1e-7);
    graph->push_back(gtsam::ExpressionFactor<SE3>(overlay_detail::noiseOf(s, false), evaluate(prior_time_ns),
                                                  gtsam::Expression<SE3>(track().getValueKey(prior_time_ns))));
  }
  void appendOdometryFactors(const Time& optimization_min_time_ns, const Time& optimization_max_time_ns,
                             gtsam::noiseModel::Base::shared_ptr noise_model, gtsam::NonlinearFactorGraph* graph) const {
    check(graph);
    for (const auto& r : track().getOdometryMeasurements())
      if (r.time_a_ns >= optimization_min_time_ns && r.time_b_ns <= optimization_max_time_ns)
        graph->push_back(relativeFactor(r, noise_model, false));
  }
  void appendICPFactors(const Time& optimization_min_time_ns, const Time& optimization_max_time_ns,
                        gtsam::noiseModel::Base::shared_ptr noise_model, gtsam::NonlinearFactorGraph* graph) const {
    check(graph);
    appendWindowed(track().getIcpTransformations(), optimization_min_time_ns, optimization_max_time_ns, noise_model, graph);
  }
  void appendLoopClosureFactors(const Time&, const Time&, gtsam::noiseModel::Base::shared_ptr, gtsam::NonlinearFactorGraph* graph) const {
    check(graph);   // loop_closures_ is never filled in the reference either (no writer in laser_track.cpp): nothing to append
  }
  void initializeGTSAMValues(const gtsam::KeySet& keys, gtsam::Values* values) const {                        // :97
    check(values);
    laser_slam_amd::TrajectoryMap m;
    track().getTrajectory(&m);
    for (const auto& kv : m) {
      const Key k = track().getValueKey(kv.first);
      if (keys.count(k)) values->insert(k, overlay_detail::fromMirror(kv.second));
    }
  }
  void updateFromGTSAMValues(const gtsam::Values& values) { track().updateFromValues(overlay_detail::toMirror(values)); }   // :101
  void updateCovariancesFromGTSAMValues(const gtsam::NonlinearFactorGraph& factor_graph, const gtsam::Values& values) {    // :104-105
    std::lock_guard<std::recursive_mutex> lock(cache_mutex_);
    gtsam::KeySet keys = factor_graph.keys();
    gtsam::Marginals marginals(factor_graph, values);
    for (const auto& key : keys) covariances_.push_back(marginals.marginalCovariance(key));
  }
  size_t getNumScans() { return track().getNumScans(); }                                                      // :108-111
  void printTrajectory() {}                                                                                   // :114 (debug print)
  Pose findNearestPose(const Time& timestamp_ns) const { return overlay_detail::fromMirror(track().findNearestPose(timestamp_ns)); }  // :121
  void buildSubMapAroundTime(const Time& time_ns, const unsigned int sub_maps_radius, DataPoints* submap_out) const {      // :123-125
    check(submap_out);
    laser_slam_amd::DataPoints d;
    track().buildSubMapAroundTime(time_ns, sub_maps_radius, &d);
    *submap_out = overlay_detail::fromMirror(d);
  }
  gtsam::Expression<SE3> getValueExpression(const Time& time_ns) { return gtsam::Expression<SE3>(track().getValueKey(time_ns)); }   // :127-130
  SE3 evaluate(const Time& time_ns) const { return overlay_detail::fromMirror(track().evaluate(time_ns)); }   // :132-135
  void getScanMatchingTimes(std::map<Time, double>* scan_matching_times) const {                              // :137-140
    check(scan_matching_times);
    *scan_matching_times = track().getScanMatchingTimes();
  }
  void saveTrajectory(const std::string& filename) const {                                                    // :142-144: "time, 4x4 row major" per line
    laser_slam_amd::TrajectoryMap m;
    track().getTrajectory(&m);
    std::ofstream out(filename.c_str());
    for (const auto& kv : m) {
      const auto T = kv.second.transformationMatrixF();   // column major
      out << kv.first;
      for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) out << ", " << T[(size_t)(4 * c + r)];
      out << "\n";
    }
  }

  laser_slam_amd::LaserTrack& mirrorTrack() { return track(); }   // the tested object underneath

 private:
  template <class P> static void check(P* p) { if (!p) throw std::logic_error("null output pointer"); }   // glog CHECK_NOTNULL upstream
  laser_slam_amd::LaserTrack& track() const {
    if (!t_) throw std::logic_error("laser_slam::LaserTrack was default-constructed");
    return *t_;
  }
  static laser_slam_amd::LaserScan mirror(const LaserScan& s) {
    laser_slam_amd::LaserScan o;
    o.scan = overlay_detail::toMirror(s.scan); o.time_ns = s.time_ns; o.key = s.key;
    return o;
  }
  gtsam::ExpressionFactor<SE3> relativeFactor(const laser_slam_amd::RelativePose& r, gtsam::noiseModel::Base::shared_ptr noise_model,
                                              bool fix_first_node) const {                                    // laser_track.cpp:431-451
    const gtsam::Expression<SE3> T_w_b(r.key_b);
    const gtsam::Expression<SE3> T_w_a = fix_first_node ? gtsam::Expression<SE3>(evaluate(r.time_a_ns)) : gtsam::Expression<SE3>(r.key_a);
    return gtsam::ExpressionFactor<SE3>(noise_model, overlay_detail::fromMirror(r.T_a_b),
                                        kindr::minimal::compose(kindr::minimal::inverse(T_w_a), T_w_b));
  }
  void appendWindowed(const std::vector<laser_slam_amd::RelativePose>& rel, Time t_min, Time t_max,
                      gtsam::noiseModel::Base::shared_ptr noise_model, gtsam::NonlinearFactorGraph* graph) const {
    for (const auto& r : rel) {
      if (r.time_b_ns < t_min || r.time_b_ns > t_max) continue;
      const bool a_inside = r.time_a_ns >= t_min && r.time_a_ns <= t_max;
      graph->push_back(relativeFactor(r, noise_model, !a_inside));
    }
  }

  std::shared_ptr<laser_slam_amd::LaserTrack> t_;
  mutable std::recursive_mutex cache_mutex_;
  mutable std::vector<LaserScan> laser_scans_cache_;
  std::vector<Covariance> covariances_;
};

// IncrementalEstimator over the real gtsam::ISAM2; the laser tracks, the loop-closure ICP (device) and the prior-removal
// bookkeeping (laser_slam_amd::WorkerLinks) come from the mirror, the graph is GTSAM's.
class IncrementalEstimator {
 public:
  IncrementalEstimator() {}
  explicit IncrementalEstimator(const EstimatorParams& parameters, unsigned int n_laser_slam_workers = 1u)
      : params_(parameters), icp_(parameters.laser_track_params.device) {
    gtsam::ISAM2Params isam2_params;                      // incremental_estimator.cpp:17-20
    isam2_params.setRelinearizeSkip(1);
    isam2_params.setRelinearizeThreshold(0.001);
    isam2_ = gtsam::ISAM2(isam2_params);
    for (unsigned int i = 0u; i < n_laser_slam_workers; ++i)    // :23-26
      laser_tracks_.push_back(std::make_shared<LaserTrack>(parameters.laser_track_params, i));
    loop_closure_noise_model_ = overlay_detail::noiseOf(params_.loop_closure_noise_model, params_.add_m_estimator_on_loop_closures);   // :29-38
    first_association_noise_model_ = overlay_detail::noiseOf({0.05, 0.05, 0.05, 0.015, 0.015, 0.015}, false);                            // :40-48
    std::ifstream ifs(params_.laser_track_params.icp_configuration_file.c_str());                                                        // :50-60
    if (!params_.laser_track_params.icp_configuration_file.empty() && ifs.good()) icp_.loadFromYaml(ifs);
    else icp_.setDefault();
  }
  ~IncrementalEstimator() {}

  // incremental_estimator.cpp:63-149
  void processLoopClosure(const RelativePose& loop_closure) {
    std::lock_guard<std::recursive_mutex> lock(full_class_mutex_);
    LaserTrack& track_a = *laser_tracks_.at(loop_closure.track_id_a);
    LaserTrack& track_b = *laser_tracks_.at(loop_closure.track_id_b);
    if (loop_closure.track_id_a == loop_closure.track_id_b && !(loop_closure.time_a_ns < loop_closure.time_b_ns))
      throw std::logic_error("Loop closure has invalid time.");
    if (loop_closure.time_a_ns < track_a.getMinTime() || loop_closure.time_a_ns > track_a.getMaxTime() ||
        loop_closure.time_b_ns < track_b.getMinTime() || loop_closure.time_b_ns > track_b.getMaxTime())
      throw std::logic_error("Loop closure has invalid time.");

    RelativePose updated_loop_closure = loop_closure;
    // w_T_a_b aligns the source cloud with the target cloud in the world frame -> frame of a (:81-88)
    const SE3 T_w_a = track_a.evaluate(loop_closure.time_a_ns);
    const SE3 T_w_b = track_b.evaluate(loop_closure.time_b_ns);
    updated_loop_closure.T_a_b = T_w_a.inverse() * loop_closure.T_a_b * T_w_b;

    if (params_.do_icp_step_on_loop_closures) {   // :91-115, the sub-maps stay in the mirror's cloud type: no PointMatcher copy
      laser_slam_amd::TransformationParameters initial_guess = overlay_detail::toMirror(updated_loop_closure.T_a_b).transformationMatrixF();
      laser_slam_amd::DataPoints sub_map_a, sub_map_b;
      track_a.mirrorTrack().buildSubMapAroundTime(loop_closure.time_a_ns, params_.loop_closures_sub_maps_radius, &sub_map_a);
      track_b.mirrorTrack().buildSubMapAroundTime(loop_closure.time_b_ns, params_.loop_closures_sub_maps_radius, &sub_map_b);
      // (the guess goes in uncorrected and exceptions of icp_.compute propagate, as at incremental_estimator.cpp:89-108)
      const laser_slam_amd::TransformationParameters icp_solution = icp_.compute(sub_map_b, sub_map_a, initial_guess);
      updated_loop_closure.T_a_b = overlay_detail::fromMirror(laser_slam_amd::SE3::fromTransformationMatrix(icp_solution.data()));
    }

    gtsam::NonlinearFactorGraph new_factors, new_associations_factors;   // :117-132
    gtsam::Expression<SE3> exp_T_w_b(track_b.getValueExpression(updated_loop_closure.time_b_ns));
    gtsam::Expression<SE3> exp_T_w_a(track_a.getValueExpression(updated_loop_closure.time_a_ns));
    gtsam::Expression<SE3> exp_T_a_w(kindr::minimal::inverse(exp_T_w_a));
    gtsam::Expression<SE3> exp_relative(kindr::minimal::compose(exp_T_a_w, exp_T_w_b));
    new_factors.push_back(gtsam::ExpressionFactor<SE3>(loop_closure_noise_model_, updated_loop_closure.T_a_b, exp_relative));
    new_associations_factors.push_back(gtsam::ExpressionFactor<SE3>(first_association_noise_model_, updated_loop_closure.T_a_b, exp_relative));

    std::vector<unsigned int> affected_worker_ids;                        // :134-142
    affected_worker_ids.push_back(loop_closure.track_id_a);
    affected_worker_ids.push_back(loop_closure.track_id_b);
    gtsam::Values new_values;
    gtsam::Values result = estimateAndRemove(new_factors, new_associations_factors, new_values, affected_worker_ids,
                                             updated_loop_closure.time_b_ns);
    for (auto& track : laser_tracks_) track->updateFromGTSAMValues(result);   // :144-148
  }

  Pose getCurrentPose(unsigned int laser_track_id = 0u) const {            // incremental_estimator.hpp:30-33
    std::lock_guard<std::recursive_mutex> lock(full_class_mutex_);
    return laser_tracks_.at(laser_track_id)->getCurrentPose();
  }
  std::shared_ptr<LaserTrack> getLaserTrack(unsigned int laser_track_id) {  // incremental_estimator.cpp:293-298
    std::lock_guard<std::recursive_mutex> lock(full_class_mutex_);
    return laser_tracks_.at(laser_track_id);
  }
  std::vector<std::shared_ptr<LaserTrack> > getAllLaserTracks() {           // :300-303
    std::lock_guard<std::recursive_mutex> lock(full_class_mutex_);
    return laser_tracks_;
  }

  // :151-163: update with the new factors, two more updates, the whole estimate
  gtsam::Values estimate(const gtsam::NonlinearFactorGraph& new_factors, const gtsam::Values& new_values, Time /*timestamp_ns*/ = 0u) {
    std::lock_guard<std::recursive_mutex> lock(full_class_mutex_);
    isam2_.update(new_factors, new_values);
    isam2_.update();
    isam2_.update();
    return isam2_.calculateEstimate();
  }
  // :165-266: the first loop closure between two robots removes the absorbed group's prior (WorkerLinks::link) and adds
  // the first-association factors instead of the loop-closure factors
  gtsam::Values estimateAndRemove(const gtsam::NonlinearFactorGraph& new_factors,
                                  const gtsam::NonlinearFactorGraph& new_associations_factors, const gtsam::Values& new_values,
                                  const std::vector<unsigned int>& affected_worker_ids, Time /*timestamp_ns*/ = 0u) {
    std::lock_guard<std::recursive_mutex> lock(full_class_mutex_);
    const std::vector<size_t> to_remove = links_.link(affected_worker_ids);
    gtsam::FactorIndices factor_indices_to_remove(to_remove.begin(), to_remove.end());
    isam2_.update(factor_indices_to_remove.empty() ? new_factors : new_associations_factors, new_values, factor_indices_to_remove);
    isam2_.update();
    isam2_.update();
    return isam2_.calculateEstimate();
  }
  // :268-291
  gtsam::Values registerPrior(const gtsam::NonlinearFactorGraph& new_factors, const gtsam::Values& new_values, const unsigned int worker_id) {
    std::lock_guard<std::recursive_mutex> lock(full_class_mutex_);
    const gtsam::ISAM2Result update_result = isam2_.update(new_factors, new_values);
    if (update_result.newFactorsIndices.size() != 1u) throw std::logic_error("registerPrior expects exactly one factor");
    links_.registerPrior(worker_id, update_result.newFactorsIndices.at(0u));
    isam2_.update();
    isam2_.update();
    return isam2_.calculateEstimate();
  }

 private:
  EstimatorParams params_;
  mutable std::recursive_mutex full_class_mutex_;
  std::vector<std::shared_ptr<LaserTrack> > laser_tracks_;
  gtsam::ISAM2 isam2_;
  laser_slam_amd::ICP icp_;                                 // the loop-closure ICP: second call site of icp_.compute (:108)
  gtsam::noiseModel::Base::shared_ptr loop_closure_noise_model_;
  gtsam::noiseModel::Base::shared_ptr first_association_noise_model_;
  laser_slam_amd::WorkerLinks links_;
};

}  // namespace laser_slam
