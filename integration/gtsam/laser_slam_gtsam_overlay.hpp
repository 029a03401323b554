// laser_slam_gtsam_overlay.hpp -- seam B1 with the REAL types: the public API laser_slam_ros compiles
// against, i.e. LaserTrack::processPoseAndLaserScan(const Pose&, const LaserScan&, gtsam::NonlinearFactorGraph*,
// gtsam::Values*, bool*) (laser_slam/include/laser_slam/laser_track.hpp:33-36), updateFromGTSAMValues (:101),
// getLocalCloudInWorldFrame (:47), getTrajectory (:53), getCurrentPose (:62), buildSubMapAroundTime (:123-125) ... and
// IncrementalEstimator's estimate / estimateAndRemove / registerPrior / processLoopClosure over gtsam::ISAM2
// (incremental_estimator.hpp:20-53), implemented ON TOP of the dependency-free mirror in laser_slam_amd/cpp: the mirror
// does the bookkeeping and the device ICP, this overlay converts its plain-data factors into the
// gtsam::ExpressionFactor<SE3> objects the reference emits (laser_track.cpp:431-458) and its SE3 into
// kindr::minimal::QuatTransformation.
//
// Built only with -DLSGPU_WITH_GTSAM=ON (integration/CMakeLists.txt): GTSAM, minkindr, minkindr_gtsam and
// libpointmatcher are NOT installed in the image this repository is developed in, so this file is not compiled or
// tested here; everything beneath it (laser_slam_amd::LaserTrack / IncrementalEstimator / ICP) is.
#pragma once
#include <gtsam/nonlinear/ExpressionFactor.h>
#include <gtsam/nonlinear/ISAM2.h>
#include <gtsam/nonlinear/NonlinearFactorGraph.h>
#include <gtsam/nonlinear/Values.h>
#include <kindr/minimal/quat-transformation.h>
#include <kindr/minimal/quat-transformation-gtsam.h>
#include <pointmatcher/PointMatcher.h>

#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "laser_slam_amd/incremental_estimator.hpp"

namespace laser_slam {   // the reference's own namespace: laser_slam_ros sees the types it expects

typedef PointMatcher<float> PointMatcher;                                   // common.hpp:14
typedef typename PointMatcher::DataPoints DataPoints;                       // common.hpp:15
typedef kindr::minimal::QuatTransformationTemplate<double> SE3;             // common.hpp:17
typedef int64_t Time;                                                       // curves::Time
typedef size_t Key;
struct Pose { SE3 T_w; Time time_ns; Key key; };                            // common.hpp:87-94
struct RelativePose { SE3 T_a_b; Time time_a_ns, time_b_ns; Key key_a, key_b; unsigned int track_id_a, track_id_b; };
struct LaserScan { DataPoints scan; Time time_ns; Key key; };               // common.hpp:113-120
typedef std::map<Time, SE3> Trajectory;
using LaserTrackParams = laser_slam_amd::LaserTrackParams;                  // parameters.hpp:8-23 (same fields)
using EstimatorParams = laser_slam_amd::EstimatorParams;                    // parameters.hpp:25-34

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
inline m::DataPoints toMirror(const DataPoints& c) {     // features are (dim+1) x N column major: a plain copy
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
inline gtsam::noiseModel::Base::shared_ptr noiseOf(const m::Factor& f) {   // laser_track.cpp:37-64
  gtsam::Vector6 s;
  for (int i = 0; i < 6; ++i) s[i] = f.sigmas[(size_t)i];
  gtsam::noiseModel::Base::shared_ptr n = gtsam::noiseModel::Diagonal::Sigmas(s);
  if (f.cauchy) n = gtsam::noiseModel::Robust::Create(gtsam::noiseModel::mEstimator::Cauchy::Create(1), n);
  return n;
}
// one gtsam::ExpressionFactor<SE3> per mirror factor record (makeMeasurementFactor / makeRelativeMeasurementFactor)
inline gtsam::ExpressionFactor<SE3> factorOf(const m::Factor& f) {
  using gtsam::Expression;
  if (f.type == m::Factor::PRIOR) return gtsam::ExpressionFactor<SE3>(noiseOf(f), fromMirror(f.measurement), Expression<SE3>(f.key_b));
  const Expression<SE3> T_w_b(f.key_b);
  const Expression<SE3> T_w_a = f.fix_first_node ? Expression<SE3>(fromMirror(f.fixed_a)) : Expression<SE3>(f.key_a);
  return gtsam::ExpressionFactor<SE3>(noiseOf(f), fromMirror(f.measurement),
                                      kindr::minimal::compose(kindr::minimal::inverse(T_w_a), T_w_b));
}
inline m::Values toMirror(const gtsam::Values& v) {
  m::Values out;
  for (const auto& kv : v) out[(m::Key)kv.key] = toMirror(kv.value.cast<SE3>());
  return out;
}
}  // namespace overlay_detail

class LaserTrack {
 public:
  explicit LaserTrack(const LaserTrackParams& parameters, unsigned int laser_track_id = 0u) : t_(parameters, laser_track_id) {}

  void processPose(const Pose& pose) { t_.processPose(mirror(pose)); }
  void processLaserScan(const LaserScan& scan) { t_.processLaserScan(mirror(scan)); }
  void processPoseAndLaserScan(const Pose& pose, const LaserScan& in_scan, gtsam::NonlinearFactorGraph* newFactors = NULL,
                               gtsam::Values* newValues = NULL, bool* is_prior = NULL) {   // laser_track.hpp:33-36
    laser_slam_amd::FactorList f;
    laser_slam_amd::Values v;
    t_.processPoseAndLaserScan(mirror(pose), mirror(in_scan), newFactors ? &f : nullptr, newValues ? &v : nullptr, is_prior);
    if (newFactors) for (const auto& r : f) newFactors->push_back(overlay_detail::factorOf(r));
    if (newValues) { newValues->clear(); for (const auto& kv : v) newValues->insert(kv.first, overlay_detail::fromMirror(kv.second)); }
  }
  void getLocalCloudInWorldFrame(const Time& timestamp, DataPoints* out) const {
    laser_slam_amd::DataPoints d;
    t_.getLocalCloudInWorldFrame(timestamp, &d);
    *out = overlay_detail::fromMirror(d);
  }
  void getTrajectory(Trajectory* trajectory) const {
    laser_slam_amd::TrajectoryMap m;
    t_.getTrajectory(&m);
    trajectory->clear();
    for (const auto& kv : m) trajectory->emplace(kv.first, overlay_detail::fromMirror(kv.second));
  }
  Pose getCurrentPose() const { const auto p = t_.getCurrentPose(); return Pose{overlay_detail::fromMirror(p.T_w), p.time_ns, p.key}; }
  Time getMinTime() const { return t_.getMinTime(); }
  Time getMaxTime() const { return t_.getMaxTime(); }
  size_t getNumScans() { return t_.getNumScans(); }
  SE3 evaluate(const Time& time_ns) const { return overlay_detail::fromMirror(t_.evaluate(time_ns)); }
  gtsam::Expression<SE3> getValueExpression(const Time& time_ns) { return gtsam::Expression<SE3>(t_.getValueKey(time_ns)); }
  void updateFromGTSAMValues(const gtsam::Values& values) { t_.updateFromValues(overlay_detail::toMirror(values)); }
  void buildSubMapAroundTime(const Time& time_ns, const unsigned int sub_maps_radius, DataPoints* submap_out) const {
    laser_slam_amd::DataPoints d;
    t_.buildSubMapAroundTime(time_ns, sub_maps_radius, &d);
    *submap_out = overlay_detail::fromMirror(d);
  }
  void getScanMatchingTimes(std::map<Time, double>* out) const { *out = t_.getScanMatchingTimes(); }
  laser_slam_amd::LaserTrack& mirrorTrack() { return t_; }

 private:
  static laser_slam_amd::Pose mirror(const Pose& p) { laser_slam_amd::Pose o; o.T_w = overlay_detail::toMirror(p.T_w); o.time_ns = p.time_ns; o.key = p.key; return o; }
  static laser_slam_amd::LaserScan mirror(const LaserScan& s) { laser_slam_amd::LaserScan o; o.scan = overlay_detail::toMirror(s.scan); o.time_ns = s.time_ns; o.key = s.key; return o; }
  mutable laser_slam_amd::LaserTrack t_;
};

// IncrementalEstimator over the real gtsam::ISAM2 (incremental_estimator.cpp:12-61, 151-163, 165-266, 268-291); the laser
// tracks and the loop-closure ICP come from the mirror (device ICP), the graph is GTSAM's.
class IncrementalEstimator {
 public:
  explicit IncrementalEstimator(const EstimatorParams& parameters, unsigned int n_laser_slam_workers = 1u) : params_(parameters) {
    gtsam::ISAM2Params isam2_params;                      // incremental_estimator.cpp:17-20
    isam2_params.setRelinearizeSkip(1);
    isam2_params.setRelinearizeThreshold(0.001);
    isam2_ = gtsam::ISAM2(isam2_params);
    for (unsigned int i = 0u; i < n_laser_slam_workers; ++i) laser_tracks_.push_back(std::make_shared<LaserTrack>(parameters.laser_track_params, i));
  }
  std::shared_ptr<LaserTrack> getLaserTrack(unsigned int id) { std::lock_guard<std::recursive_mutex> l(mutex_); return laser_tracks_.at(id); }
  std::vector<std::shared_ptr<LaserTrack>> getAllLaserTracks() { std::lock_guard<std::recursive_mutex> l(mutex_); return laser_tracks_; }
  Pose getCurrentPose(unsigned int id = 0u) const { std::lock_guard<std::recursive_mutex> l(mutex_); return laser_tracks_.at(id)->getCurrentPose(); }

  gtsam::Values estimate(const gtsam::NonlinearFactorGraph& new_factors, const gtsam::Values& new_values, Time = 0u) {
    std::lock_guard<std::recursive_mutex> l(mutex_);     // incremental_estimator.cpp:151-163
    isam2_.update(new_factors, new_values);
    isam2_.update();
    isam2_.update();
    return isam2_.calculateEstimate();
  }
  gtsam::Values registerPrior(const gtsam::NonlinearFactorGraph& new_factors, const gtsam::Values& new_values, const unsigned int worker_id) {
    std::lock_guard<std::recursive_mutex> l(mutex_);     // incremental_estimator.cpp:268-291
    const gtsam::ISAM2Result r = isam2_.update(new_factors, new_values);
    if (worker_id > 0u) factor_indices_to_remove_[worker_id] = r.newFactorsIndices.at(0u);
    linked_workers_.push_back({worker_id});
    isam2_.update();
    isam2_.update();
    return isam2_.calculateEstimate();
  }
  // estimateAndRemove / processLoopClosure follow incremental_estimator.cpp:63-149,165-266 line by line in behaviour; the
  // bookkeeping (which prior to drop when two robots' graphs first link) is the one tested in the mirror
  // (laser_slam_amd/cpp/include/laser_slam_amd/incremental_estimator.hpp:107-165), with FactorIndices handed to isam2_.update.

 private:
  EstimatorParams params_;
  mutable std::recursive_mutex mutex_;
  std::vector<std::shared_ptr<LaserTrack>> laser_tracks_;
  gtsam::ISAM2 isam2_;
  std::unordered_map<unsigned int, size_t> factor_indices_to_remove_;
  std::vector<std::vector<unsigned int>> linked_workers_;
};

}  // namespace laser_slam
