// laser_track.hpp -- dependency-free C++ mirror of the part of laser_slam::LaserTrack that sits on
// the ICP hot path (§8b seam B1).
//
// Reference: laser_slam/include/laser_slam/laser_track.hpp:17-236, laser_slam/src/laser_track.cpp.
// Same method names, argument meaning and error behaviour for
//   processPose / processLaserScan / processPoseAndLaserScan      laser_track.cpp:66-231
//   computeICPTransformations / localScanToSubMap                 laser_track.cpp:460-519
//   buildSubMapAroundTime                                         laser_track.cpp:602-651
//   getLocalCloudInWorldFrame                                     laser_track.cpp:247-266
//   getTrajectory / getCurrentPose / getMin/MaxTime / getNumScans / evaluate / getLaserScans
// What differs, because GTSAM, mincurves and minkindr are not available:
//   * gtsam::NonlinearFactorGraph / gtsam::Values become the plain-data FactorList / Values below
//     (one record per ExpressionFactor<SE3> the reference would emit, laser_track.cpp:431-458);
//   * curves::DiscreteSE3Curve becomes Trajectory (time-keyed SE3 nodes, interpolating evaluate);
//   * glog CHECKs become std::logic_error.
// The ICP itself runs on the GPU through laser_slam_amd::ICP (icp.hpp -> include/lsgpu_icp.h).
#pragma once
#include <algorithm>
#include <chrono>
#include <fstream>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <thread>
#include <vector>

#include "laser_slam_amd/icp.hpp"
#include "laser_slam_amd/cloud_io.hpp"
#include "laser_slam_amd/se3.hpp"
#include "laser_slam_amd/pose_graph.hpp"

namespace laser_slam_amd {

using Time = int64_t;  // curves::Time, nanoseconds

struct Pose {          // laser_slam/include/laser_slam/common.hpp:87-94
  SE3 T_w;
  Time time_ns = 0;
  Key key = 0;
};
struct RelativePose {  // common.hpp:97-110
  SE3 T_a_b;
  Time time_a_ns = 0, time_b_ns = 0;
  Key key_a = 0, key_b = 0;
  unsigned int track_id_a = 0, track_id_b = 0;
};
struct LaserScan {     // common.hpp:113-120
  DataPoints scan;
  Time time_ns = 0;
  Key key = 0;
};

struct LaserTrackParams {  // laser_slam/include/laser_slam/parameters.hpp:8-23
  std::array<double, 6> odometry_noise_model{};
  std::array<double, 6> icp_noise_model{};
  bool add_m_estimator_on_odom = false;
  bool add_m_estimator_on_icp = false;
  std::string icp_configuration_file;
  std::string icp_input_filters_file;   // input filter chain (K0), a libpointmatcher DataPointsFilters YAML list.  As in
                                        // the reference the file MUST be readable (laser_track.cpp:24-30 LOG(FATAL)s
                                        // otherwise); a file without modules is an empty chain
  bool use_icp_factors = true;
  bool use_odom_factors = true;
  int nscan_in_sub_map = 3;
  bool save_icp_results = false;        // laser_track.cpp:504-513: dump the reading, the sub-map and the reading moved by the
                                        // guess / by the solution as .vtk after every ICP (cloud_io.hpp)
  std::string save_icp_results_dir = "/tmp";   // (the reference's paths are /tmp/last_scan.vtk ...; not a reference parameter)
  bool force_priors = false;
  int device = 0;                       // HIP device of this track's ICP handle
  int scans_on_device = 16;             // most recent scans kept in HBM for sub-map assembly (0: host assembly)
  bool overlap_scan_copy = true;        // processPoseAndLaserScan with an empty input chain: the host copy of the scan is made
                                        // beside the device's registration instead of in front of it (not a reference parameter)
};

// Factor / FactorList / Values: pose_graph.hpp
using TrajectoryMap = std::map<Time, SE3>;

// curves::DiscreteSE3Curve stand-in
class Trajectory {
 public:
  bool isEmpty() const { return nodes_.empty(); }
  Key extend(Time t, const SE3& v) {
    if (!nodes_.empty() && t <= nodes_.back().t) throw std::logic_error("trajectory times must increase");
    nodes_.push_back({t, v, next_key_});
    return next_key_++;
  }
  SE3 evaluate(Time t) const {
    if (nodes_.empty()) throw std::logic_error("empty trajectory");
    if (t <= nodes_.front().t) return nodes_.front().v;
    if (t >= nodes_.back().t) return nodes_.back().v;
    auto hi = std::lower_bound(nodes_.begin(), nodes_.end(), t, [](const Node& n, Time x) { return n.t < x; });
    if (hi->t == t) return hi->v;
    auto lo = hi - 1;
    return SE3::interpolate(lo->v, hi->v, double(t - lo->t) / double(hi->t - lo->t));
  }
  Time getMinTime() const { return nodes_.empty() ? 0 : nodes_.front().t; }
  Time getMaxTime() const { return nodes_.empty() ? 0 : nodes_.back().t; }
  void getCurveTimes(std::vector<Time>* out) const { out->clear(); for (auto& n : nodes_) out->push_back(n.t); }
  void update(const Values& v) { for (auto& n : nodes_) { auto it = v.find(n.key); if (it != v.end()) n.v = it->second; } }
  void setFirstKey(Key k) { if (nodes_.empty()) next_key_ = k; }
  // key of the node at exactly time t (DiscreteSE3Curve::getValueExpression needs an existing node)
  Key keyAt(Time t) const {
    auto it = std::lower_bound(nodes_.begin(), nodes_.end(), t, [](const Node& n, Time x) { return n.t < x; });
    if (it == nodes_.end() || it->t != t) throw std::logic_error("no trajectory node at the requested time");
    return it->key;
  }

 private:
  struct Node { Time t; SE3 v; Key key; };
  std::vector<Node> nodes_;
  Key next_key_ = 0;
};

class LaserTrack {
 public:
  explicit LaserTrack(const LaserTrackParams& parameters, unsigned int laser_track_id = 0u)
      : params_(parameters), laser_track_id_(laser_track_id), icp_(parameters.device) {
    // laser_track.cpp:14-21: YAML if readable, otherwise the default chain
    std::ifstream ifs(params_.icp_configuration_file.c_str());
    if (!params_.icp_configuration_file.empty() && ifs.good()) icp_.loadFromYaml(ifs);
    else icp_.setDefault();
    // laser_track.cpp:24-30: the input filter chain; LOG(FATAL) upstream if the file cannot be opened
    std::ifstream ifs_filters(params_.icp_input_filters_file.c_str());
    if (!ifs_filters.good()) throw ConfigError("Could not open ICP input filters configuration file.");
    input_filters_ = DataPointsFilters(ifs_filters, params_.device);
    trajectory_.setFirstKey((Key)laser_track_id_ << 40);  // per-track key range
  }

  void processPose(const Pose& pose) {  // laser_track.cpp:66-72
    std::lock_guard<std::recursive_mutex> lock(mutex_);
    pose_measurements_.push_back(pose);
  }

  void processLaserScan(const LaserScan& in_scan) {  // laser_track.cpp:74-120
    std::lock_guard<std::recursive_mutex> lock(mutex_);
    LaserScan scan = in_scan;
    input_filters_.apply(scan.scan);  // laser_track.cpp:81
    // (this entry point runs the ICP BEFORE it stores the new scan, laser_track.cpp:112-119: the transformation it
    // records matches the previously stored scan against its predecessors)
    registerScan(&scan, nullptr, /*icp_before_store=*/true);
  }

  void processPoseAndLaserScan(const Pose& pose, const LaserScan& in_scan, FactorList* newFactors = nullptr,
                               Values* newValues = nullptr, bool* is_prior = nullptr) {  // :122-231
    std::lock_guard<std::recursive_mutex> lock(mutex_);
    const auto t0 = std::chrono::steady_clock::now();
    if (newFactors && !newFactors->empty()) throw std::logic_error("newFactors must be empty");
    if (newValues) newValues->clear();
    // The working copy of the scan (laser_track.cpp:145-146: `LaserScan scan = in_scan` + the input filters; it ends up in
    // laser_scans_).  16 MB for 1 M points, 2.5 ms of page faults and memcpy on the host -- a third of what the device needs
    // for the whole registration.  With an empty input chain the copy IS in_scan's points, so it is made on a helper thread
    // while the device registers in_scan's own buffer (valid for the whole call), and joins laser_scans_.back() before
    // anything else looks at it.  Same results, same state afterwards; any other configuration copies first, as upstream.
    LaserScan scan;
    scan.time_ns = in_scan.time_ns; scan.key = in_scan.key;
    bool overlap = params_.overlap_scan_copy && input_filters_.empty() && params_.scans_on_device > 0 && params_.use_icp_factors &&
                   !params_.save_icp_results && !icp_.hasComputeOverride() && !trajectory_.isEmpty();
#ifdef LSGPU_TEST_SEAMS
    overlap = overlap && !icp_.hasComputeObserver();
#endif
    struct CopyJoin {   // joins on every path out of this call and puts the points where they belong
      LaserTrack* self; size_t n_before; std::thread t; DataPoints points; bool armed = false;
      ~CopyJoin() {   // (an exception on the way: the scan, if it was stored, still gets its points)
        if (t.joinable()) t.join();
        self->pending_reading_ = nullptr;
        if (armed && self->laser_scans_.size() == n_before + 1) self->laser_scans_.back().scan = std::move(points);
      }
    } copy{this, laser_scans_.size(), {}, {}};
    if (overlap) {
      try {
        copy.t = std::thread([&copy, &in_scan] { copy.points = in_scan.scan; });
      } catch (const std::system_error&) { overlap = false; }   // no thread to be had: copy here
    }
    if (overlap) {
      pending_reading_ = &in_scan.scan;
      copy.armed = true;   // (registerScan below stores the scan, points to follow, before it runs the ICP)
    } else {
      scan.scan = in_scan.scan;
      input_filters_.apply(scan.scan);  // laser_track.cpp:146
    }
    stage_times_ = StageTimes{};
    stage_times_.copy_ms = msSince(t0);
    pose_measurements_.push_back(pose);
    const bool first = trajectory_.isEmpty();
    RelativePose odom;
    registerScan(&scan, &odom, /*icp_before_store=*/false);
    if (overlap) {   // the points join the stored scan here (not at the end of the call: the factors below do not need them, a reader of laser_scans_ does)
      const auto tj = std::chrono::steady_clock::now();
      copy.t.join();
      pending_reading_ = nullptr;
      laser_scans_.back().scan = std::move(copy.points);
      copy.armed = false;
      stage_times_.copy_ms += msSince(tj);
    }
    if (first) {
      if (newFactors) {
        Pose prior = pose;
        if (params_.force_priors)  // laser_track.cpp:166-170: priors 100 m apart along y, one per track
          prior.T_w = SE3({1, 0, 0, 0}, {0.0, kDistanceBetweenPriorPoses_m * laser_track_id_, 0.0});
        Factor f{Factor::PRIOR, 0, scan.key, prior.T_w, {}, false};
        f.sigmas.fill(kPriorSigma);
        newFactors->push_back(f);
      }
      if (is_prior) *is_prior = true;
    } else {
      scan_matching_times_[scan.time_ns] =
          std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (newFactors) {
        if (params_.use_odom_factors)
          newFactors->push_back({Factor::ODOMETRY, odom.key_a, odom.key_b, odom.T_a_b,
                                 params_.odometry_noise_model, params_.add_m_estimator_on_odom});
        if (params_.use_icp_factors && !icp_transformations_.empty()) {
          const RelativePose& r = icp_transformations_.back();
          newFactors->push_back({Factor::ICP, r.key_a, r.key_b, r.T_a_b, params_.icp_noise_model,
                                 params_.add_m_estimator_on_icp});
        }
      }
      if (is_prior) *is_prior = false;
    }
    if (newValues) (*newValues)[scan.key] = pose.T_w;
  }

  void getLocalCloudInWorldFrame(const Time& timestamp_ns, DataPoints* out) const {  // :247-266
    std::lock_guard<std::recursive_mutex> lock(mutex_);
    if (!out) throw std::logic_error("null output");
    auto it = scanAtTime(timestamp_ns);
    TransformationParameters T = trajectory_.evaluate(timestamp_ns).transformationMatrixF();
    correctTransformationMatrix(&T);
    *out = RigidTransformation::compute(it->scan, T);
  }

  const std::vector<LaserScan>& getLaserScans() const { return laser_scans_; }

  void getTrajectory(TrajectoryMap* trajectory) const {  // :268-279
    std::lock_guard<std::recursive_mutex> lock(mutex_);
    trajectory->clear();
    std::vector<Time> times;
    trajectory_.getCurveTimes(&times);
    for (Time t : times) trajectory->emplace(t, trajectory_.evaluate(t));
  }

  // laser_track.cpp:310-316: the pose MEASUREMENTS as they came in (odometry only, no ICP, no graph), by time stamp.
  // laser_slam_ros publishes it beside the estimated trajectory (laser_slam_worker.cpp:519).
  void getOdometryTrajectory(TrajectoryMap* trajectory) const {
    std::lock_guard<std::recursive_mutex> lock(mutex_);
    if (!trajectory) throw std::logic_error("null output");
    trajectory->clear();
    for (const Pose& pose : pose_measurements_) trajectory->emplace(pose.time_ns, pose.T_w);  // (emplace: first one wins)
  }

  Pose getPreviousPose() const {  // :302-312: the node before the last one, default Pose while there is none
    std::lock_guard<std::recursive_mutex> lock(mutex_);
    Pose p;
    std::vector<Time> times;
    trajectory_.getCurveTimes(&times);
    if (times.size() > 1u) { p.time_ns = times[times.size() - 2]; p.T_w = trajectory_.evaluate(p.time_ns); }
    return p;
  }

  void getLaserScansTimes(std::vector<Time>* out_times_ns) const {  // :328-334
    std::lock_guard<std::recursive_mutex> lock(mutex_);
    if (!out_times_ns) throw std::logic_error("null output");
    out_times_ns->clear();
    for (const LaserScan& s : laser_scans_) out_times_ns->push_back(s.time_ns);
  }

  // :557-572: the trajectory evaluated at the requested time (despite the name), which must not be later than the
  // latest pose measurement; the key is not filled in
  Pose findNearestPose(const Time& timestamp_ns) const {
    std::lock_guard<std::recursive_mutex> lock(mutex_);
    if (pose_measurements_.empty()) throw std::logic_error("Cannot find nearest pose as no pose was registered.");
    if (timestamp_ns > pose_measurements_.back().time_ns)
      throw std::logic_error("The requested time is later than the latest pose time.");
    Pose p;
    p.time_ns = timestamp_ns;
    p.T_w = trajectory_.evaluate(timestamp_ns);
    return p;
  }

  Pose getCurrentPose() const {  // :292-300
    std::lock_guard<std::recursive_mutex> lock(mutex_);
    Pose p;
    if (!trajectory_.isEmpty()) { p.time_ns = trajectory_.getMaxTime(); p.T_w = trajectory_.evaluate(p.time_ns); }
    return p;
  }
  Time getMinTime() const { std::lock_guard<std::recursive_mutex> l(mutex_); return trajectory_.getMinTime(); }
  Time getMaxTime() const { std::lock_guard<std::recursive_mutex> l(mutex_); return trajectory_.getMaxTime(); }
  size_t getNumScans() const { std::lock_guard<std::recursive_mutex> l(mutex_); return laser_scans_.size(); }
  SE3 evaluate(const Time& t) const { std::lock_guard<std::recursive_mutex> l(mutex_); return trajectory_.evaluate(t); }
  // key of the trajectory node at time t: what getValueExpression(t) binds a factor to (laser_track.cpp:435)
  Key getValueKey(const Time& t) const { std::lock_guard<std::recursive_mutex> l(mutex_); return trajectory_.keyAt(t); }
  unsigned int getId() const { return laser_track_id_; }
  void updateFromValues(const Values& v) { std::lock_guard<std::recursive_mutex> l(mutex_); trajectory_.update(v); }
  const std::map<Time, double>& getScanMatchingTimes() const { return scan_matching_times_; }
  const std::vector<RelativePose>& getIcpTransformations() const { return icp_transformations_; }
  const std::vector<RelativePose>& getOdometryMeasurements() const { return odometry_measurements_; }
  const lsgpu_icp_stats& lastIcpStats() const { return icp_.lastStats(); }
  // where the wall time of the last processPoseAndLaserScan went (milliseconds; not a reference accessor): the working copy
  // of the scan + the input filters, the scans that had to cross PCIe, icp_.compute on the resident clouds
  struct StageTimes { double copy_ms = 0, upload_ms = 0, icp_ms = 0; };
  const StageTimes& lastStageTimes() const { return stage_times_; }
  ICP& icp() { return icp_; }  // configuration access (seed, test seam); the reference keeps icp_ private
  DataPointsFilters& inputFilters() { return input_filters_; }

  // laser_track.cpp:602-651: the scan at time_ns plus up to `radius` scans on either side, in its frame
  void buildSubMapAroundTime(const Time& time_ns, unsigned int sub_maps_radius, DataPoints* sub_map_out) const {
    std::lock_guard<std::recursive_mutex> lock(mutex_);
    if (!sub_map_out) throw std::logic_error("null output");
    const SE3 T_w_a = trajectory_.evaluate(time_ns);
    auto it = scanAtTime(time_ns);
    DataPoints sub_map = it->scan;
    auto add = [&](std::vector<LaserScan>::const_iterator s) {
      TransformationParameters T = (T_w_a.inverse() * trajectory_.evaluate(s->time_ns)).transformationMatrixF();
      correctTransformationMatrix(&T);
      sub_map.concatenate(RigidTransformation::compute(s->scan, T));
    };
    auto before = it;
    for (unsigned int i = 0; i < sub_maps_radius && before != laser_scans_.begin(); ++i) add(--before);
    auto after = it;
    for (unsigned int i = 0; i < sub_maps_radius; ++i) {
      if (++after == laser_scans_.end()) break;
      add(after);
    }
    *sub_map_out = sub_map;
  }

 private:
  static constexpr double kDistanceBetweenPriorPoses_m = 100.0;  // laser_track.hpp:235
  static double msSince(std::chrono::steady_clock::time_point t) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
  }
  static constexpr double kPriorSigma = 1e-7;                    // laser_track.cpp:56-64

  SE3 getPoseMeasurement(Time t) const {  // findPose (:521-555): exact time stamp required
    for (auto it = pose_measurements_.rbegin(); it != pose_measurements_.rend(); ++it)
      if (it->time_ns == t) return it->T_w;
    throw std::logic_error("The requested time does not exist in the pose measurements.");
  }
  void setPoseKey(Time t, Key k) {
    for (auto it = pose_measurements_.rbegin(); it != pose_measurements_.rend(); ++it)
      if (it->time_ns == t) { it->key = k; return; }
  }
  Key getPoseKey(Time t) const {
    for (auto it = pose_measurements_.rbegin(); it != pose_measurements_.rend(); ++it)
      if (it->time_ns == t) return it->key;
    throw std::logic_error("no pose key at the requested time");
  }
  std::vector<LaserScan>::const_iterator scanAtTime(Time t) const {  // :584-600
    for (auto it = laser_scans_.begin(); it != laser_scans_.end(); ++it)
      if (it->time_ns == t) return it;
    throw std::logic_error("Could not find the scan.");
  }

  // shared tail of processLaserScan (:86-120: ICP, then store) and processPoseAndLaserScan (:154-206: store, then ICP).
  // The caller's working copy is MOVED into laser_scans_ (its points are gone afterwards, time and key stay valid): a
  // second 16 MB copy per 1 M-point scan cost more host time than the whole device side of the registration.
  void registerScan(LaserScan* scan, RelativePose* odom_out, bool icp_before_store) {
    if (trajectory_.isEmpty()) {
      scan->key = trajectory_.extend(scan->time_ns, getPoseMeasurement(scan->time_ns));
      setPoseKey(scan->time_ns, scan->key);
      laser_scans_.push_back(std::move(*scan));
      return;
    }
    const Time t_last = trajectory_.getMaxTime();
    RelativePose rel;
    rel.T_a_b = getPoseMeasurement(t_last).inverse() * getPoseMeasurement(scan->time_ns);
    rel.time_a_ns = t_last;
    rel.key_a = getPoseKey(t_last);
    rel.time_b_ns = scan->time_ns;
    scan->key = trajectory_.extend(scan->time_ns, trajectory_.evaluate(t_last) * rel.T_a_b);
    if (!icp_before_store) { setPoseKey(scan->time_ns, scan->key); laser_scans_.push_back(std::move(*scan)); }
    rel.key_b = scan->key;
    rel.track_id_a = rel.track_id_b = laser_track_id_;
    odometry_measurements_.push_back(rel);
    if (params_.use_icp_factors) computeICPTransformations();
    if (icp_before_store) { setPoseKey(scan->time_ns, scan->key); laser_scans_.push_back(std::move(*scan)); }
    if (odom_out) *odom_out = rel;
  }

  void computeICPTransformations() {  // :460-464
    if (laser_scans_.size() > 1u) localScanToSubMap();
  }

  // laser_track.cpp:466-519
  void localScanToSubMap() {
    const size_t n = laser_scans_.size();
    const LaserScan& last_scan = laser_scans_.back();
    const DataPoints& reading = pending_reading_ ? *pending_reading_ : last_scan.scan;   // (processPoseAndLaserScan: the copy may still be under way)
    RelativePose icp;
    icp.time_b_ns = last_scan.time_ns;
    icp.time_a_ns = laser_scans_[n - 2].time_ns;
    // sub-map = scan n-2 plus the (nscan_in_sub_map - 1) scans before it, in the frame of scan n-2
    const SE3 T_w_a = trajectory_.evaluate(icp.time_a_ns);
    // (nscan_in_sub_map = 0: the reference's `nscan_in_sub_map - 1u` wraps, laser_track.cpp:478, and the loop takes EVERY
    // earlier scan; the mirror clamps to one scan instead -- the reference's configurations all set >= 1)
    const size_t extra = std::min(n - 2, size_t(std::max(params_.nscan_in_sub_map, 1) - 1));
    std::vector<size_t> members{n - 2};
    std::vector<TransformationParameters> member_T{identityTransformation()};
    for (size_t i = 0; i < extra; ++i) {
      const LaserScan& prev = laser_scans_[n - 3 - i];
      TransformationParameters T = (T_w_a.inverse() * trajectory_.evaluate(prev.time_ns)).transformationMatrixF();
      correctTransformationMatrix(&T);
      members.push_back(n - 3 - i);
      member_T.push_back(T);
    }
    // initial guess from the (odometry-extended) trajectory
    const SE3 guess = trajectory_.evaluate(icp.time_a_ns).inverse() * trajectory_.evaluate(icp.time_b_ns);
    const TransformationParameters T_init = guess.transformationMatrixF();
    TransformationParameters solution = T_init;
    try {
      if (params_.scans_on_device > 0 && (int)members.size() + 1 <= params_.scans_on_device &&
          !icp_.hasComputeOverride()) {
        // the scans stay in HBM; the sub-map is assembled there (same arithmetic as RigidTransformation::compute)
        const auto t_up = std::chrono::steady_clock::now();
        std::vector<int> slots;
        for (size_t m : members) slots.push_back(deviceSlot(m));
        // the new scan is matched exactly once, right after it arrived: its upload goes out WITH the registration and
        // crosses PCIe while the sub-map -- scans already in HBM -- is assembled and filtered
        const bool reading_resident = slotHolds(n - 1);
        const int reading_slot = reading_resident ? deviceSlot(n - 1) : claimSlot(n - 1);
        stage_times_.upload_ms = msSince(t_up);
        const auto t_icp = std::chrono::steady_clock::now();
        if (reading_resident) {
          solution = icp_.computeClouds(reading_slot, slots, member_T, T_init);
        } else {
          // the slot counts as holding the new scan only once the fused call has stored it there: on return, or on a
          // ConvergenceError (the upload is done whatever the registration's outcome); anything else -- a size mismatch,
          // a failed copy -- leaves the slot unowned, so that the next scan uploads instead of matching a stale cloud
          try {
            solution = icp_.computeCloudsUploading(reading_slot, reading, slots, member_T, T_init);
            confirmSlot(reading_slot, n - 1);
          } catch (const ConvergenceError&) {
            confirmSlot(reading_slot, n - 1);
            throw;
          }
        }
        stage_times_.icp_ms = msSince(t_icp);
#ifdef LSGPU_TEST_SEAMS
        if (icp_.hasComputeObserver()) {   // parity drivers: hand the observer the clouds the device just matched
          DataPoints sub_map = laser_scans_[members[0]].scan;
          for (size_t i = 1; i < members.size(); ++i)
            sub_map.concatenate(RigidTransformation::compute(laser_scans_[members[i]].scan, member_T[i]));
          icp_.notifyObserver(reading, sub_map, T_init, solution);
        }
#endif
      } else {
        DataPoints sub_map = laser_scans_[members[0]].scan;
        for (size_t i = 1; i < members.size(); ++i)
          sub_map.concatenate(RigidTransformation::compute(laser_scans_[members[i]].scan, member_T[i]));
        solution = icp_.compute(reading, sub_map, T_init);
      }
    } catch (const ConvergenceError&) {
      // keep the initial guess (laser_track.cpp:499-502)
    }
    if (params_.save_icp_results) {   // laser_track.cpp:504-513 (incl. the corrected solution going on into the factor)
      DataPoints sub_map = laser_scans_[members[0]].scan;
      for (size_t i = 1; i < members.size(); ++i)
        sub_map.concatenate(RigidTransformation::compute(laser_scans_[members[i]].scan, member_T[i]));
      const std::string dir = params_.save_icp_results_dir;
      saveVTK(reading, dir + "/last_scan.vtk");
      saveVTK(sub_map, dir + "/sub_map.vtk");
      TransformationParameters guess_corrected = T_init;
      correctTransformationMatrix(&guess_corrected);
      saveVTK(RigidTransformation::compute(reading, guess_corrected), dir + "/last_scan_alligned_by_initial_guess.vtk");
      correctTransformationMatrix(&solution);
      saveVTK(RigidTransformation::compute(reading, solution), dir + "/last_scan_alligned_by_solution.vtk");
    }
    icp.T_a_b = SE3::fromTransformationMatrix(solution.data());  // convertTransformationMatrixToSE3
    icp.key_a = getPoseKey(icp.time_a_ns);
    icp.key_b = getPoseKey(icp.time_b_ns);
    icp.track_id_a = icp.track_id_b = laser_track_id_;
    icp_transformations_.push_back(icp);
  }

  // (is scan `index` in its slot already?  /  its slot, reserved for it: the fused call uploads (lsgpu_icp_compute_clouds_upload),
  // confirmSlot() marks the slot as holding the scan once it does)
  bool slotHolds(size_t index) {
    const int slot = (int)(index % (size_t)params_.scans_on_device);
    return slot_generation_ == icp_.generation() && (size_t)slot < slot_owner_.size() && slot_owner_[(size_t)slot] == index &&
           icp_.hasCloud(slot);
  }
  int claimSlot(size_t index) {
    const int slot = (int)(index % (size_t)params_.scans_on_device);
    if (slot_generation_ != icp_.generation()) { slot_owner_.clear(); slot_generation_ = icp_.generation(); }
    if ((int)slot_owner_.size() < params_.scans_on_device) slot_owner_.resize((size_t)params_.scans_on_device, (size_t)-1);
    slot_owner_[(size_t)slot] = (size_t)-1;   // nobody's until confirmSlot()
    (void)index;
    return slot;
  }
  void confirmSlot(int slot, size_t index) {
    if ((size_t)slot < slot_owner_.size()) slot_owner_[(size_t)slot] = index;
    slot_generation_ = icp_.generation();
  }
  // Device slot of scan `index`: scan i lives in slot i % scans_on_device while it is among the most recent
  // ones; anything else (or everything, after the ICP object was reconfigured) is uploaded on demand.
  int deviceSlot(size_t index) {
    const int slot = (int)(index % (size_t)params_.scans_on_device);
    if (slot_generation_ != icp_.generation()) { slot_owner_.clear(); slot_generation_ = icp_.generation(); }
    if ((int)slot_owner_.size() < params_.scans_on_device) slot_owner_.resize((size_t)params_.scans_on_device, (size_t)-1);
    if (slot_owner_[(size_t)slot] != index || !icp_.hasCloud(slot)) {
      icp_.uploadCloud(slot, (pending_reading_ && index + 1 == laser_scans_.size()) ? *pending_reading_ : laser_scans_[index].scan);
      slot_owner_[(size_t)slot] = index;
      slot_generation_ = icp_.generation();  // (uploadCloud may have created the handle)
    }
    return slot;
  }

  LaserTrackParams params_;
  unsigned int laser_track_id_;
  ICP icp_;
  DataPointsFilters input_filters_;
  Trajectory trajectory_;
  std::vector<Pose> pose_measurements_;
  std::vector<RelativePose> odometry_measurements_;
  std::vector<RelativePose> icp_transformations_;
  std::vector<LaserScan> laser_scans_;
  std::map<Time, double> scan_matching_times_;
  std::vector<size_t> slot_owner_;   // which scan each device slot holds
  StageTimes stage_times_;
  const DataPoints* pending_reading_ = nullptr;   // the newest scan's points while their copy into laser_scans_ is under way
  unsigned slot_generation_ = 0;
  mutable std::recursive_mutex mutex_;
};

}  // namespace laser_slam_amd
