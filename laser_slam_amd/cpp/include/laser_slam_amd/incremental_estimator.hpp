// incremental_estimator.hpp -- host-side mirror of laser_slam::IncrementalEstimator
// (laser_slam/include/laser_slam/incremental_estimator.hpp:18-80,
//  laser_slam/src/incremental_estimator.cpp): owns the laser tracks and the pose graph, registers
// priors, runs the per-scan estimate, and processes loop closures -- the SECOND call site of
// icp_.compute (sub-map vs sub-map, incremental_estimator.cpp:89-115), which runs on the device.
//
// gtsam::ISAM2 is replaced by PoseGraph (pose_graph.hpp): each call that performs k iSAM2 updates in
// the reference performs k Gauss-Newton steps here.  Same method names, argument meaning and
// bookkeeping (prior removal when two robots' trajectories get linked, first-association noise,
// Cauchy(1) on loop closures).  glog CHECKs become std::logic_error.
#pragma once
#include <algorithm>
#include <fstream>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "laser_slam_amd/laser_track.hpp"
#include "laser_slam_amd/pose_graph.hpp"

namespace laser_slam_amd {

struct EstimatorParams {  // laser_slam/include/laser_slam/parameters.hpp:25-34
  std::array<double, 6> loop_closure_noise_model{};
  bool add_m_estimator_on_loop_closures = false;
  bool do_icp_step_on_loop_closures = false;
  int loop_closures_sub_maps_radius = 3;
  LaserTrackParams laser_track_params;
};

// Which prior goes when two robots' graphs first link (incremental_estimator.cpp:176-241, 276-283): every worker
// registers a prior and starts as its own group; the factor index of that prior is remembered for every worker but 0.
// The first loop closure between two groups merges them into the group that holds worker 0 (the second worker's group if
// neither does) and hands back the ONE prior index of the absorbed group, which the caller removes from its graph while
// it adds the first-association factor instead of the loop-closure factor.  Graph-agnostic (indices are whatever the
// graph returned), so the GTSAM-typed overlay (integration/gtsam/) runs the same, tested, bookkeeping over gtsam::ISAM2.
class WorkerLinks {
 public:
  void registerPrior(unsigned int worker_id, size_t factor_index) {
    if (worker_id > 0u) factor_indices_to_remove_.emplace(worker_id, factor_index);   // (insert semantics: the first index stays)
    linked_workers_.push_back({worker_id});
  }
  // indices to remove for a loop closure between the two workers: none (same worker / already linked) or exactly one
  std::vector<size_t> link(const std::vector<unsigned int>& affected_worker_ids) {
    if (affected_worker_ids.size() != 2u) throw std::logic_error("two affected workers expected");
    std::vector<size_t> to_remove;
    const unsigned int first = affected_worker_ids[0], second = affected_worker_ids[1];
    if (first == second) return to_remove;
    int group_first = -1, group_second = -1;
    for (size_t g = 0; g < linked_workers_.size(); ++g) {
      const auto& grp = linked_workers_[g];
      if (std::find(grp.begin(), grp.end(), first) != grp.end()) group_first = (int)g;
      if (std::find(grp.begin(), grp.end(), second) != grp.end()) group_second = (int)g;
    }
    if (group_first < 0 || group_second < 0) throw std::logic_error("worker without a registered prior");
    if (group_first == group_second) return to_remove;
    const auto& gf = linked_workers_[(size_t)group_first];
    const bool keep_first = std::find(gf.begin(), gf.end(), 0u) != gf.end();
    const int keep = keep_first ? group_first : group_second, drop = keep_first ? group_second : group_first;
    for (unsigned int worker : linked_workers_[(size_t)drop]) {
      auto it = factor_indices_to_remove_.find(worker);
      if (it != factor_indices_to_remove_.end()) {
        to_remove.push_back(it->second);
        factor_indices_to_remove_.erase(it);
      }
      linked_workers_[(size_t)keep].push_back(worker);
    }
    if (to_remove.size() != 1u) throw std::logic_error("exactly one prior must be removed");
    linked_workers_.erase(linked_workers_.begin() + drop);
    return to_remove;
  }
  const std::vector<std::vector<unsigned int>>& groups() const { return linked_workers_; }

 private:
  std::unordered_map<unsigned int, size_t> factor_indices_to_remove_;
  std::vector<std::vector<unsigned int>> linked_workers_;
};

class IncrementalEstimator {
 public:
  explicit IncrementalEstimator(const EstimatorParams& parameters, unsigned int n_laser_slam_workers = 1u)
      : params_(parameters), n_laser_slam_workers_(n_laser_slam_workers), icp_(parameters.laser_track_params.device) {
    for (unsigned int i = 0; i < n_laser_slam_workers_; ++i)
      laser_tracks_.push_back(std::make_shared<LaserTrack>(parameters.laser_track_params, i));
    // incremental_estimator.cpp:40-47
    first_association_sigmas_ = {0.05, 0.05, 0.05, 0.015, 0.015, 0.015};
    // incremental_estimator.cpp:49-59: the loop-closure ICP uses the lidar-odometry configuration
    std::ifstream ifs(params_.laser_track_params.icp_configuration_file.c_str());
    if (!params_.laser_track_params.icp_configuration_file.empty() && ifs.good()) icp_.loadFromYaml(ifs);
    else icp_.setDefault();
  }

  // incremental_estimator.cpp:62-149
  void processLoopClosure(const RelativePose& loop_closure) {
    std::lock_guard<std::recursive_mutex> lock(full_class_mutex_);
    LaserTrack& track_a = *laser_tracks_.at(loop_closure.track_id_a);
    LaserTrack& track_b = *laser_tracks_.at(loop_closure.track_id_b);
    if (loop_closure.track_id_a == loop_closure.track_id_b && !(loop_closure.time_a_ns < loop_closure.time_b_ns))
      throw std::logic_error("Loop closure has invalid time.");
    if (loop_closure.time_a_ns < track_a.getMinTime() || loop_closure.time_a_ns > track_a.getMaxTime() ||
        loop_closure.time_b_ns < track_b.getMinTime() || loop_closure.time_b_ns > track_b.getMaxTime())
      throw std::logic_error("Loop closure has invalid time.");

    RelativePose updated = loop_closure;
    // w_T_a_b aligns the source cloud with the target cloud in the world frame -> frame of a (:83-89)
    const SE3 T_w_a = track_a.evaluate(loop_closure.time_a_ns);
    const SE3 T_w_b = track_b.evaluate(loop_closure.time_b_ns);
    updated.T_a_b = T_w_a.inverse() * loop_closure.T_a_b * T_w_b;

    if (params_.do_icp_step_on_loop_closures) {  // :92-115
      TransformationParameters initial_guess = updated.T_a_b.transformationMatrixF();
      DataPoints sub_map_a, sub_map_b;
      track_a.buildSubMapAroundTime(loop_closure.time_a_ns, params_.loop_closures_sub_maps_radius, &sub_map_a);
      track_b.buildSubMapAroundTime(loop_closure.time_b_ns, params_.loop_closures_sub_maps_radius, &sub_map_b);
      // (the guess goes in as it is: unlike laser_track.cpp:489-491 this call site does not correct it, and an exception of
      // icp_.compute -- ConvergenceError, or TransformationError for a guess that is not rigid -- propagates: no try block
      // around incremental_estimator.cpp:108)
      const TransformationParameters icp_solution = icp_.compute(sub_map_b, sub_map_a, initial_guess);
      updated.T_a_b = SE3::fromTransformationMatrix(icp_solution.data());
      last_loop_closure_icp_stats_ = icp_.lastStats();
    }

    Factor f;
    f.type = Factor::LOOP_CLOSURE;
    f.key_a = track_a.getValueKey(updated.time_a_ns);
    f.key_b = track_b.getValueKey(updated.time_b_ns);
    f.measurement = updated.T_a_b;
    f.sigmas = params_.loop_closure_noise_model;
    f.cauchy = params_.add_m_estimator_on_loop_closures;
    Factor first_association = f;  // :129-132: plain diagonal noise for the factor that links two robots
    first_association.sigmas = first_association_sigmas_;
    first_association.cauchy = false;

    const Values result = estimateAndRemove({f}, {first_association}, Values(),
                                            {loop_closure.track_id_a, loop_closure.track_id_b}, updated.time_b_ns);
    for (auto& track : laser_tracks_) track->updateFromValues(result);
    last_loop_closure_ = updated;
  }

  Pose getCurrentPose(unsigned int laser_track_id = 0u) const {
    std::lock_guard<std::recursive_mutex> lock(full_class_mutex_);
    return laser_tracks_.at(laser_track_id)->getCurrentPose();
  }

  std::shared_ptr<LaserTrack> getLaserTrack(unsigned int laser_track_id) {
    std::lock_guard<std::recursive_mutex> lock(full_class_mutex_);
    return laser_tracks_.at(laser_track_id);
  }
  std::vector<std::shared_ptr<LaserTrack>> getAllLaserTracks() {
    std::lock_guard<std::recursive_mutex> lock(full_class_mutex_);
    return laser_tracks_;
  }

  // incremental_estimator.cpp:151-163: update(new) + 2 x update()
  Values estimate(const FactorList& new_factors, const Values& new_values, Time /*timestamp_ns*/ = 0) {
    std::lock_guard<std::recursive_mutex> lock(full_class_mutex_);
    graph_.insert(new_values);
    for (const Factor& f : new_factors) graph_.addFactor(f);
    graph_.optimize(3);
    return graph_.values();
  }

  // incremental_estimator.cpp:165-266
  Values estimateAndRemove(const FactorList& new_factors, const FactorList& new_associations_factors,
                           const Values& new_values, const std::vector<unsigned int>& affected_worker_ids,
                           Time /*timestamp_ns*/ = 0) {
    std::lock_guard<std::recursive_mutex> lock(full_class_mutex_);
    const std::vector<size_t> factor_indices_to_remove = links_.link(affected_worker_ids);
    graph_.insert(new_values);
    for (size_t idx : factor_indices_to_remove) graph_.removeFactor(idx);
    for (const Factor& f : (factor_indices_to_remove.empty() ? new_factors : new_associations_factors))
      graph_.addFactor(f);
    graph_.optimize(3);
    return graph_.values();
  }

  // incremental_estimator.cpp:268-291
  Values registerPrior(const FactorList& new_factors, const Values& new_values, const unsigned int worker_id) {
    std::lock_guard<std::recursive_mutex> lock(full_class_mutex_);
    if (new_factors.size() != 1u) throw std::logic_error("registerPrior expects exactly one factor");
    graph_.insert(new_values);
    const size_t index = graph_.addFactor(new_factors[0]);
    links_.registerPrior(worker_id, index);
    graph_.optimize(3);
    return graph_.values();
  }

  ICP& loopClosureIcp() { return icp_; }  // configuration access (seed, test seam)
  const PoseGraph& graph() const { return graph_; }
  const std::vector<std::vector<unsigned int>>& linkedWorkers() const { return links_.groups(); }
  const RelativePose& lastLoopClosure() const { return last_loop_closure_; }
  const lsgpu_icp_stats& lastLoopClosureIcpStats() const { return last_loop_closure_icp_stats_; }

 private:
  EstimatorParams params_;
  unsigned int n_laser_slam_workers_;
  mutable std::recursive_mutex full_class_mutex_;
  std::vector<std::shared_ptr<LaserTrack>> laser_tracks_;
  PoseGraph graph_;
  ICP icp_;
  std::array<double, 6> first_association_sigmas_{};
  WorkerLinks links_;
  RelativePose last_loop_closure_;
  lsgpu_icp_stats last_loop_closure_icp_stats_{};
};

}  // namespace laser_slam_amd
