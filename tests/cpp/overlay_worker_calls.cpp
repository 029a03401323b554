// overlay_worker_calls.cpp -- PARSE CHECK ONLY (g++ -fsyntax-only, tests/cpp/mock/README.md): every call that
// laser_slam_ros/src/laser_slam_worker.cpp:47-600 makes on laser_slam::LaserTrack / IncrementalEstimator, written against
// the GTSAM-typed overlay under the include names laser_slam_ros uses.  Each call cites the worker line it restates.
// Never linked, never run: it shows that the names and signatures resolve, nothing about behaviour.
#include <laser_slam/common.hpp>
#include <laser_slam/incremental_estimator.hpp>
#include <laser_slam/laser_track.hpp>
#include <laser_slam/parameters.hpp>

void worker_calls(std::shared_ptr<laser_slam::IncrementalEstimator> incremental_estimator, unsigned int worker_id,
                  const laser_slam::Pose& pose_from_tf, const laser_slam::LaserScan& new_scan) {
  using namespace laser_slam;
  std::shared_ptr<LaserTrack> laser_track = incremental_estimator->getLaserTrack(worker_id);                      // :47
  gtsam::NonlinearFactorGraph new_factors;
  gtsam::Values new_values;
  bool is_prior = false;
  laser_track->processPoseAndLaserScan(pose_from_tf, new_scan, &new_factors, &new_values, &is_prior);             // :133, :158
  if (laser_track->getNumScans() > 2u) {                                                                          // :140
    Pose current_pose = laser_track->getCurrentPose();                                                            // :141
    Time previous_pose_time = current_pose.time_ns;
    if (previous_pose_time >= laser_track->getMinTime() && previous_pose_time <= laser_track->getMaxTime()) {     // :146-147
      SE3 previous_pose = laser_track->evaluate(previous_pose_time);                                              // :148
      Pose new_pose;
      new_pose.T_w = pose_from_tf.T_w * previous_pose.inverse() * current_pose.T_w;                               // :149-151
      (void)new_pose;
    }
  }
  gtsam::Values result;
  if (is_prior) result = incremental_estimator->registerPrior(new_factors, new_values, worker_id);                // :167
  else result = incremental_estimator->estimate(new_factors, new_values, new_scan.time_ns);                       // :169
  laser_track->updateFromGTSAMValues(result);                                                                     // :173
  Pose current_pose = laser_track->getCurrentPose();                                                              // :176
  (void)current_pose;
  DataPoints new_fixed_cloud;
  laser_track->getLocalCloudInWorldFrame(laser_track->getMaxTime(), &new_fixed_cloud);                            // :197
  laser_slam::PointMatcher::TransformationParameters transformation_matrix;
  correctTransformationMatrix(&transformation_matrix);                                                            // :204
  Clock clock;                                                                                                    // :432
  clock.takeTime();
  (void)clock.getRealTime();

  // get_laser_track service, :260-317
  std::vector<std::shared_ptr<LaserTrack> > laser_tracks = incremental_estimator->getAllLaserTracks();            // :264
  for (const auto& track : laser_tracks) {
    Trajectory trajectory;
    track->getTrajectory(&trajectory);                                                                            // :271
    for (const auto& scan : track->getLaserScans()) {                                                             // :272
      const DataPoints& cloud = scan.scan;                                                                        // :276
      SE3 pose = trajectory.at(scan.time_ns);                                                                     // :279
      (void)cloud; (void)pose;
    }
  }
  Trajectory out_trajectory;
  laser_track->getTrajectory(&out_trajectory);                                                                    // :374, :515, :526, :545, :553, :571
  laser_track->getOdometryTrajectory(&out_trajectory);                                                            // :519
  (void)laser_track->getMaxTime();                                                                                // :600

  // the loop-closure entry point laser_slam_ros' users call (segmatch / laser_mapper): incremental_estimator.hpp:28
  RelativePose loop_closure;
  incremental_estimator->processLoopClosure(loop_closure);
  (void)incremental_estimator->getCurrentPose(worker_id);
  std::vector<unsigned int> affected{0u, 1u};
  (void)incremental_estimator->estimateAndRemove(new_factors, new_factors, new_values, affected, new_scan.time_ns);

  // the remaining public surface of laser_track.hpp:20-144
  laser_track->processPose(pose_from_tf);
  laser_track->processLaserScan(new_scan);
  std::vector<Covariance> covariances;
  laser_track->getCovariances(&covariances);
  (void)laser_track->getPreviousPose();
  std::vector<Time> times;
  laser_track->getLaserScansTimes(&times);
  laser_track->appendPriorFactors(0, &new_factors);
  laser_track->appendOdometryFactors(0, 1, gtsam::noiseModel::Base::shared_ptr(), &new_factors);
  laser_track->appendICPFactors(0, 1, gtsam::noiseModel::Base::shared_ptr(), &new_factors);
  laser_track->appendLoopClosureFactors(0, 1, gtsam::noiseModel::Base::shared_ptr(), &new_factors);
  laser_track->initializeGTSAMValues(new_factors.keys(), &new_values);
  laser_track->updateCovariancesFromGTSAMValues(new_factors, new_values);
  laser_track->printTrajectory();
  (void)laser_track->findNearestPose(0);
  DataPoints sub_map;
  laser_track->buildSubMapAroundTime(0, 3u, &sub_map);
  (void)laser_track->getValueExpression(0);
  std::map<Time, double> scan_matching_times;
  laser_track->getScanMatchingTimes(&scan_matching_times);
  laser_track->saveTrajectory("trajectory.csv");
  laser_track->getLastPointCloud(&sub_map);
  laser_track->getPointCloudOfTimeInterval(std::make_pair<Time, Time>(0, 1), &sub_map);
  (void)convertTransformationMatrixToSE3(transformation_matrix);

  // construction as laser_slam_ros' laser_mapper does it
  EstimatorParams params;
  params.loop_closure_noise_model[0] = 0.01;
  params.laser_track_params.odometry_noise_model[5] = 0.1;
  std::shared_ptr<IncrementalEstimator> estimator(new IncrementalEstimator(params, 2u));
}
