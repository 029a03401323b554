// host_checks.cpp -- CPU-only checks of the C++ mirror (no GPU needed): YAML subset parser, SE3
// algebra, RigidTransformation on the host, sub-map bookkeeping errors, and that a GPU-less box
// produces a loud DeviceError from compute() instead of a fallback.
#include <cmath>
#include <cstdio>
#include <sstream>

#include "laser_slam_amd/incremental_estimator.hpp"
#include "laser_slam_amd/laser_track.hpp"

using namespace laser_slam_amd;
static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

int main(int argc, char** argv) {
  const std::string golden = std::getenv("LSGPU_GOLDEN_DIR") ? std::getenv("LSGPU_GOLDEN_DIR") : "tests/golden";
  const std::string no_filters = golden + "/input_filters_none.yaml";

  // --- YAML
  {
    ICP icp;
    std::stringstream y(
        "readingDataPointsFilters:\n  - RandomSamplingDataPointsFilter:\n      prob: 0.5\n\n"
        "referenceDataPointsFilters:\n  - SamplingSurfaceNormalDataPointsFilter:\n      knn: 10\n"
        "matcher:\n  KDTreeMatcher:\n    knn: 1\n    epsilon: 0 \n"
        "outlierFilters:\n  - TrimmedDistOutlierFilter:\n      ratio: 0.75\n"
        "errorMinimizer:\n  PointToPlaneErrorMinimizer\n"
        "transformationCheckers:\n  - CounterTransformationChecker:\n      maxIterationCount: 40\n"
        "  - DifferentialTransformationChecker:\n      minDiffRotErr: 0.001\n      minDiffTransErr: 0.01\n      smoothLength: 4   \n"
        "#inspector:\n#  NullInspector\ninspector:\n VTKFileInspector:\n     baseFileName: x\n     dumpPerfOnExit: 0\nlogger:\n  NullLogger\n");
    icp.loadFromYaml(y);
    CHECK(std::fabs(icp.config().trim_ratio - 0.75f) < 1e-7f);
    CHECK(icp.config().max_iterations == 40 && icp.config().smooth_length == 4);
    CHECK(std::fabs(icp.config().min_diff_trans - 0.01f) < 1e-7f);
    CHECK(std::fabs(icp.readingSamplingProb() - 0.5f) < 1e-7f && icp.surfaceNormalKnn() == 10);
    std::stringstream inl("referenceDataPointsFilters:\n  - SamplingSurfaceNormalDataPointsFilter\nmatcher: KDTreeMatcher\n"
                          "outlierFilters:\n  - TrimmedDistOutlierFilter: {ratio: 0.9}\nerrorMinimizer: PointToPlaneErrorMinimizer\n"
                          "transformationCheckers:\n  - CounterTransformationChecker\n  - DifferentialTransformationChecker\n");
    icp.loadFromYaml(inl);
    CHECK(std::fabs(icp.config().trim_ratio - 0.9f) < 1e-7f && icp.config().smooth_length == 3);   // module defaults
    CHECK(icp.surfaceNormalKnn() == 7 && icp.config().max_iterations == 40);
    icp.setDefault();
    CHECK(std::fabs(icp.config().trim_ratio - 0.85f) < 1e-7f && icp.surfaceNormalKnn() == 7);
    bool threw = false;
    try { std::stringstream bad("outlierFilters:\n  - MaxDistOutlierFilter:\n      maxDist: 1\n"); icp.loadFromYaml(bad); }
    catch (const ConfigError&) { threw = true; }
    CHECK(threw);
  }
  // --- SE3
  {
    SE3 a({0.9, 0.1, -0.2, 0.3}, {1, 2, 3}), b({0.7, -0.3, 0.2, 0.1}, {-1, 0.5, 2});
    SE3 id = a * a.inverse();
    CHECK(std::fabs(id.quaternion()[0]) > 1 - 1e-12 && std::fabs(id.position()[0]) < 1e-12);
    SE3 ab = a * b;
    auto T = ab.transformationMatrixF();
    SE3 back = SE3::fromTransformationMatrix(T.data());
    double dot = 0;
    for (int i = 0; i < 4; ++i) dot += back.quaternion()[i] * ab.quaternion()[i];
    CHECK(std::fabs(std::fabs(dot) - 1) < 1e-6);
    for (int i = 0; i < 3; ++i) CHECK(std::fabs(back.position()[i] - ab.position()[i]) < 1e-6);
    SE3 mid = SE3::interpolate(a, b, 0.0);
    CHECK(std::fabs(mid.position()[1] - 2) < 1e-12);
  }
  // --- RigidTransformation + correctTransformationMatrix
  {
    SE3 a({0.9, 0.1, -0.2, 0.3}, {1, 2, 3});
    TransformationParameters T = a.transformationMatrixF();
    CHECK(RigidTransformation::checkParameters(T));
    DataPoints d;
    d.features = {1, 0, 0, 1, 0, 2, 0, 1};
    d.normals = {1, 0, 0, 0, 1, 0};
    DataPoints o = RigidTransformation::compute(d, T);
    double R[9];
    a.rotationMatrix(R);
    CHECK(std::fabs(o.features[0] - (R[0] + 1)) < 1e-5 && std::fabs(o.features[5] - (2 * R[4] + 2)) < 1e-5);
    CHECK(std::fabs(o.normals[3] - R[1]) < 1e-6 && o.features[3] == 1.f);
    TransformationParameters bad = T;
    bad[0] *= 1.05f; bad[1] *= 1.05f; bad[2] *= 1.05f;
    CHECK(!RigidTransformation::checkParameters(bad));
    correctTransformationMatrix(&bad);
    CHECK(RigidTransformation::checkParameters(bad));
    d.concatenate(o);
    CHECK(d.getNbPoints() == 4 && d.normals.size() == 12);
  }
  // --- LaserTrack bookkeeping without touching the GPU (use_icp_factors = false)
  {
    LaserTrackParams p;
    p.use_icp_factors = false;
    {  // laser_track.cpp:24-30: an unreadable input-filter file is fatal
      bool fatal = false;
      try { LaserTrack t(p, 0u); } catch (const ConfigError&) { fatal = true; }
      CHECK(fatal);
    }
    p.icp_input_filters_file = no_filters;
    LaserTrack track(p, 2u);
    CHECK(track.inputFilters().empty());
    bool threw = false;
    LaserScan s; s.time_ns = 5; s.scan.features = {0, 0, 0, 1};
    try { track.processLaserScan(s); } catch (const std::logic_error&) { threw = true; }  // no pose registered
    CHECK(threw);
    for (int i = 0; i < 3; ++i) {
      Pose pose; pose.time_ns = 100 * i; pose.T_w = SE3({1, 0, 0, 0}, {0.8 * i, 0, 0});
      LaserScan sc; sc.time_ns = 100 * i; sc.scan.features = {float(i), 0, 0, 1};
      FactorList f; Values v; bool prior = false;
      track.processPoseAndLaserScan(pose, sc, &f, &v, &prior);
      CHECK(prior == (i == 0));
      CHECK(f.size() == 1 && v.size() == 1);
      CHECK(f[0].type == (i == 0 ? Factor::PRIOR : Factor::ODOMETRY));
      if (i > 0) CHECK(std::fabs(f[0].measurement.position()[0] - 0.8) < 1e-12);
    }
    CHECK(track.getNumScans() == 3 && track.getMaxTime() == 200);
    CHECK(std::fabs(track.evaluate(150).position()[0] - 1.2) < 1e-9);  // interpolating curve
    DataPoints sub;
    track.buildSubMapAroundTime(100, 1, &sub);
    CHECK(sub.getNbPoints() == 3);
    DataPoints w;
    track.getLocalCloudInWorldFrame(200, &w);
    CHECK(std::fabs(w.features[0] - 3.6f) < 1e-5);
    TrajectoryMap tm;
    track.getTrajectory(&tm);
    CHECK(tm.size() == 3);
    // getOdometryTrajectory (laser_track.cpp:310-316; caller laser_slam_worker.cpp:519): the pose MEASUREMENTS by time
    // stamp -- untouched by a graph update of the trajectory, and a pose registered without a scan is part of it
    Values moved; moved[track.getValueKey(100)] = SE3({1, 0, 0, 0}, {5.0, 0, 0});
    track.updateFromValues(moved);
    Pose extra; extra.time_ns = 250; extra.T_w = SE3({1, 0, 0, 0}, {2.0, 0, 0});
    track.processPose(extra);
    TrajectoryMap odo;
    track.getOdometryTrajectory(&odo);
    CHECK(odo.size() == 4 && std::fabs(odo.at(100).position()[0] - 0.8) < 1e-12 && std::fabs(odo.at(250).position()[0] - 2.0) < 1e-12);
    track.getTrajectory(&tm);
    CHECK(tm.size() == 3 && std::fabs(tm.at(100).position()[0] - 5.0) < 1e-12);
    // getPreviousPose (:302-312), getLaserScansTimes (:328-334), findNearestPose (:557-572)
    CHECK(track.getPreviousPose().time_ns == 100 && std::fabs(track.getPreviousPose().T_w.position()[0] - 5.0) < 1e-12);
    std::vector<Time> times;
    track.getLaserScansTimes(&times);
    CHECK(times.size() == 3 && times[0] == 0 && times[2] == 200);
    CHECK(track.findNearestPose(200).time_ns == 200 && std::fabs(track.findNearestPose(200).T_w.position()[0] - 1.6) < 1e-12);
    bool late = false;
    try { track.findNearestPose(251); } catch (const std::logic_error&) { late = true; }   // later than the latest pose
    CHECK(late);
  }
  // --- WorkerLinks: which prior goes when two robots' graphs first link (incremental_estimator.cpp:176-241, 276-283)
  {
    WorkerLinks links;
    links.registerPrior(0u, 10); links.registerPrior(1u, 11); links.registerPrior(2u, 12);
    CHECK(links.groups().size() == 3);
    CHECK(links.link({1u, 1u}).empty());                           // same worker: nothing to remove
    std::vector<size_t> r = links.link({2u, 1u});                  // neither group holds worker 0: the SECOND worker's group stays
    CHECK(r.size() == 1 && r[0] == 12 && links.groups().size() == 2);
    CHECK(links.link({1u, 2u}).empty());                           // already linked
    r = links.link({1u, 0u});                                      // worker 0's group is kept, {1, 2} loses its remaining prior
    CHECK(r.size() == 1 && r[0] == 11 && links.groups().size() == 1 && links.groups()[0].size() == 3);
    bool threw2 = false;
    try { links.link({0u, 7u}); } catch (const std::logic_error&) { threw2 = true; }   // a worker that never registered
    CHECK(threw2);
  }
  // --- SE3 chart: retract / localCoordinates are inverse of each other
  {
    SE3 a({0.9, 0.1, -0.2, 0.3}, {1, 2, 3});
    const double d[6] = {0.3, -0.2, 0.1, 0.4, -0.5, 0.6};
    double back[6];
    a.localCoordinates(a.retract(d), back);
    for (int i = 0; i < 6; ++i) CHECK(std::fabs(back[i] - d[i]) < 1e-12);
    const double z[6] = {0, 0, 0, 0, 0, 0};
    a.localCoordinates(a.retract(z), back);
    for (int i = 0; i < 6; ++i) CHECK(std::fabs(back[i]) < 1e-15);
  }
  // --- pose graph: a drifting square loop closed by one loop-closure factor
  {
    auto yaw = [](double a) { return std::array<double, 4>{std::cos(a / 2), 0, 0, std::sin(a / 2)}; };
    const int per_side = 6, n = 4 * per_side;
    std::vector<SE3> truth;
    SE3 cur;  // identity
    for (int i = 0; i < n; ++i) {
      truth.push_back(cur);
      const bool corner = (i + 1) % per_side == 0;
      cur = cur * SE3(yaw(corner ? M_PI / 2 : 0.0), {1.0, 0, 0});
    }
    // odometry: every step is off by 2 cm along x and 0.4 deg in yaw (a consistent drift)
    const SE3 bias(yaw(0.4 * M_PI / 180), {0.02, 0.0, 0.0});
    PoseGraph g;
    Values init;
    init[0] = truth[0];
    g.insert(init);
    Factor prior;
    prior.type = Factor::PRIOR; prior.key_b = 0; prior.measurement = truth[0]; prior.sigmas.fill(1e-7);
    g.addFactor(prior);
    SE3 dead = truth[0];
    for (int i = 1; i < n; ++i) {
      const SE3 odo = truth[i - 1].inverse() * truth[i] * bias;
      dead = dead * odo;
      Values v; v[(Key)i] = dead; g.insert(v);
      Factor f;
      f.type = Factor::ODOMETRY; f.key_a = (Key)i - 1; f.key_b = (Key)i; f.measurement = odo;
      f.sigmas = {0.05, 0.05, 0.05, 0.01, 0.01, 0.01};
      g.addFactor(f);
    }
    g.optimize(3);
    auto endErr = [&](const PoseGraph& gr) {
      double d[6]; truth[n - 1].localCoordinates(gr.values().at((Key)n - 1), d);
      return std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    };
    const double before = endErr(g);
    CHECK(before > 0.5);             // dead reckoning drifted
    CHECK(g.error() < 1e-12);        // ...but the chain alone is perfectly consistent
    Factor lc;
    lc.type = Factor::LOOP_CLOSURE; lc.key_a = 0; lc.key_b = (Key)n - 1;
    lc.measurement = truth[0].inverse() * truth[n - 1];
    lc.sigmas = {0.005, 0.005, 0.005, 0.001, 0.001, 0.001};
    const size_t lc_index = g.addFactor(lc);
    const double last = g.optimize(10);
    CHECK(last < 1e-9);              // Gauss-Newton converged
    CHECK(endErr(g) < 0.02 && endErr(g) < before / 25);
    // the closed loop spreads the correction: mid-loop error shrinks as well
    double dm[6]; truth[n / 2].localCoordinates(g.values().at((Key)n / 2), dm);
    CHECK(std::sqrt(dm[0] * dm[0] + dm[1] * dm[1]) < 0.6 * before);
    // a wrong loop closure with the Cauchy m-estimator barely moves the solution; without it, it does
    PoseGraph robust = g, plain = g;
    Factor bad = lc;
    bad.measurement = lc.measurement * SE3(yaw(0.3), {3.0, -2.0, 0.0});
    bad.cauchy = true;
    robust.addFactor(bad); robust.optimize(10);
    bad.cauchy = false;
    plain.addFactor(bad); plain.optimize(10);
    CHECK(endErr(robust) < 0.05);
    CHECK(endErr(plain) > 0.5);
    // removing a factor restores the previous optimum
    g.removeFactor(lc_index);
    g.optimize(10);
    CHECK(g.error() < 1e-10 && g.numFactors() == (size_t)n);
    bool threw = false;
    try { g.removeFactor(lc_index); } catch (const std::out_of_range&) { threw = true; }
    CHECK(threw);
  }
  // --- the active-set steps of the pose graph (round 6) against whole-graph steps: a 300-pose chain built pose by pose the
  // way IncrementalEstimator::estimate does (two disagreeing between-factors per new pose, three steps per pose), a loop
  // closure two thirds in, the chain carried on behind it.  Same estimates to 1e-8 m / rad at every pose; between loop
  // closures only a handful of poses are touched per update.
  {
    auto yaw = [](double a) { return std::array<double, 4>{std::cos(a / 2), 0, 0, std::sin(a / 2)}; };
    const int n = 300, lc_at = 200;
    PoseGraph inc, whole;
    whole.setWholeGraphSteps(true);
    std::vector<SE3> truth;
    SE3 cur;
    for (int i = 0; i < n; ++i) { truth.push_back(cur); cur = cur * SE3(yaw(2.0 * M_PI / 180), {0.8, 0.02 * std::sin(0.1 * i), 0.0}); }
    size_t most_active_between = 0, active_after_lc = 0;
    SE3 dead = truth[0];
    for (int i = 0; i < n; ++i) {
      for (PoseGraph* g : {&inc, &whole}) {
        if (i == 0) {
          Values v; v[0] = truth[0]; g->insert(v);
          Factor prior; prior.type = Factor::PRIOR; prior.key_b = 0; prior.measurement = truth[0]; prior.sigmas.fill(1e-4);
          g->addFactor(prior);
        } else {
          const SE3 rel = truth[i - 1].inverse() * truth[i];
          const SE3 odo = rel * SE3(yaw(0.3 * M_PI / 180), {0.01, -0.004, 0.002});             // drifting odometry
          const SE3 icp = rel * SE3(yaw(0.01 * std::sin(1.3 * i) * M_PI / 180), {0.001 * std::cos(0.7 * i), 0.0, 0.0});
          if (g == &inc) dead = g->values().at((Key)i - 1) * odo;
          Values v; v[(Key)i] = dead; g->insert(v);
          Factor fo; fo.type = Factor::ODOMETRY; fo.key_a = (Key)i - 1; fo.key_b = (Key)i; fo.measurement = odo;
          fo.sigmas = {0.05, 0.05, 0.05, 0.01, 0.01, 0.01};
          Factor fi = fo; fi.type = Factor::ICP; fi.measurement = icp; fi.sigmas = {0.01, 0.01, 0.01, 0.002, 0.002, 0.002}; fi.cauchy = true;
          g->addFactor(fo); g->addFactor(fi);
        }
        if (i == lc_at) {
          Factor lc; lc.type = Factor::LOOP_CLOSURE; lc.key_a = 5; lc.key_b = (Key)i;
          lc.measurement = truth[5].inverse() * truth[i]; lc.sigmas = {0.005, 0.005, 0.005, 0.001, 0.001, 0.001}; lc.cauchy = true;
          g->addFactor(lc);
        }
        g->optimize(3);
      }
      if (i == lc_at) active_after_lc = inc.numActive();
      else if (i > 10 && (i < lc_at || i > lc_at + 10)) most_active_between = std::max(most_active_between, inc.numActive());
    }
    double worst = 0;
    for (int i = 0; i < n; ++i) {
      double d[6]; whole.values().at((Key)i).localCoordinates(inc.values().at((Key)i), d);
      for (double v : d) worst = std::max(worst, std::fabs(v));
    }
    CHECK(worst < 1e-8);
    CHECK(most_active_between <= 12);        // (a handful: the new pose and the ones its factors still nudge)
    CHECK(active_after_lc > 100);            // the loop closure moved the whole loop
  }
  // --- IncrementalEstimator bookkeeping (no ICP: use_icp_factors off, loop closure without ICP step):
  // two robots, priors 100 m apart (force_priors), linked by a loop closure -> robot 1's prior is removed,
  // the first-association noise model is used, and both trajectories end up in one frame
  {
    EstimatorParams ep;
    ep.laser_track_params.icp_input_filters_file = no_filters;
    ep.laser_track_params.use_icp_factors = false;
    ep.laser_track_params.force_priors = true;
    ep.laser_track_params.odometry_noise_model = {0.05, 0.05, 0.05, 0.01, 0.01, 0.01};
    ep.loop_closure_noise_model = {0.005, 0.005, 0.005, 0.001, 0.001, 0.001};
    ep.add_m_estimator_on_loop_closures = true;
    IncrementalEstimator est(ep, 2u);
    DataPoints dummy; dummy.features.assign(4 * 8, 1.f);
    auto drive = [&](unsigned int id, double y0) {
      auto track = est.getLaserTrack(id);
      for (int i = 0; i < 5; ++i) {
        Pose pose; pose.time_ns = 1000 * (i + 1); pose.T_w = SE3({1, 0, 0, 0}, {1.0 * i, y0, 0.0});
        LaserScan scan; scan.time_ns = pose.time_ns; scan.scan = dummy;
        FactorList f; Values v; bool is_prior = false;
        track->processPoseAndLaserScan(pose, scan, &f, &v, &is_prior);
        const Values result = is_prior ? est.registerPrior(f, v, id) : est.estimate(f, v, pose.time_ns);
        for (auto& t : est.getAllLaserTracks()) t->updateFromValues(result);
      }
    };
    drive(0, 0.0);
    drive(1, 7.0);   // robot 1 really drives 7 m beside robot 0, but its forced prior puts it at y = 100
    CHECK(std::fabs(est.getCurrentPose(1).T_w.position()[1] - 100.0) < 1e-6);
    CHECK(est.linkedWorkers().size() == 2 && est.graph().numFactors() == 2 + 2 * 4);
    RelativePose lc;  // robot 1's pose 2 seen from robot 0's pose 2: 7 m to the left; given in the WORLD frame
    lc.track_id_a = 0; lc.track_id_b = 1; lc.time_a_ns = 3000; lc.time_b_ns = 3000;
    const SE3 T_w_a = est.getLaserTrack(0)->evaluate(3000), T_w_b = est.getLaserTrack(1)->evaluate(3000);
    const SE3 a_T_a_b({1, 0, 0, 0}, {0.0, 7.0, 0.0});
    lc.T_a_b = T_w_a * a_T_a_b * T_w_b.inverse();   // processLoopClosure converts it back (:83-89)
    est.processLoopClosure(lc);
    CHECK(est.linkedWorkers().size() == 1 && est.linkedWorkers()[0].size() == 2);
    CHECK(est.graph().numFactors() == 2 + 2 * 4);   // one prior removed, one association factor added
    CHECK(std::fabs(est.getCurrentPose(1).T_w.position()[1] - 7.0) < 1e-3);
    CHECK(std::fabs(est.getCurrentPose(1).T_w.position()[0] - 4.0) < 1e-3);
    CHECK(std::fabs(est.getCurrentPose(0).T_w.position()[1]) < 1e-6);
    CHECK(std::fabs(est.lastLoopClosure().T_a_b.position()[1] - 7.0) < 1e-9);
    // a second closure between the (now linked) robots adds the regular loop-closure factor, removes nothing
    lc.time_a_ns = 5000; lc.time_b_ns = 5000;
    lc.T_a_b = est.getLaserTrack(0)->evaluate(5000) * a_T_a_b * est.getLaserTrack(1)->evaluate(5000).inverse();
    est.processLoopClosure(lc);
    CHECK(est.graph().numFactors() == 3 + 2 * 4);
    bool threw = false;
    lc.time_a_ns = 99999;
    try { est.processLoopClosure(lc); } catch (const std::logic_error&) { threw = true; }
    CHECK(threw);
  }
  // --- BASELINE config 5 in shape (host part only): 240 poses on a two-loop figure-eight, drifting odometry,
  // accurate scan-matching factors, loop closures where the loops cross; driven through
  // IncrementalEstimator::registerPrior / estimate / estimateAndRemove exactly like the worker does
  {
    EstimatorParams ep;
    ep.laser_track_params.icp_input_filters_file = no_filters;
    ep.laser_track_params.use_icp_factors = false;
    IncrementalEstimator est(ep, 1u);
    const int n = 240;
    auto yawq = [](double a) { return std::array<double, 4>{std::cos(a / 2), 0, 0, std::sin(a / 2)}; };
    std::vector<SE3> truth;
    for (int i = 0; i < n; ++i) {  // lemniscate of Gerono, 0.8 m steps on average
      const double t = 4 * M_PI * i / n, x = 15 * std::sin(t), y = 15 * std::sin(t) * std::cos(t);
      const double dx = 15 * std::cos(t), dy = 15 * std::cos(2 * t);
      truth.push_back(SE3(yawq(std::atan2(dy, dx)), {x, y, 0.0}));
    }
    unsigned rng = 12345u;
    auto noise = [&](double s) { rng = rng * 1664525u + 1013904223u; return s * ((double)(rng >> 8) / 8388608.0 - 1.0); };
    SE3 dead = truth[0];
    std::vector<SE3> dead_reckoning{dead};
    const std::array<double, 6> odo_sig = {0.05, 0.05, 0.05, 0.01, 0.01, 0.01}, icp_sig = {0.005, 0.005, 0.005, 0.001, 0.001, 0.001};
    for (int i = 0; i < n; ++i) {
      FactorList f;
      Values v;
      if (i == 0) {
        Factor pr; pr.type = Factor::PRIOR; pr.key_b = 0; pr.measurement = truth[0]; pr.sigmas.fill(1e-7);
        f.push_back(pr); v[0] = truth[0];
        est.registerPrior(f, v, 0u);
        continue;
      }
      const SE3 rel = truth[i - 1].inverse() * truth[i];
      const SE3 odo = rel * SE3(yawq(0.002 + noise(0.002)), {0.02 + noise(0.01), noise(0.01), 0.0});   // biased
      const SE3 icp = rel * SE3(yawq(noise(0.0005)), {noise(0.003), noise(0.003), 0.0});
      dead = dead * odo;
      dead_reckoning.push_back(dead);
      Factor fo; fo.type = Factor::ODOMETRY; fo.key_a = (Key)i - 1; fo.key_b = (Key)i; fo.measurement = odo; fo.sigmas = odo_sig;
      Factor fi = fo; fi.type = Factor::ICP; fi.measurement = icp; fi.sigmas = icp_sig;
      f.push_back(fo); f.push_back(fi);
      v[(Key)i] = est.graph().values().at((Key)i - 1) * odo;   // new node from the odometry, like the trajectory curve
      est.estimate(f, v, i);
      // the two lobes cross at the origin: poses n/2 and n-1 come back to pose 0's place
      if (i == n / 2 || i == n - 1) {
        Factor lc; lc.type = Factor::LOOP_CLOSURE; lc.key_a = 0; lc.key_b = (Key)i;
        lc.measurement = truth[0].inverse() * truth[i] * SE3(yawq(noise(0.0005)), {noise(0.003), noise(0.003), 0.0});
        lc.sigmas = icp_sig; lc.cauchy = true;
        est.estimateAndRemove({lc}, {lc}, Values(), {0u, 0u}, i);
      }
    }
    auto rmse = [&](auto&& pose_of) {
      double s = 0;
      for (int i = 0; i < n; ++i) {
        const auto& p = pose_of(i).position(); const auto& q = truth[i].position();
        s += (p[0] - q[0]) * (p[0] - q[0]) + (p[1] - q[1]) * (p[1] - q[1]);
      }
      return std::sqrt(s / n);
    };
    const double e_dead = rmse([&](int i) -> const SE3& { return dead_reckoning[(size_t)i]; });
    const double e_graph = rmse([&](int i) -> const SE3& { return est.graph().values().at((Key)i); });
    CHECK(e_dead > 1.0);                      // the biased odometry alone is metres off
    CHECK(e_graph < 0.05 && e_graph < e_dead / 40);
    CHECK(est.graph().numFactors() == (size_t)(1 + 2 * (n - 1) + 2));
  }
  // --- the input filter chain file (laser_track.cpp:24-30): libpointmatcher's YAML list -> device filter descriptors
  {
    std::ifstream f(golden + "/input_filters.yaml");
    CHECK(f.good());
    DataPointsFilters chain(f);
    CHECK(chain.size() == 5);
    const auto& m = chain.modules();
    CHECK(m[0].type == LSGPU_FILTER_BOUNDING_BOX && m[0].flag == 1 && m[0].v[0] == -1.5f && m[0].v[5] == 0.5f);
    CHECK(m[1].type == LSGPU_FILTER_MAX_DIST && m[1].dim == -1 && m[1].v[0] == 60.f);
    CHECK(m[2].type == LSGPU_FILTER_MIN_DIST && m[2].v[0] == 2.5f);
    CHECK(m[3].type == LSGPU_FILTER_FIX_STEP_SAMPLING && m[3].v[0] == 3.f && m[3].v[1] == 5.f && std::fabs(m[3].v[2] - 1.3f) < 1e-6f);
    CHECK(m[4].type == LSGPU_FILTER_RANDOM_SAMPLING && m[4].v[0] == 0.8f);
    std::istringstream unknown("- SurfaceNormalDataPointsFilter: {knn: 5}\n");
    bool threw = false;
    try { DataPointsFilters bad(unknown); } catch (const ConfigError&) { threw = true; }
    CHECK(threw);
    std::istringstream nan_chain("- RemoveNaNDataPointsFilter\n- MinDistDataPointsFilter: {minDist: 1}\n");
    DataPointsFilters with_nan(nan_chain);
    CHECK(with_nan.size() == 2 && with_nan.modules()[0].type == LSGPU_FILTER_REMOVE_NAN);
    std::istringstream max_count("- MaxPointCountDataPointsFilter: {maxCount: 1000}\n");   // version-dependent selection: refused
    threw = false;
    try { DataPointsFilters bad(max_count); } catch (const ConfigError&) { threw = true; }
    CHECK(threw);
    std::istringstream typo("- MaxDistDataPointsFilter: {maxDistance: 5}\n");
    threw = false;
    try { DataPointsFilters bad(typo); } catch (const ConfigError&) { threw = true; }
    CHECK(threw);
    std::istringstream none("# nothing\n");
    DataPointsFilters empty(none);
    CHECK(empty.empty());
    DataPoints d; d.features = {1, 2, 3, 1};
    empty.apply(d);  // an empty chain touches neither the cloud nor the GPU
    CHECK(d.getNbPoints() == 1);
  }
  // --- loadFromYaml starts from EMPTY chains: an absent section is "no module", not "the default module"
  {
    ICP icp;
    std::istringstream y("referenceDataPointsFilters:\n  - SamplingSurfaceNormalDataPointsFilter: {knn: 9}\n"
                         "matcher:\n  KDTreeMatcher: {knn: 1}\nerrorMinimizer: PointToPlaneErrorMinimizer\n"
                         "transformationCheckers:\n  - CounterTransformationChecker: {maxIterationCount: 12}\n");
    icp.loadFromYaml(y);
    CHECK(icp.readingSamplingProb() < 0.f);        // no reading filter: every point
    CHECK(icp.config().trim_ratio == 1.0f);          // no outlier filter: every pair
    CHECK(icp.surfaceNormalKnn() == 9 && icp.config().max_iterations == 12);
    CHECK(icp.config().min_diff_rot < 0.f);          // no differential checker: only the counter stops the loop
    std::istringstream no_counter("referenceDataPointsFilters:\n  - SamplingSurfaceNormalDataPointsFilter\n"
                                  "matcher:\n  KDTreeMatcher\nerrorMinimizer: PointToPlaneErrorMinimizer\n");
    bool threw = false;
    try { icp.loadFromYaml(no_counter); } catch (const ConfigError&) { threw = true; }
    CHECK(threw);
    std::istringstream no_normals("matcher:\n  KDTreeMatcher\nerrorMinimizer: PointToPlaneErrorMinimizer\n"
                                  "transformationCheckers:\n  - CounterTransformationChecker\n");
    threw = false;
    try { icp.loadFromYaml(no_normals); } catch (const ConfigError&) { threw = true; }
    CHECK(threw);
  }
  // --- processLaserScan runs the ICP BEFORE it stores the scan (laser_track.cpp:112-119), processPoseAndLaserScan
  // after (:197-206): with three scans the first records one transformation (scan 0 -> 1), the second two
  {
    auto fake = [](const ICP&, const DataPoints&, const DataPoints&, const TransformationParameters& T) { return T; };
    LaserTrackParams p;
    p.icp_input_filters_file = no_filters;
    LaserTrackParams pb = p;
    if (const char* dump = std::getenv("LSGPU_TEST_DUMP_DIR")) {   // save_icp_results (laser_track.cpp:504-513): the test reads the files back
      pb.save_icp_results = true;
      pb.save_icp_results_dir = dump;
    }
    LaserTrack a(p, 0u), b(pb, 0u);
    a.icp().setComputeOverride(fake);
    b.icp().setComputeOverride(fake);
    for (int i = 0; i < 3; ++i) {
      Pose pose; pose.time_ns = 100 * (i + 1); pose.T_w = SE3({1, 0, 0, 0}, {0.8 * i, 0, 0});
      LaserScan sc; sc.time_ns = pose.time_ns; sc.scan.features = {float(i), 0, 0, 1, 0.1f * float(i), 1.f / 3.f, -2.5e-7f, 1};
      a.processPose(pose);
      a.processLaserScan(sc);
      b.processPoseAndLaserScan(pose, sc);
    }
    CHECK(a.getNumScans() == 3 && b.getNumScans() == 3);
    CHECK(a.getIcpTransformations().size() == 1 && b.getIcpTransformations().size() == 2);
    if (a.getIcpTransformations().size() == 1) {
      CHECK(a.getIcpTransformations()[0].time_a_ns == 100 && a.getIcpTransformations()[0].time_b_ns == 200);
      CHECK(std::fabs(a.getIcpTransformations()[0].T_a_b.position()[0] - 0.8) < 1e-6);
    }
    if (b.getIcpTransformations().size() == 2)
      CHECK(b.getIcpTransformations()[1].time_a_ns == 200 && b.getIcpTransformations()[1].time_b_ns == 300);
  }
  // --- a guess that is not rigid: TransformationError out of ICP::compute (its step 5 is a RigidTransformation::compute),
  // before anything touches the device; the loop-closure call site hands its guess over uncorrected and lets the
  // exception through (incremental_estimator.cpp:92-108)
  {
    ICP icp;
    DataPoints a; a.features = {0, 0, 0, 1, 1, 0, 0, 1, 0, 1, 0, 1, 0, 0, 1, 1};
    TransformationParameters T = identityTransformation();
    T[0] = 1.1f;   // det 1.1
    bool threw = false;
    try { icp.compute(a, a, T); } catch (const TransformationError&) { threw = true; }
    CHECK(threw);
    threw = false;
    try { icp.computeClouds(0, {1}, {identityTransformation()}, T); } catch (const TransformationError&) { threw = true; }
    CHECK(threw);
    T[0] = 1.0005f;   // inside checkParameters' tolerance (|1 - det| <= 1e-3): accepted; the fake stands in for the device
    icp.setComputeOverride([](const ICP&, const DataPoints&, const DataPoints&, const TransformationParameters& Ti) { return Ti; });
    CHECK(icp.compute(a, a, T)[0] == 1.0005f);
  }
  // --- no GPU => loud error (only checked when asked, i.e. on the CPU-only container)
  if (argc > 1 && std::string(argv[1]) == "--expect-no-gpu") {
    ICP icp;
    DataPoints a; a.features.assign(4 * 64, 1.f);
    for (int i = 0; i < 64; ++i) { a.features[4 * i] = float(i % 8); a.features[4 * i + 1] = float(i / 8); a.features[4 * i + 2] = 0.01f * float(i % 3); }
    bool threw = false;
    try { icp.compute(a, a, identityTransformation()); } catch (const DeviceError&) { threw = true; }
    CHECK(threw);
  }
  std::printf(fails ? "host_checks: %d FAILED\n" : "host_checks: ok\n", fails);
  return fails ? 1 : 0;
}
