// host_checks.cpp -- CPU-only checks of the C++ mirror (no GPU needed): YAML subset parser, SE3
// algebra, RigidTransformation on the host, sub-map bookkeeping errors, and that a GPU-less box
// produces a loud DeviceError from compute() instead of a fallback.
#include <cmath>
#include <cstdio>
#include <sstream>

#include "laser_slam_amd/laser_track.hpp"

using namespace laser_slam_amd;
static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

int main(int argc, char** argv) {
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
    std::stringstream inl("outlierFilters:\n  - TrimmedDistOutlierFilter: {ratio: 0.9}\nerrorMinimizer: PointToPlaneErrorMinimizer\n");
    icp.loadFromYaml(inl);
    CHECK(std::fabs(icp.config().trim_ratio - 0.9f) < 1e-7f && icp.config().smooth_length == 3);
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
    LaserTrack track(p, 2u);
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
