// lsgpu_icp_shim.hpp -- the drop-in a laser_slam maintainer puts where `PointMatcher::ICP icp_` and
// `PointMatcher::DataPointsFilters input_filters_` are declared today:
//     laser_slam/include/laser_slam/laser_track.hpp:217,220        incremental_estimator.hpp:70
//   - PointMatcher::ICP icp_;                    + LsgpuICP<PointMatcher> icp_;
//   - PointMatcher::DataPointsFilters input_filters_;   + LsgpuDataPointsFilters<PointMatcher> input_filters_;
// Nothing else changes: loadFromYaml / setDefault / compute keep their signatures (laser_track.cpp:17,20,496,
// incremental_estimator.cpp:108), ConvergenceError is the PointMatcher one (caught at laser_track.cpp:499),
// input_filters_ = LsgpuDataPointsFilters<PointMatcher>(ifs) and .apply(scan.scan) read as before (:27,:81,:146).
//
// The header is a template over the PointMatcher type so that the SAME code is compiled and tested in this
// repository against the dependency-free mirror types (LsgpuMirrorPM below, tests/cpp/shim_check.cpp) and compiles
// against the real `PointMatcher<float>` (Eigen matrices) through LsgpuCloudTraits: the only operations it needs from
// a cloud are "pointer to the (dim+1) x N column-major features", "number of points" and "keep these columns".
#pragma once
#include <cstdint>
#include <cstring>
#include <istream>
#include <stdexcept>
#include <string>
#include <vector>

#include "laser_slam_amd/icp.hpp"   // YAML subset parser + module -> device configuration (no GPU touched by parsing)
#include "lsgpu_icp.h"

// ---- how the shim looks at a cloud.  Primary template: the real libpointmatcher (Eigen) types.
template <class PM>
struct LsgpuCloudTraits {
  using DataPoints = typename PM::DataPoints;
  static const float* features(const DataPoints& d) { return d.features.data(); }      // 4 x N, column major
  static int64_t size(const DataPoints& d) { return (int64_t)d.features.cols(); }
  // keep the listed columns (ascending) of features and of every descriptor, drop the others
  static void keepColumns(DataPoints& d, const std::vector<int64_t>& cols) {
    for (size_t j = 0; j < cols.size(); ++j) {
      if ((int64_t)j == cols[j]) continue;
      d.features.col((Eigen_Index)j) = d.features.col((Eigen_Index)cols[j]);
      if (d.descriptors.cols() > 0) d.descriptors.col((Eigen_Index)j) = d.descriptors.col((Eigen_Index)cols[j]);
    }
    d.features.conservativeResize(d.features.rows(), (Eigen_Index)cols.size());
    if (d.descriptors.cols() > 0) d.descriptors.conservativeResize(d.descriptors.rows(), (Eigen_Index)cols.size());
  }
 private:
  using Eigen_Index = decltype(std::declval<DataPoints>().features.cols());
};

// ---- the in-tree mirror types (what the tests instantiate the shim with)
struct LsgpuMirrorPM {
  using DataPoints = laser_slam_amd::DataPoints;
  using TransformationParameters = laser_slam_amd::TransformationParameters;
  using ConvergenceError = laser_slam_amd::ConvergenceError;
};
template <>
struct LsgpuCloudTraits<LsgpuMirrorPM> {
  using DataPoints = laser_slam_amd::DataPoints;
  static const float* features(const DataPoints& d) { return d.features.data(); }
  static int64_t size(const DataPoints& d) { return d.getNbPoints(); }
  static void keepColumns(DataPoints& d, const std::vector<int64_t>& cols) {
    const bool nrm = !d.normals.empty();
    for (size_t j = 0; j < cols.size(); ++j) {
      if ((int64_t)j == cols[j]) continue;
      std::memcpy(&d.features[4 * j], &d.features[4 * (size_t)cols[j]], 4 * sizeof(float));
      if (nrm) std::memcpy(&d.normals[3 * j], &d.normals[3 * (size_t)cols[j]], 3 * sizeof(float));
    }
    d.features.resize(4 * cols.size());
    if (nrm) d.normals.resize(3 * cols.size());
  }
};

// ---- PointMatcher::ICP stand-in
template <class PM>
class LsgpuICP {
 public:
  using DataPoints = typename PM::DataPoints;
  using TransformationParameters = typename PM::TransformationParameters;
  using Traits = LsgpuCloudTraits<PM>;

  explicit LsgpuICP(int device = 0) : device_(device) { setDefault(); }
  ~LsgpuICP() { reset(); }
  LsgpuICP(const LsgpuICP&) = delete;
  LsgpuICP& operator=(const LsgpuICP&) = delete;

  void setDefault() {                                   // laser_track.cpp:20
    laser_slam_amd::ICP parsed;                         // == ICP::setDefault() values
    take(parsed);
  }
  void loadFromYaml(std::istream& in) {                 // laser_track.cpp:17
    laser_slam_amd::ICP parsed;
    try {
      parsed.loadFromYaml(in);                          // the module chain of icp_default.yaml; anything else throws
    } catch (const laser_slam_amd::ConfigError& e) {
      throw std::runtime_error(std::string("LsgpuICP::loadFromYaml: ") + e.what());   // PointMatcher: InvalidModuleType
    }
    take(parsed);
  }
  void setSeed(int64_t seed) { seed_ = seed; }          // >= 0: reproducible filter draws per compute()

  // T with p_reference = T * p_reading (laser_track.cpp:496, incremental_estimator.cpp:108).  The whole chain runs on
  // the device: reference filter, centring + grid, reading filter, the loop.  features are (dim+1) x N column major ==
  // x,y,z,1 per point and TransformationParameters is a 4x4 column-major float matrix: both are passed as they are.
  TransformationParameters compute(const DataPoints& reading, const DataPoints& reference,
                                   const TransformationParameters& T_init) {
    if (!h_ && lsgpu_icp_create(&cfg_, device_, &h_) != LSGPU_OK)
      throw std::runtime_error("LsgpuICP: lsgpu_icp_create failed (no ROCm GPU visible?)");
    lsgpu_chain_config chain;
    lsgpu_chain_config_default(&chain);
    chain.reading_prob = prob_; chain.ssn_knn = knn_; chain.ssn_ratio = ratio_; chain.seed = seed_;
    TransformationParameters T = T_init;
    const int rc = lsgpu_icp_compute(h_, Traits::features(reading), Traits::size(reading), Traits::features(reference),
                                     Traits::size(reference), T_init.data(), &chain, T.data(), &stats_);
    if (rc == LSGPU_NO_CONVERGENCE) throw typename PM::ConvergenceError(lsgpu_last_error(h_));
    if (rc != LSGPU_OK) throw std::runtime_error(std::string("LsgpuICP::compute: ") + lsgpu_strerror(rc) + " [" + lsgpu_last_error(h_) + "]");
    return T;
  }
  const lsgpu_icp_stats& lastStats() const { return stats_; }

 private:
  void take(const laser_slam_amd::ICP& parsed) {
    cfg_ = parsed.config();
    prob_ = parsed.readingSamplingProb(); knn_ = parsed.surfaceNormalKnn(); ratio_ = parsed.surfaceNormalRatio();
    reset();
  }
  void reset() { if (h_) { lsgpu_icp_destroy(h_); h_ = nullptr; } }
  int device_ = 0;
  lsgpu_icp_config cfg_{};
  lsgpu_icp* h_ = nullptr;
  lsgpu_icp_stats stats_{};
  float prob_ = 0.75f, ratio_ = 0.5f;
  int knn_ = 7;
  int64_t seed_ = -1;
};

// ---- PointMatcher::DataPointsFilters stand-in (laser_track.cpp:24-30, :81, :146)
template <class PM>
class LsgpuDataPointsFilters {
 public:
  using DataPoints = typename PM::DataPoints;
  using Traits = LsgpuCloudTraits<PM>;
  LsgpuDataPointsFilters() = default;
  explicit LsgpuDataPointsFilters(std::istream& in, int device = 0) : device_(device) {
    try {
      laser_slam_amd::DataPointsFilters parsed(in, device);
      filters_ = parsed.modules();
    } catch (const laser_slam_amd::ConfigError& e) {
      throw std::runtime_error(std::string("LsgpuDataPointsFilters: ") + e.what());
    }
  }
  ~LsgpuDataPointsFilters() { if (h_) lsgpu_icp_destroy(h_); }
  LsgpuDataPointsFilters(const LsgpuDataPointsFilters&) = delete;
  LsgpuDataPointsFilters& operator=(const LsgpuDataPointsFilters&) = delete;
  LsgpuDataPointsFilters(LsgpuDataPointsFilters&& o) noexcept { *this = std::move(o); }
  LsgpuDataPointsFilters& operator=(LsgpuDataPointsFilters&& o) noexcept {
    if (this != &o) { if (h_) lsgpu_icp_destroy(h_); filters_ = std::move(o.filters_); h_ = o.h_; o.h_ = nullptr; device_ = o.device_; seed_ = o.seed_; }
    return *this;
  }
  void setSeed(int64_t seed) { seed_ = seed; }
  size_t size() const { return filters_.size(); }

  // In place.  The device filters only look at x, y, z and carry the 4th component through untouched, so the column
  // index travels in it: what comes back tells which columns survived, and the descriptors are thinned accordingly.
  void apply(DataPoints& cloud) {
    if (filters_.empty()) return;
    const int64_t n = Traits::size(cloud);
    if (n == 0) throw typename PM::ConvergenceError("no points to filter");   // as DataPointsFilters::apply does upstream
    if (!h_) {
      lsgpu_icp_config c;
      lsgpu_icp_config_default(&c);
      if (lsgpu_icp_create(&c, device_, &h_) != LSGPU_OK) throw std::runtime_error("LsgpuDataPointsFilters: no ROCm GPU visible");
    }
    std::vector<float> tagged((size_t)n * 4), out((size_t)n * 4);
    std::memcpy(tagged.data(), Traits::features(cloud), (size_t)n * 16);
    for (int64_t i = 0; i < n; ++i) { const uint32_t tag = (uint32_t)i; std::memcpy(&tagged[4 * (size_t)i + 3], &tag, 4); }
    int64_t m = 0;
    const int rc = lsgpu_apply_point_filters(h_, filters_.data(), (int)filters_.size(), tagged.data(), n, seed_, out.data(), &m);
    if (rc == LSGPU_NO_CONVERGENCE) throw typename PM::ConvergenceError("no points to filter");
    if (rc != LSGPU_OK) throw std::runtime_error(std::string("LsgpuDataPointsFilters::apply: ") + lsgpu_strerror(rc));
    std::vector<int64_t> cols((size_t)m);
    for (int64_t j = 0; j < m; ++j) { uint32_t tag; std::memcpy(&tag, &out[4 * (size_t)j + 3], 4); cols[(size_t)j] = (int64_t)tag; }
    Traits::keepColumns(cloud, cols);
  }

 private:
  std::vector<lsgpu_point_filter> filters_;
  lsgpu_icp* h_ = nullptr;
  int device_ = 0;
  int64_t seed_ = -1;
};
