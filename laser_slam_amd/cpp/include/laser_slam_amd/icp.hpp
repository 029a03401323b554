// icp.hpp -- C++ host-side mirror of the ICP object the reference owns (SURVEY.md §8b seam B2).
//
// The reference's `PointMatcher::ICP icp_` (laser_slam/include/laser_slam/laser_track.hpp:217,
// incremental_estimator.hpp:70) is used through three members only:
//     icp_.loadFromYaml(std::istream&)      laser_slam/src/laser_track.cpp:17
//     icp_.setDefault()                     laser_slam/src/laser_track.cpp:20
//     icp_.compute(reading, reference, T)   laser_slam/src/laser_track.cpp:496,
//                                           laser_slam/src/incremental_estimator.cpp:108
// laser_slam_amd::ICP keeps those names, argument meaning and error behaviour (ConvergenceError)
// over the C ABI of include/lsgpu_icp.h.  No libpointmatcher / Eigen / yaml-cpp dependency: clouds
// are plain float vectors in PointMatcher's own memory layout.
#pragma once
#include <array>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <istream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "lsgpu_icp.h"

namespace laser_slam_amd {

// PointMatcher<float>::DataPoints, reduced to what the path uses: features = (dim+1) x N column major
// (x,y,z,1 per point), optional "normals" descriptor 3 x N (laser_slam/.../common.hpp:14-15).
struct DataPoints {
  std::vector<float> features;  // 4 * N
  std::vector<float> normals;   // 3 * N or empty
  int64_t getNbPoints() const { return (int64_t)(features.size() / 4); }
  // DataPoints::concatenate (laser_track.cpp:485)
  void concatenate(const DataPoints& o) {
    const bool both = !normals.empty() && !o.normals.empty();
    features.insert(features.end(), o.features.begin(), o.features.end());
    if (both) normals.insert(normals.end(), o.normals.begin(), o.normals.end());
    else normals.clear();
  }
};

using TransformationParameters = std::array<float, 16>;  // 4x4 float, column major

inline TransformationParameters identityTransformation() {
  TransformationParameters T{};
  T[0] = T[5] = T[10] = T[15] = 1.f;
  return T;
}

// PointMatcher::ConvergenceError (caught at laser_track.cpp:499)
struct ConvergenceError : std::runtime_error {
  explicit ConvergenceError(const std::string& w) : std::runtime_error(w) {}
};
// PointMatcher::TransformationError: RigidTransformation::compute on a matrix that fails checkParameters -- also what
// ICP::compute throws for a non-rigid initial guess (its step 5 moves the reading with RigidTransformation)
struct TransformationError : std::runtime_error {
  explicit TransformationError(const std::string& w) : std::runtime_error(w) {}
};
struct ConfigError : std::runtime_error {
  explicit ConfigError(const std::string& w) : std::runtime_error(w) {}
};
struct DeviceError : std::runtime_error {
  explicit DeviceError(const std::string& w) : std::runtime_error(w) {}
};

// RigidTransformation (laser_track.cpp:33): compute / checkParameters / correctParameters
class RigidTransformation {
 public:
  static bool checkParameters(const TransformationParameters& T) { return lsgpu_check_rigid(T.data()) != 0; }
  static TransformationParameters correctParameters(const TransformationParameters& T) {
    TransformationParameters o;
    lsgpu_correct_rigid(T.data(), o.data());
    return o;
  }
  // features' = T * features; the "normals" descriptor is rotated.  Host arithmetic identical to the
  // device kernel (fma chain), so clouds built on either side agree bit for bit.
  static DataPoints compute(const DataPoints& in, const TransformationParameters& T) {
    if (!checkParameters(T)) throw TransformationError("RigidTransformation: matrix is not rigid");
    DataPoints out;
    const int64_t n = in.getNbPoints();
    out.features.resize((size_t)n * 4);
    auto M = [&](int r, int c) { return T[c * 4 + r]; };
    for (int64_t i = 0; i < n; ++i) {
      const float x = in.features[4 * i], y = in.features[4 * i + 1], z = in.features[4 * i + 2];
      for (int r = 0; r < 3; ++r)
        out.features[4 * i + r] = std::fma(M(r, 2), z, std::fma(M(r, 1), y, std::fma(M(r, 0), x, M(r, 3))));
      out.features[4 * i + 3] = in.features[4 * i + 3];
    }
    if (!in.normals.empty()) {
      out.normals.resize((size_t)n * 3);
      for (int64_t i = 0; i < n; ++i) {
        const float x = in.normals[3 * i], y = in.normals[3 * i + 1], z = in.normals[3 * i + 2];
        for (int r = 0; r < 3; ++r)
          out.normals[3 * i + r] = std::fma(M(r, 2), z, std::fma(M(r, 1), y, M(r, 0) * x));
      }
    }
    return out;
  }
};

// correctTransformationMatrix (laser_slam/include/laser_slam/common.hpp:136-149)
inline void correctTransformationMatrix(TransformationParameters* T) {
  if (!RigidTransformation::checkParameters(*T)) *T = RigidTransformation::correctParameters(*T);
}

namespace detail {

struct YamlModule { std::string section, name; std::map<std::string, std::string> params; };

inline std::string trim(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && std::isspace((unsigned char)s[a])) ++a;
  while (b > a && std::isspace((unsigned char)s[b - 1])) --b;
  return s.substr(a, b - a);
}
inline void parseInline(const std::string& body, std::map<std::string, std::string>* out) {  // {k: v, k: v}
  std::stringstream ss(body);
  std::string kv;
  while (std::getline(ss, kv, ',')) {
    const size_t c = kv.find(':');
    if (c != std::string::npos) (*out)[trim(kv.substr(0, c))] = trim(kv.substr(c + 1));
  }
}
// A libpointmatcher DataPointsFilters file: a top-level YAML sequence of modules, `- Name` / `- Name:` followed by an
// indented `key: value` block, or `- Name: {key: value, ...}`; `#` comments.  An empty document is an empty chain.
inline std::vector<YamlModule> parseYamlList(std::istream& in) {
  std::vector<YamlModule> mods;
  std::string line;
  int item_indent = -1;
  while (std::getline(in, line)) {
    const size_t hash = line.find('#');
    if (hash != std::string::npos) line = line.substr(0, hash);
    std::string t = trim(line);
    if (t.empty() || t == "---" || t == "[]") continue;
    int indent = 0;
    while (indent < (int)line.size() && line[indent] == ' ') ++indent;
    const bool item = t[0] == '-';
    if (item) t = trim(t.substr(1));
    const size_t c = t.find(':');
    const std::string key = trim(c == std::string::npos ? t : t.substr(0, c));
    const std::string val = c == std::string::npos ? "" : trim(t.substr(c + 1));
    if (item) {
      YamlModule m{"", key, {}};
      if (!val.empty() && val.front() == '{' && val.back() == '}') parseInline(val.substr(1, val.size() - 2), &m.params);
      else if (!val.empty()) throw std::runtime_error("input filters yaml: unexpected scalar after " + key);
      mods.push_back(m);
      item_indent = indent;
    } else {
      if (mods.empty() || indent <= item_indent) throw std::runtime_error("input filters yaml: a sequence of modules is expected");
      mods.back().params[key] = val;
    }
  }
  return mods;
}

}  // namespace detail

// PointMatcher::DataPointsFilters as LaserTrack uses it: built from a YAML stream (laser_slam/src/laser_track.cpp:27),
// applied in place to every incoming scan (laser_track.cpp:81, :146).  The supported modules run on the device
// (lsgpu_apply_point_filters); any other module is a configuration error, like an unknown name is for
// PointMatcher's registrar.  FixStepSampling keeps its step between calls, as the upstream object does.
class DataPointsFilters {
 public:
  DataPointsFilters() = default;
  explicit DataPointsFilters(std::istream& in, int device = 0) : device_(device) {
    for (const auto& m : detail::parseYamlList(in)) {
      auto num = [&](const char* key, double def) {
        auto it = m.params.find(key);
        return it == m.params.end() ? def : std::stod(it->second);
      };
      auto only = [&](std::initializer_list<const char*> keys) {
        for (const auto& kv : m.params) {
          bool known = false;
          for (const char* k : keys) known = known || kv.first == k;
          if (!known) throw ConfigError(m.name + ": unknown parameter " + kv.first);
        }
      };
      lsgpu_point_filter f;
      std::memset(&f, 0, sizeof(f));
      if (m.name == "MaxDistDataPointsFilter") {
        only({"dim", "maxDist"});
        f.type = LSGPU_FILTER_MAX_DIST; f.dim = (int)num("dim", -1); f.v[0] = (float)num("maxDist", 1.0);
      } else if (m.name == "MinDistDataPointsFilter") {
        only({"dim", "minDist"});
        f.type = LSGPU_FILTER_MIN_DIST; f.dim = (int)num("dim", -1); f.v[0] = (float)num("minDist", 1.0);
      } else if (m.name == "BoundingBoxDataPointsFilter") {
        only({"xMin", "xMax", "yMin", "yMax", "zMin", "zMax", "removeInside"});
        f.type = LSGPU_FILTER_BOUNDING_BOX;
        f.v[0] = (float)num("xMin", -1.0); f.v[1] = (float)num("xMax", 1.0);
        f.v[2] = (float)num("yMin", -1.0); f.v[3] = (float)num("yMax", 1.0);
        f.v[4] = (float)num("zMin", -1.0); f.v[5] = (float)num("zMax", 1.0);
        f.flag = (int)num("removeInside", 1);
      } else if (m.name == "FixStepSamplingDataPointsFilter") {
        only({"startStep", "endStep", "stepMult"});
        f.type = LSGPU_FILTER_FIX_STEP_SAMPLING;
        f.v[0] = (float)num("startStep", 10); f.v[1] = (float)num("endStep", 10); f.v[2] = (float)num("stepMult", 1);
      } else if (m.name == "RandomSamplingDataPointsFilter") {
        only({"prob"});
        f.type = LSGPU_FILTER_RANDOM_SAMPLING; f.v[0] = (float)num("prob", 0.75);
      } else if (m.name == "RemoveNaNDataPointsFilter") {
        only({});
        f.type = LSGPU_FILTER_REMOVE_NAN;
      } else {
        throw ConfigError("input filters: module " + m.name + " is not implemented on the HIP path");
      }
      filters_.push_back(f);
    }
  }
  ~DataPointsFilters() { if (h_) lsgpu_icp_destroy(h_); }
  DataPointsFilters(const DataPointsFilters&) = delete;
  DataPointsFilters& operator=(const DataPointsFilters&) = delete;
  DataPointsFilters(DataPointsFilters&& o) noexcept { *this = std::move(o); }
  DataPointsFilters& operator=(DataPointsFilters&& o) noexcept {
    if (this != &o) {
      if (h_) lsgpu_icp_destroy(h_);
      filters_ = std::move(o.filters_); h_ = o.h_; o.h_ = nullptr; device_ = o.device_; seed_ = o.seed_;
    }
    return *this;
  }

  size_t size() const { return filters_.size(); }
  bool empty() const { return filters_.empty(); }
  const std::vector<lsgpu_point_filter>& modules() const { return filters_; }
  void setSeed(int64_t seed) { seed_ = seed; }  // >= 0: reseed the draw stream at every apply(); < 0: continue it

  // DataPointsFilters::apply: in place; throws ConvergenceError if a filter is handed an empty cloud.
  void apply(DataPoints& cloud);

 private:
  std::vector<lsgpu_point_filter> filters_;
  lsgpu_icp* h_ = nullptr;
  int device_ = 0;
  int64_t seed_ = -1;
};

class ICP {
 public:
  ICP() { setDefault(); }
  explicit ICP(int device) : device_(device) { setDefault(); }
  ~ICP() { release(); }
  ICP(const ICP&) = delete;
  ICP& operator=(const ICP&) = delete;

  // ICP::setDefault(): RandomSampling 0.75 / SamplingSurfaceNormal knn 7 / KDTree 1,0 / TrimmedDist 0.85 /
  // PointToPlane / Counter 40 + Differential 1e-3, 1e-3, 3
  void setDefault() {
    lsgpu_icp_config_default(&cfg_);
    prob_ = 0.75f; knn_ = 7; ratio_ = 0.5f;
    release();
  }

  // Accepts the module chain of laser_slam/configurations/icp_default.yaml; any other module is a
  // configuration error (PointMatcher's registrar throws on unknown names as well).
  void loadFromYaml(std::istream& in) {
    lsgpu_icp_config c;
    lsgpu_icp_config_default(&c);
    float prob = 0.75f, ratio = 0.5f;
    int knn = 7;
    const auto mods = parseYaml(in);
    // libpointmatcher's loadFromYaml starts from EMPTY chains: a section the file does not mention means "no such
    // module", not "the default module".  A missing filter section therefore keeps every point (prob / ratio 1 is
    // not expressible for the reference filter: the normals come from it); the modules the device loop cannot run
    // without are required.
    bool has_reading = false, has_reference = false, has_matcher = false, has_outlier = false, has_minimizer = false,
         has_counter = false, has_differential = false;
    prob = -1.0f;   // no reading filter module: lsgpu_chain_config::reading_prob < 0 (every point, no draws)
    c.trim_ratio = 1.0f;
    for (const auto& m : mods) {
      const std::string& sec = m.section;
      const std::string& name = m.name;
      auto num = [&](const char* key, double def) {
        auto it = m.params.find(key);
        return it == m.params.end() ? def : std::stod(it->second);
      };
      if (sec == "readingDataPointsFilters" && name == "RandomSamplingDataPointsFilter") {
        if (has_reading) throw ConfigError("readingDataPointsFilters: one RandomSamplingDataPointsFilter at most");
        has_reading = true; prob = (float)num("prob", 0.75);
      } else if (sec == "referenceDataPointsFilters" && name == "SamplingSurfaceNormalDataPointsFilter") {
        if (has_reference) throw ConfigError("referenceDataPointsFilters: one SamplingSurfaceNormalDataPointsFilter at most");
        has_reference = true;
        knn = (int)num("knn", 7); ratio = (float)num("ratio", 0.5);
        if ((int)num("samplingMethod", 0) != 0) throw ConfigError("samplingMethod != 0 is not implemented");
      } else if (sec == "matcher" && name == "KDTreeMatcher") {
        has_matcher = true;
        if ((int)num("knn", 1) != 1 || num("epsilon", 0) != 0.0) throw ConfigError("only knn 1 / epsilon 0");
      } else if (sec == "outlierFilters" && name == "TrimmedDistOutlierFilter") {
        if (has_outlier) throw ConfigError("outlierFilters: one TrimmedDistOutlierFilter at most");
        has_outlier = true; c.trim_ratio = (float)num("ratio", 0.85);
      } else if (sec == "errorMinimizer" && name == "PointToPlaneErrorMinimizer") { has_minimizer = true; }
      else if (sec == "transformationCheckers" && name == "CounterTransformationChecker") {
        has_counter = true; c.max_iterations = (int)num("maxIterationCount", 40);
      } else if (sec == "transformationCheckers" && name == "DifferentialTransformationChecker") {
        has_differential = true;
        c.min_diff_rot = (float)num("minDiffRotErr", 0.001);
        c.min_diff_trans = (float)num("minDiffTransErr", 0.001);
        c.smooth_length = (int)num("smoothLength", 3);
      } else if (sec == "inspector" || sec == "logger") {}  // debug output only (yaml:32-44)
      else throw ConfigError(sec + ": module " + name + " is not implemented on the HIP path");
    }
    // what the device loop needs: normals for the point-to-plane minimiser, the 1-NN matcher, the minimiser itself
    // and a stopping rule.  (Absent reading filter: every point; absent outlier filter: every pair, ratio 1.)
    if (!has_reference) throw ConfigError("referenceDataPointsFilters: SamplingSurfaceNormalDataPointsFilter is required (it provides the normals)");
    if (!has_matcher) throw ConfigError("matcher: KDTreeMatcher is required");
    if (!has_minimizer) throw ConfigError("errorMinimizer: PointToPlaneErrorMinimizer is required");
    if (!has_counter) throw ConfigError("transformationCheckers: CounterTransformationChecker is required (the loop would not stop)");
    if (!has_differential) { c.min_diff_rot = -1.f; c.min_diff_trans = -1.f; c.smooth_length = 1; }  // never satisfied: the counter stops
    cfg_ = c; prob_ = prob; knn_ = knn; ratio_ = ratio;
    release();
  }

  static void requireRigid(const TransformationParameters& T_init) {
    if (!RigidTransformation::checkParameters(T_init)) throw TransformationError("ICP::compute: the initial guess is not a rigid transformation");
  }

  // T with p_reference = T * p_reading.  Throws ConvergenceError exactly where PointMatcher would.
  TransformationParameters compute(const DataPoints& reading, const DataPoints& reference,
                                   const TransformationParameters& T_init) {
    // step 5 of ICP::compute moves the reading by T_refMean_dataIn with RigidTransformation::compute, which refuses a
    // matrix that is not rigid; neither call site corrects its guess (laser_track.cpp:489-496, incremental_estimator.cpp:
    // 92-108: only the sub-map transforms go through correctTransformationMatrix) and neither catches this exception.
    // Upstream throws at its step 5, AFTER both filters have consumed their rand() draws.  So does the C ABI
    // (lsgpu_icp_align rejects the guess behind the filters), and so does this facade: a guess that is not rigid is handed
    // to the device all the same, whose filters run and whose step 5 refuses it; only then the exception is thrown.  A
    // caller that catches it and goes on sees the draws a libpointmatcher process would see.  (Without a device there is
    // no draw stream to keep in step: the exception is thrown at once.)
    const bool rigid = RigidTransformation::checkParameters(T_init);
#ifdef LSGPU_TEST_SEAMS
    if (override_) { requireRigid(T_init); return override_(*this, reading, reference, T_init); }
#endif
    if (!rigid) {
      try { ensureHandle(); } catch (const DeviceError&) { requireRigid(T_init); }
    } else {
      ensureHandle();
    }
    const int64_t nr = reference.getNbPoints(), nq = reading.getNbPoints();
    if (nr <= 0 || nq <= 0) { requireRigid(T_init); throw ConvergenceError("empty cloud"); }
    // ICP::compute steps 1-7 on the device: referenceDataPointsFilters, centring + grid,
    // readingDataPointsFilters, the loop (lsgpu_icp_compute)
    lsgpu_chain_config chain;
    lsgpu_chain_config_default(&chain);
    chain.reading_prob = prob_; chain.ssn_knn = knn_; chain.ssn_ratio = ratio_; chain.seed = seed_;
    TransformationParameters T = T_init;
    const int rc = lsgpu_icp_compute(h_, reading.features.data(), nq, reference.features.data(), nr, T_init.data(),
                                     &chain, T.data(), &stats_);
    if (!rigid) requireRigid(T_init);          // (rc is LSGPU_BAD_ARG from step 5, the filters have run)
    check(rc, "lsgpu_icp_compute");
#ifdef LSGPU_TEST_SEAMS
    if (observer_) observer_(*this, reading, reference, T_init, T);
#endif
    return T;
  }

  // ---- scans kept in HBM: upload once, match many times
  // Slots live in the handle; loadFromYaml / setDefault drop them (generation() changes).
  void uploadCloud(int slot, const DataPoints& cloud) {
    ensureHandle();
    check(lsgpu_cloud_upload(h_, slot, cloud.features.data(), cloud.getNbPoints()), "lsgpu_cloud_upload");
  }
  void releaseCloud(int slot) { if (h_) lsgpu_cloud_release(h_, slot); }
  bool hasCloud(int slot) {
    if (!h_) return false;
    int64_t n = -1;
    lsgpu_cloud_size(h_, slot, &n);
    return n >= 0;
  }
  unsigned generation() const { return generation_; }
  // compute() with reading = slot `reading` and reference = concat_i(T_i * slot refs[i]) assembled on the
  // device: what localScanToSubMap builds on the host at laser_track.cpp:474-486
  TransformationParameters computeClouds(int reading, const std::vector<int>& refs,
                                         const std::vector<TransformationParameters>& ref_T,
                                         const TransformationParameters& T_init) {
    const bool rigid = RigidTransformation::checkParameters(T_init);   // (refused at step 5, behind the filters: see compute())
    if (!rigid) {
      try { ensureHandle(); } catch (const DeviceError&) { requireRigid(T_init); }
    } else {
      ensureHandle();
    }
    if (refs.size() != ref_T.size()) throw std::logic_error("one transform per reference cloud");
    lsgpu_chain_config chain;
    lsgpu_chain_config_default(&chain);
    chain.reading_prob = prob_; chain.ssn_knn = knn_; chain.ssn_ratio = ratio_; chain.seed = seed_;
    std::vector<float> flat(16 * refs.size());
    for (size_t i = 0; i < refs.size(); ++i) std::memcpy(&flat[16 * i], ref_T[i].data(), 16 * sizeof(float));
    TransformationParameters T = T_init;
    const int rc = lsgpu_icp_compute_clouds(h_, reading, refs.data(), flat.data(), (int)refs.size(), T_init.data(), &chain,
                                            T.data(), &stats_);
    if (!rigid) requireRigid(T_init);
    check(rc, "lsgpu_icp_compute_clouds");
    return T;
  }

  // uploadCloud(reading, cloud) + computeClouds(reading, ...) in one call: the new scan crosses PCIe while the sub-map is
  // assembled and filtered (lsgpu_icp_compute_clouds_upload).  The slot holds the scan afterwards, also when this throws.
  TransformationParameters computeCloudsUploading(int reading, const DataPoints& cloud, const std::vector<int>& refs,
                                                  const std::vector<TransformationParameters>& ref_T,
                                                  const TransformationParameters& T_init) {
    const bool rigid = RigidTransformation::checkParameters(T_init);   // (refused at step 5, behind the filters: see compute())
    if (!rigid) {
      try { ensureHandle(); } catch (const DeviceError&) { requireRigid(T_init); }
    } else {
      ensureHandle();
    }
    if (refs.size() != ref_T.size()) throw std::logic_error("one transform per reference cloud");
    lsgpu_chain_config chain;
    lsgpu_chain_config_default(&chain);
    chain.reading_prob = prob_; chain.ssn_knn = knn_; chain.ssn_ratio = ratio_; chain.seed = seed_;
    std::vector<float> flat(16 * refs.size());
    for (size_t i = 0; i < refs.size(); ++i) std::memcpy(&flat[16 * i], ref_T[i].data(), 16 * sizeof(float));
    TransformationParameters T = T_init;
    const int rc = lsgpu_icp_compute_clouds_upload(h_, reading, cloud.features.data(), cloud.getNbPoints(), refs.data(),
                                                   flat.data(), (int)refs.size(), T_init.data(), &chain, T.data(), &stats_);
    if (!rigid) requireRigid(T_init);
    check(rc, "lsgpu_icp_compute_clouds_upload");
    return T;
  }

  // >= 0: reseed the filters' draw stream at every compute() (reproducible runs); < 0: continue it
  void setSeed(int64_t seed) { seed_ = seed; }
  int64_t seed() const { return seed_; }

#ifdef LSGPU_TEST_SEAMS
  // (compiled only into the parity-test drivers, tests/cpp/*: -DLSGPU_TEST_SEAMS; the shipped header has no hook)
  // Test seam: replaces compute() by another implementation of the same call (the parity tests inject the CPU
  // oracle here, so that LaserTrack / IncrementalEstimator run unchanged on either ICP).  Never set by the
  // product; with an override in place nothing touches the GPU (no handle is created).
  using ComputeOverride = std::function<TransformationParameters(const ICP&, const DataPoints& reading,
                                                                 const DataPoints& reference,
                                                                 const TransformationParameters& T_init)>;
  void setComputeOverride(ComputeOverride f) { override_ = std::move(f); }
  // Test seam: called after every successful device compute() with its inputs and its result (the parity tests run
  // the CPU oracle on exactly the clouds the facade handed to the device).  Never set by the product.
  using ComputeObserver = std::function<void(const ICP&, const DataPoints& reading, const DataPoints& reference,
                                             const TransformationParameters& T_init, const TransformationParameters& T)>;
  void setComputeObserver(ComputeObserver f) { observer_ = std::move(f); }
  bool hasComputeObserver() const { return (bool)observer_; }
  // the resident-scan path (computeClouds) never sees the assembled sub-map on the host; LaserTrack assembles it for the
  // observer's benefit when (and only when) one is set
  void notifyObserver(const DataPoints& reading, const DataPoints& reference, const TransformationParameters& T_init,
                      const TransformationParameters& T) const { if (observer_) observer_(*this, reading, reference, T_init, T); }
  bool hasComputeOverride() const { return (bool)override_; }
#else
  bool hasComputeOverride() const { return false; }
#endif

  // Steps 2-7 on already filtered clouds (device or host pointers).
  TransformationParameters computeFiltered(const float* reading_xyz1, int64_t nq, const float* ref_xyz1,
                                           const float* ref_normals, int64_t nr,
                                           const TransformationParameters& T_init) {
    ensureHandle();
    check(lsgpu_icp_set_reference(h_, ref_xyz1, ref_normals, nr), "lsgpu_icp_set_reference");
    TransformationParameters T = T_init;
    check(lsgpu_icp_align(h_, reading_xyz1, nq, T_init.data(), T.data(), &stats_), "lsgpu_icp_align");
    return T;
  }

  const lsgpu_icp_stats& lastStats() const { return stats_; }
  const lsgpu_icp_config& config() const { return cfg_; }
  lsgpu_icp* handle() { ensureHandle(); return h_; }  // the C-ABI handle (device conversions of ros_msgs.hpp)
  float readingSamplingProb() const { return prob_; }
  int surfaceNormalKnn() const { return knn_; }
  float surfaceNormalRatio() const { return ratio_; }

 private:
  using Module = detail::YamlModule;
  static std::string trim(const std::string& s) { return detail::trim(s); }
  static void parseInline(const std::string& body, std::map<std::string, std::string>* out) { detail::parseInline(body, out); }
  // The YAML subset libpointmatcher configurations use: top-level `section:`, module either as a list
  // item `- Name:` / `- Name` or as a mapping `Name:` / scalar `section: Name`, parameters as an
  // indented `key: value` block or an inline `{...}` map; `#` comments.
  static std::vector<Module> parseYaml(std::istream& in) {
    std::vector<Module> mods;
    std::string line, section;
    int mod_indent = -1;
    while (std::getline(in, line)) {
      const size_t hash = line.find('#');
      if (hash != std::string::npos) line = line.substr(0, hash);
      if (trim(line).empty()) continue;
      int indent = 0;
      while (indent < (int)line.size() && line[indent] == ' ') ++indent;
      std::string t = trim(line);
      bool item = false;
      if (t[0] == '-') { item = true; t = trim(t.substr(1)); }
      const size_t c = t.find(':');
      std::string key = trim(c == std::string::npos ? t : t.substr(0, c));
      std::string val = c == std::string::npos ? "" : trim(t.substr(c + 1));
      if (indent == 0 && !item) {  // section
        section = key;
        mod_indent = -1;
        if (!val.empty() && val[0] != '{') { mods.push_back({section, val, {}}); }
        continue;
      }
      if (section.empty()) throw ConfigError("yaml: content before the first section");
      const bool is_param = mod_indent >= 0 && indent > mod_indent && !item;
      if (is_param) {
        mods.back().params[key] = val;
      } else {  // a module
        Module m{section, key, {}};
        if (!val.empty() && val.front() == '{' && val.back() == '}') parseInline(val.substr(1, val.size() - 2), &m.params);
        mods.push_back(m);
        mod_indent = indent;
      }
    }
    return mods;
  }

  void ensureHandle() {
    if (h_) return;
    const int rc = lsgpu_icp_create(&cfg_, device_, &h_);
    if (rc == LSGPU_BAD_CONFIG) throw ConfigError("lsgpu_icp_create: bad configuration");
    if (rc != LSGPU_OK) throw DeviceError("lsgpu_icp_create failed (no ROCm GPU visible?)");
  }
  void release() { if (h_) { lsgpu_icp_destroy(h_); h_ = nullptr; ++generation_; } }
  void check(int rc, const char* what) {
    if (rc == LSGPU_OK) return;
    const std::string msg = std::string(what) + ": " + lsgpu_strerror(rc) + " [" + lsgpu_last_error(h_) + "]";
    if (rc == LSGPU_NO_CONVERGENCE) throw ConvergenceError(msg);
    if (rc == LSGPU_BAD_CONFIG) throw ConfigError(msg);
    throw DeviceError(msg);
  }

  int device_ = 0;
  lsgpu_icp_config cfg_{};
  lsgpu_icp* h_ = nullptr;
  lsgpu_icp_stats stats_{};
  float prob_ = 0.75f, ratio_ = 0.5f;
  int knn_ = 7;
  int64_t seed_ = -1;
  unsigned generation_ = 0;
#ifdef LSGPU_TEST_SEAMS
  ComputeOverride override_;
  ComputeObserver observer_;
#endif
};

inline void DataPointsFilters::apply(DataPoints& cloud) {
  if (filters_.empty()) return;
  if (!h_) {
    lsgpu_icp_config c;
    lsgpu_icp_config_default(&c);
    if (lsgpu_icp_create(&c, device_, &h_) != LSGPU_OK) throw DeviceError("lsgpu_icp_create failed (no ROCm GPU visible?)");
  }
  const int64_t n = cloud.getNbPoints();
  std::vector<float> out((size_t)std::max<int64_t>(n, 1) * 4);
  int64_t m = 0;
  // A cloud with descriptors (normals): the device filters look at x, y, z only and carry the 4th component through, so
  // the point's index travels in it and the descriptor columns of the survivors are picked afterwards -- the filters keep
  // or drop whole columns, features and descriptors alike, as upstream's do.
  const bool tagged = !cloud.normals.empty();
  std::vector<float> in_tagged;
  if (tagged) {
    in_tagged = cloud.features;
    for (int64_t i = 0; i < n; ++i) { const uint32_t tag = (uint32_t)i; std::memcpy(&in_tagged[(size_t)(4 * i + 3)], &tag, 4); }
  }
  const int rc = lsgpu_apply_point_filters(h_, filters_.data(), (int)filters_.size(), tagged ? in_tagged.data() : cloud.features.data(),
                                           n, seed_, out.data(), &m);
  if (rc == LSGPU_NO_CONVERGENCE) throw ConvergenceError("no points to filter");
  if (rc == LSGPU_BAD_CONFIG) throw ConfigError(std::string("input filters: ") + lsgpu_last_error(h_));
  if (rc != LSGPU_OK) throw DeviceError(std::string("lsgpu_apply_point_filters: ") + lsgpu_strerror(rc) + " [" + lsgpu_last_error(h_) + "]");
  out.resize((size_t)m * 4);
  if (tagged) {
    std::vector<float> nrm((size_t)m * 3);
    for (int64_t j = 0; j < m; ++j) {
      uint32_t tag;
      std::memcpy(&tag, &out[(size_t)(4 * j + 3)], 4);
      out[(size_t)(4 * j + 3)] = cloud.features[(size_t)(4 * (int64_t)tag + 3)];
      for (int d = 0; d < 3; ++d) nrm[(size_t)(3 * j + d)] = cloud.normals[(size_t)(3 * (int64_t)tag + d)];
    }
    cloud.normals.swap(nrm);
  }
  cloud.features.swap(out);
}

}  // namespace laser_slam_amd
