// PARSE-CHECK STAND-IN (tests/cpp/mock/README.md): gtsam::ISAM2 as incremental_estimator.cpp:17-20, 151-163, 253, 272 uses it
#pragma once
#include <gtsam/nonlinear/NonlinearFactorGraph.h>
namespace gtsam {
struct ISAM2Params {
  void setRelinearizeSkip(int skip);
  void setRelinearizeThreshold(double threshold);
};
struct ISAM2Result {
  FactorIndices newFactorsIndices;
  void print(const char* s = "") const;
};
class ISAM2 {
 public:
  ISAM2();
  explicit ISAM2(const ISAM2Params& params);
  ISAM2Result update(const NonlinearFactorGraph& newFactors = NonlinearFactorGraph(), const Values& newTheta = Values(),
                     const FactorIndices& removeFactorIndices = FactorIndices());
  Values calculateEstimate() const;
};
}  // namespace gtsam
