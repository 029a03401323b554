// PARSE-CHECK STAND-IN (tests/cpp/mock/README.md)
#pragma once
#include <gtsam/nonlinear/Expression.h>
namespace gtsam {
class NonlinearFactor { public: virtual ~NonlinearFactor(); };
template <class T> class ExpressionFactor : public NonlinearFactor {
 public:
  ExpressionFactor(const SharedNoiseModel& noiseModel, const T& measurement, const Expression<T>& expression);
};
}  // namespace gtsam
