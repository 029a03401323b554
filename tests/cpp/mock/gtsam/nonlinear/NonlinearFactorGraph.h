// PARSE-CHECK STAND-IN (tests/cpp/mock/README.md)
#pragma once
#include <gtsam/nonlinear/ExpressionFactor.h>
namespace gtsam {
class NonlinearFactorGraph {
 public:
  NonlinearFactorGraph();
  template <class F> void push_back(const F& factor);
  std::size_t size() const;
  KeySet keys() const;
};
}  // namespace gtsam
