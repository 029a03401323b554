// PARSE-CHECK STAND-IN (tests/cpp/mock/README.md)
#pragma once
#include <gtsam/nonlinear/Values.h>
namespace gtsam {
template <int R, int C> class OptionalJacobian {};
template <class T> class Expression {
 public:
  Expression(const T& constant);
  Expression(const Key& key);
};
}  // namespace gtsam
