// PARSE-CHECK STAND-IN (tests/cpp/mock/README.md): laser_track.cpp:420-428
#pragma once
#include <gtsam/nonlinear/NonlinearFactorGraph.h>
namespace gtsam {
class Marginals {
 public:
  Marginals(const NonlinearFactorGraph& graph, const Values& solution);
  Matrix marginalCovariance(Key variable) const;
};
}  // namespace gtsam
