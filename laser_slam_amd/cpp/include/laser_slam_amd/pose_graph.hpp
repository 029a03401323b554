// pose_graph.hpp -- host pose-graph back end standing in for gtsam::ISAM2 as laser_slam uses it
// (SURVEY.md §8f row N2): SE3 variables, prior and relative-pose factors with diagonal sigmas
// [translation; rotation] and an optional Cauchy(1) m-estimator
// (laser_slam/src/laser_track.cpp:37-64,431-458, incremental_estimator.cpp:28-46), solved by
// Gauss-Newton steps: each IncrementalEstimator update runs as many steps as the reference runs iSAM2
// updates (incremental_estimator.cpp:151-163).  Host arithmetic in double, not on the accelerated path.
//
// Like iSAM2 -- and unlike rounds 1-5, whose every step re-linearised and re-solved the WHOLE graph: 21 ms
// per pose at 2000 poses, O(N^2) over a sequence -- a step only touches the variables that still move (round 6):
//   * a new variable and the variables of a new factor are ACTIVE; a loop-closure factor or a removed factor
//     activates every variable (the whole loop moves);
//   * a step linearises the factors of the active variables only, the others are constants in them;
//   * a variable whose update stayed below kSettled leaves the active set; one whose update exceeded kExpand
//     activates its neighbours (so that a correction can travel as far as it has to).
// New odometry / ICP factors hang a new pose on the end of the chain: relative factors are invariant under a
// common motion of their two poses, so the optimum of the older poses does not change and the active set stays
// a handful of poses between loop closures; after one it is the whole graph for the few steps Gauss-Newton needs.
// The result equals whole-graph steps to the two thresholds (tests/cpp/host_checks.cpp compares both).
//
// Error of a factor: localCoordinates(measurement, prediction) in minkindr's chart
// [position; rotation vector] (se3.hpp), whitened by the sigmas.  Linear solve: block elimination
// (6x6 blocks) in greedy minimum-degree order, so a trajectory chain with loop closures costs O(N).
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <map>
#include <set>
#include <stdexcept>
#include <vector>

#include "laser_slam_amd/se3.hpp"

namespace laser_slam_amd {

using Key = size_t;
using Values = std::map<Key, SE3>;

// One record per factor the reference pushes into the gtsam graph.
struct Factor {
  enum Type { PRIOR, ODOMETRY, ICP, LOOP_CLOSURE } type = PRIOR;
  Key key_a = 0, key_b = 0;        // PRIOR uses key_b only
  SE3 measurement;                 // T_w (prior) or T_a_b (between)
  std::array<double, 6> sigmas{};  // diagonal noise model [translation; rotation]
  bool cauchy = false;             // Cauchy(1) m-estimator (laser_track.cpp:47-54)
  bool fix_first_node = false;     // between factor whose first pose is the constant `fixed_a`
  SE3 fixed_a;                     //   (makeRelativeMeasurementFactor(..., fix_first_node), laser_track.cpp:440-444)
};
using FactorList = std::vector<Factor>;

namespace detail {

using M6 = std::array<double, 36>;
using V6 = std::array<double, 6>;

inline M6 mul(const M6& a, const M6& b) {
  M6 c{};
  for (int i = 0; i < 6; ++i)
    for (int k = 0; k < 6; ++k) {
      const double v = a[i * 6 + k];
      for (int j = 0; j < 6; ++j) c[i * 6 + j] += v * b[k * 6 + j];
    }
  return c;
}
inline V6 mul(const M6& a, const V6& x) {
  V6 y{};
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) y[i] += a[i * 6 + j] * x[j];
  return y;
}
// inverse of a symmetric positive definite 6x6 (Cholesky); throws if it is not
inline M6 invSpd(const M6& a) {
  double L[6][6] = {};
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = a[i * 6 + j];
      for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
      if (i == j) {
        if (!(s > 0)) throw std::runtime_error("pose graph: information matrix is not positive definite");
        L[i][i] = std::sqrt(s);
      } else {
        L[i][j] = s / L[j][j];
      }
    }
  M6 inv{};
  for (int c = 0; c < 6; ++c) {
    double y[6], x[6];
    for (int i = 0; i < 6; ++i) {
      double s = (i == c) ? 1.0 : 0.0;
      for (int k = 0; k < i; ++k) s -= L[i][k] * y[k];
      y[i] = s / L[i][i];
    }
    for (int i = 5; i >= 0; --i) {
      double s = y[i];
      for (int k = i + 1; k < 6; ++k) s -= L[k][i] * x[k];
      x[i] = s / L[i][i];
    }
    for (int i = 0; i < 6; ++i) inv[i * 6 + c] = x[i];
  }
  return inv;
}

// H x = g for a block-sparse symmetric positive definite H (both triangles stored per row)
inline std::vector<V6> solveBlockSparse(std::vector<std::map<int, M6>>& rows, std::vector<V6>& g) {
  const int n = (int)rows.size();
  struct Pivot { int p; std::vector<int> nbrs; std::vector<M6> blocks; M6 dinv; V6 rhs; };
  std::vector<Pivot> pivots;
  pivots.reserve(n);
  std::set<std::pair<int, int>> order;  // (degree, node)
  std::vector<int> degree(n);
  for (int i = 0; i < n; ++i) { degree[i] = (int)rows[i].size(); order.insert({degree[i], i}); }
  while (!order.empty()) {
    const int p = order.begin()->second;
    order.erase(order.begin());
    Pivot pv;
    pv.p = p;
    pv.dinv = invSpd(rows[p].at(p));
    pv.rhs = g[p];
    for (const auto& kv : rows[p])
      if (kv.first != p) { pv.nbrs.push_back(kv.first); pv.blocks.push_back(kv.second); }
    for (size_t a = 0; a < pv.nbrs.size(); ++a) {
      const int i = pv.nbrs[a];
      order.erase({degree[i], i});
      const M6 lip = mul(rows[i].at(p), pv.dinv);  // H_ip D^-1
      const V6 lb = mul(lip, pv.rhs);
      for (int k = 0; k < 6; ++k) g[i][k] -= lb[k];
      for (size_t b = 0; b < pv.nbrs.size(); ++b) {
        const M6 upd = mul(lip, pv.blocks[b]);     // H_ip D^-1 H_pj
        M6& dst = rows[i][pv.nbrs[b]];             // (creates the fill block, zero initialised)
        for (int k = 0; k < 36; ++k) dst[k] -= upd[k];
      }
      rows[i].erase(p);
    }
    for (int i : pv.nbrs) { degree[i] = (int)rows[i].size(); order.insert({degree[i], i}); }
    pivots.push_back(std::move(pv));
  }
  std::vector<V6> x(n);
  for (int k = (int)pivots.size() - 1; k >= 0; --k) {
    const Pivot& pv = pivots[k];
    V6 r = pv.rhs;
    for (size_t a = 0; a < pv.nbrs.size(); ++a) {
      const V6 t = mul(pv.blocks[a], x[pv.nbrs[a]]);
      for (int i = 0; i < 6; ++i) r[i] -= t[i];
    }
    x[pv.p] = mul(pv.dinv, r);
  }
  return x;
}

}  // namespace detail

class PoseGraph {
 public:
  // new variables; a key that already exists keeps its current estimate (gtsam would throw)
  void insert(const Values& v) {
    for (const auto& kv : v)
      if (values_.insert(kv).second) active_.insert(kv.first);
  }
  // returns the factor's index (what ISAM2Result::newFactorsIndices reports)
  size_t addFactor(const Factor& f) {
    factors_.push_back(f);
    alive_.push_back(true);
    const size_t idx = factors_.size() - 1;
    const bool has_a = f.type != Factor::PRIOR && !f.fix_first_node;
    if (has_a) { adjacency_[f.key_a].push_back(idx); active_.insert(f.key_a); }
    adjacency_[f.key_b].push_back(idx);
    active_.insert(f.key_b);
    if (f.type == Factor::LOOP_CLOSURE) activateAll();
    return idx;
  }
  void removeFactor(size_t index) {
    if (index >= alive_.size() || !alive_[index]) throw std::out_of_range("pose graph: no such factor");
    alive_[index] = false;
    activateAll();
  }
  void activateAll() {
    for (const auto& kv : values_) active_.insert(kv.first);
  }
  // whole-graph steps as rounds 1-5 took them (host_checks.cpp compares; a caller that wants them sets this once)
  void setWholeGraphSteps(bool on) { whole_graph_ = on; }
  size_t numActive() const { return active_.size(); }
  size_t numFactors() const { return (size_t)std::count(alive_.begin(), alive_.end(), true); }
  const Values& values() const { return values_; }

  // whitened residual of one factor at the current estimate; weight = Cauchy(1) IRLS weight
  void residual(const Factor& f, const SE3& Ta, const SE3& Tb, double r[6], double* weight) const {
    const SE3 pred = f.type == Factor::PRIOR ? Tb : Ta.inverse() * Tb;
    f.measurement.localCoordinates(pred, r);
    double n2 = 0;
    for (int i = 0; i < 6; ++i) { r[i] /= f.sigmas[i]; n2 += r[i] * r[i]; }
    *weight = f.cauchy ? 1.0 / (1.0 + n2) : 1.0;  // mEstimator::Cauchy(k = 1): w = k^2 / (k^2 + e^2)
  }

  // 0.5 * sum of (robustified) squared whitened residuals
  double error() const {
    double e = 0;
    for (size_t k = 0; k < factors_.size(); ++k) {
      if (!alive_[k]) continue;
      const Factor& f = factors_[k];
      double r[6], w;
      residual(f, poseA(f), values_.at(f.key_b), r, &w);
      double n2 = 0;
      for (double v : r) n2 += v * v;
      e += f.cauchy ? 0.5 * std::log1p(n2) : 0.5 * n2;  // Cauchy rho with k = 1
    }
    return e;
  }

  // `iterations` Gauss-Newton steps over the variables that still move; returns the largest update component of the last one
  double optimize(int iterations) {
    double last = 0;
    for (int it = 0; it < iterations; ++it) {
      if (whole_graph_) activateAll();
      if (active_.empty()) return 0;
      last = step();
      if (last < 1e-12) break;
    }
    return last;
  }

  static constexpr double kSettled = 1e-11;   // a variable whose update stayed below this leaves the active set
  static constexpr double kExpand = 1e-9;     // a variable whose update exceeded this activates its neighbours

 private:
  SE3 poseA(const Factor& f) const {
    if (f.type == Factor::PRIOR) return SE3();
    return f.fix_first_node ? f.fixed_a : values_.at(f.key_a);
  }

  double step() {
    using namespace detail;
    std::map<Key, int> index;
    std::vector<Key> keys(active_.begin(), active_.end());
    for (size_t i = 0; i < keys.size(); ++i) index[keys[i]] = (int)i;
    const int n = (int)keys.size();
    if (n == 0) return 0;
    std::vector<std::map<int, M6>> H(n);
    std::vector<V6> g(n, V6{});
    for (int i = 0; i < n; ++i) {
      M6 d{};
      for (int k = 0; k < 6; ++k) d[k * 6 + k] = 1e-9;  // keeps H positive definite without changing the solution
      H[i][i] = d;
    }
    const double h = 1e-6;
    // the factors of the active variables, each once, in index order (the order the whole-graph loop takes them in)
    std::vector<size_t> todo;
    for (const Key key : keys) {
      const auto it = adjacency_.find(key);
      if (it == adjacency_.end()) continue;
      for (size_t k : it->second)
        if (alive_[k]) todo.push_back(k);
    }
    std::sort(todo.begin(), todo.end());
    todo.erase(std::unique(todo.begin(), todo.end()), todo.end());
    for (size_t k : todo) {
      const Factor& f = factors_[k];
      const bool has_a = f.type != Factor::PRIOR && !f.fix_first_node;
      const SE3 Ta = poseA(f), Tb = values_.at(f.key_b);
      // a variable outside the active set is a constant of this step
      const auto fa = has_a ? index.find(f.key_a) : index.end();
      const auto fb = index.find(f.key_b);
      const int ia = fa != index.end() ? fa->second : -1, ib = fb != index.end() ? fb->second : -1;
      double r0[6], w;
      residual(f, Ta, Tb, r0, &w);
      // numerical Jacobians w.r.t. the retraction coordinates of each variable (central differences)
      double J[2][36];
      for (int v = 0; v < 2; ++v) {
        if ((v == 0 && ia < 0) || (v == 1 && ib < 0)) continue;
        for (int c = 0; c < 6; ++c) {
          double d[6] = {0, 0, 0, 0, 0, 0}, rp[6], rm[6], wu;
          d[c] = h;
          residual(f, v == 0 ? Ta.retract(d) : Ta, v == 1 ? Tb.retract(d) : Tb, rp, &wu);
          d[c] = -h;
          residual(f, v == 0 ? Ta.retract(d) : Ta, v == 1 ? Tb.retract(d) : Tb, rm, &wu);
          for (int i = 0; i < 6; ++i) J[v][i * 6 + c] = (rp[i] - rm[i]) / (2 * h);
        }
      }
      auto accumulate = [&](int vi, int vj, int bi, int bj) {  // H_bi,bj += w J_vi^T J_vj
        M6& dst = H[bi][bj];
        for (int a = 0; a < 6; ++a)
          for (int b = 0; b < 6; ++b) {
            double s = 0;
            for (int i = 0; i < 6; ++i) s += J[vi][i * 6 + a] * J[vj][i * 6 + b];
            dst[a * 6 + b] += w * s;
          }
      };
      auto gradient = [&](int vi, int bi) {  // g_bi -= w J_vi^T r
        for (int a = 0; a < 6; ++a) {
          double s = 0;
          for (int i = 0; i < 6; ++i) s += J[vi][i * 6 + a] * r0[i];
          g[bi][a] -= w * s;
        }
      };
      if (ib >= 0) {
        accumulate(1, 1, ib, ib);
        gradient(1, ib);
      }
      if (ia >= 0) {
        accumulate(0, 0, ia, ia);
        gradient(0, ia);
        if (ib >= 0) {
          accumulate(0, 1, ia, ib);
          accumulate(1, 0, ib, ia);
        }
      }
    }
    const std::vector<V6> dx = solveBlockSparse(H, g);
    double biggest = 0;
    std::vector<Key> wake;
    for (int i = 0; i < n; ++i) {
      SE3& T = values_.at(keys[i]);
      T = T.retract(dx[i].data());
      double mine = 0;
      for (double v : dx[i]) mine = std::max(mine, std::fabs(v));
      biggest = std::max(biggest, mine);
      if (mine < kSettled) active_.erase(keys[i]);
      else if (mine > kExpand) wake.push_back(keys[i]);
    }
    for (const Key key : wake) {   // a variable that still moves takes its neighbours along
      const auto it = adjacency_.find(key);
      if (it == adjacency_.end()) continue;
      for (size_t k : it->second) {
        if (!alive_[k]) continue;
        const Factor& f = factors_[k];
        if (f.type != Factor::PRIOR && !f.fix_first_node) active_.insert(f.key_a);
        active_.insert(f.key_b);
      }
    }
    return biggest;
  }

  Values values_;
  std::vector<Factor> factors_;
  std::vector<bool> alive_;
  std::map<Key, std::vector<size_t>> adjacency_;   // variable -> the factors it takes part in
  std::set<Key> active_;                           // variables the next step moves
  bool whole_graph_ = false;
};

}  // namespace laser_slam_amd
