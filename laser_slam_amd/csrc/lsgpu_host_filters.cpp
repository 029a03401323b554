// lsgpu_host_filters.cpp -- host-side modules of the chain (the reference runs them on the CPU
// inside PointMatcher::ICP::compute, laser_slam/src/laser_track.cpp:496):
//   RandomSamplingDataPointsFilter           laser_slam/configurations/icp_default.yaml:1-3
//   SamplingSurfaceNormalDataPointsFilter    laser_slam/configurations/icp_default.yaml:5-7
//   RigidTransformation check / correct      laser_slam/include/laser_slam/common.hpp:136-149
// GPU versions of the two filters are SURVEY.md §8f row N1/N3 ("next").
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <utility>
#include <vector>

#include "lsgpu_host_math.h"

#include "../../include/lsgpu_icp.h"
#include "lsgpu_box_normal.h"
#include "lsgpu_rand.h"

namespace {

struct SurfaceNormalBuilder {
  const float* xyz1;
  std::vector<int32_t> idx;
  std::vector<std::pair<float, int32_t>> scratch;
  int knn;
  float ratio;
  float* out_xyz1;
  float* out_nrm;
  int64_t n_out = 0;
  std::vector<unsigned char> keep;   // per ORIGINAL index: kept (the draws are taken in box-traversal order)
  std::vector<float> nrm_of;         // per original index: its box's normal

  // upstream sorts indicesToKeep ascending before it compacts the cloud in place: the output is in original index order
  void emit(int64_t n) {
    for (int64_t i = 0; i < n; ++i)
      if (keep[(size_t)i]) {
        const int64_t o = n_out++;
        std::memcpy(out_xyz1 + 4 * o, xyz1 + 4 * i, 16);
        std::memcpy(out_nrm + 3 * o, nrm_of.data() + 3 * i, 12);
      }
  }

  float coord(int32_t i, int d) const { return xyz1[4 * (int64_t)i + d]; }

  void fuse(int64_t first, int64_t last) {
    const int64_t cnt = last - first;
    if (cnt <= 0) return;
    float n[3];
    if (!lsgpu::boxnormal::box_normal((int)cnt, [&](int i, int d) { return coord(idx[first + i], d); }, n))
      return;  // degenerate box: dropped
    float draws[64];
    for (int64_t i = first; i < last; ++i) {  // samplingMethod 0: random subset keeps the box normal
      const int64_t k = (i - first) % 64;
      if (k == 0) lsgpu::DrawStream::global().take(-1, (size_t)std::min<int64_t>(64, last - i), draws);
      const float r = draws[k];
      if (r < ratio) {   // indicesToKeep.push_back(k); normals->col(k) = normal: compacted by original index in emit()
        const int64_t k_orig = idx[i];
        keep[(size_t)k_orig] = 1;
        for (int d = 0; d < 3; ++d) nrm_of[3 * (size_t)k_orig + d] = n[d];
      }
    }
  }

  void build(int64_t first, int64_t last, const float* minb, const float* maxb) {
    const int64_t count = last - first;
    if (count <= knn) { fuse(first, last); return; }
    int cut = 0;
    for (int d = 1; d < 3; ++d)
      if (maxb[d] - minb[d] > maxb[cut] - minb[cut]) cut = d;
    const int64_t right = count / 2, left = count - right;
    // std::nth_element leaves ties and the order inside each half unspecified; a stable sort fixes both,
    // so that this filter and the device filter (lsgpu_ssn.hip.h, stable radix sort) build the same boxes
    // in the same order and draw the same rand() numbers for the same points.
    // (sorted as (coordinate, index) pairs: the comparisons then touch contiguous memory only)
    scratch.resize((size_t)count);
    for (int64_t i = 0; i < count; ++i) scratch[(size_t)i] = {coord(idx[first + i], cut), idx[first + i]};
    std::stable_sort(scratch.begin(), scratch.end(),
                     [](const std::pair<float, int32_t>& a, const std::pair<float, int32_t>& b) { return a.first < b.first; });
    for (int64_t i = 0; i < count; ++i) idx[first + i] = scratch[(size_t)i].second;
    const float cutval = coord(idx[first + left], cut);
    float lmax[3] = {maxb[0], maxb[1], maxb[2]}, rmin[3] = {minb[0], minb[1], minb[2]};
    lmax[cut] = cutval; rmin[cut] = cutval;
    build(first, first + left, minb, lmax);
    build(first + left, last, rmin, maxb);
  }
};

float det3(const float* T) {
  auto m = [&](int r, int c) { return T[c * 4 + r]; };
  return m(0, 0) * (m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1)) -
         m(0, 1) * (m(1, 0) * m(2, 2) - m(1, 2) * m(2, 0)) +
         m(0, 2) * (m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0));
}

}  // namespace

extern "C" {

int64_t lsgpu_filter_random_sampling(int64_t n, float prob, int64_t seed, int64_t* keep_idx) {
  if (seed >= 0) lsgpu::DrawStream::global().take(seed, 0, nullptr);
  if (n <= 0 || !keep_idx) return 0;
  int64_t m = 0;
  float draws[1024];
  for (int64_t i0 = 0; i0 < n; i0 += 1024) {
    const int64_t k = std::min<int64_t>(1024, n - i0);
    lsgpu::DrawStream::global().take(-1, (size_t)k, draws);
    for (int64_t i = 0; i < k; ++i)
      if (draws[i] < prob) keep_idx[m++] = i0 + i;
  }
  return m;
}

int64_t lsgpu_filter_sampling_surface_normal(const float* xyz1, int64_t n, int knn, float ratio,
                                             int64_t seed, float* out_xyz1, float* out_normals) {
  if (n <= 0 || !xyz1 || !out_xyz1 || !out_normals || knn < 3) return 0;
  if (seed >= 0) lsgpu::DrawStream::global().take(seed, 0, nullptr);
  SurfaceNormalBuilder b;
  b.xyz1 = xyz1; b.knn = knn; b.ratio = ratio; b.out_xyz1 = out_xyz1; b.out_nrm = out_normals;
  b.idx.resize((size_t)n);
  std::iota(b.idx.begin(), b.idx.end(), 0);
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int64_t i = 0; i < n; ++i)
    for (int d = 0; d < 3; ++d) {
      mn[d] = std::min(mn[d], xyz1[4 * i + d]);
      mx[d] = std::max(mx[d], xyz1[4 * i + d]);
    }
  b.keep.assign((size_t)n, 0);
  b.nrm_of.resize(3 * (size_t)n);
  b.build(0, n, mn, mx);
  b.emit(n);
  return b.n_out;
}

int lsgpu_check_rigid(const float T[16]) { return std::fabs(1.0f - det3(T)) <= 0.001f; }

void lsgpu_correct_rigid(const float T[16], float out[16]) {
  float c[3][3];
  for (int k = 0; k < 3; ++k) {
    const float x = T[k * 4], y = T[k * 4 + 1], z = T[k * 4 + 2];
    const float n = std::sqrt(x * x + y * y + z * z);
    c[k][0] = x / n; c[k][1] = y / n; c[k][2] = z / n;
  }
  const float c0[3] = {c[1][1] * c[2][2] - c[1][2] * c[2][1], c[1][2] * c[2][0] - c[1][0] * c[2][2],
                       c[1][0] * c[2][1] - c[1][1] * c[2][0]};
  const float c1[3] = {c[2][1] * c0[2] - c[2][2] * c0[1], c[2][2] * c0[0] - c[2][0] * c0[2],
                       c[2][0] * c0[1] - c[2][1] * c0[0]};
  std::memcpy(out, T, 16 * sizeof(float));
  for (int r = 0; r < 3; ++r) { out[r] = c0[r]; out[4 + r] = c1[r]; out[8 + r] = c[2][r]; }
}

float lsgpu_rotation_distance(const float Ta[16], const float Tb[16]) {
  float qa[4], qb[4];
  lsgpu::hostmath::quat_from_rotation(Ta, qa);
  lsgpu::hostmath::quat_from_rotation(Tb, qb);
  return lsgpu::hostmath::angular_distance(qa, qb);
}

}  // extern "C"
