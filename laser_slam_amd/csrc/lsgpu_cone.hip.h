// lsgpu_cone.hip.h -- KDTreeMatcher::findClosests (knn 1, epsilon 0; laser_slam/configurations/icp_default.yaml:9-12,
// called once per iteration of icp_.compute, laser_slam/src/laser_track.cpp:496) for the SETTLED iterations of an align:
// the reference indexed by DIRECTION, every lane searching its own contiguous windows.
//
// Why.  k_knn_tile (lsgpu_knn.hip.h) evaluates, for all 64 lanes of a wave, the union of the 64-point voxel chunks any of
// its lanes needs: 200-235 candidates per query on the 1 M-point benchmark scan, while a searching query's ball holds 2
// ("Round 3").  A ball (q, R) lies inside the cone of half-angle asin(R / |q - O|) about q's direction seen
// from ANY fixed point O -- no assumption about the sensor.  So:
//   * the reference is sorted a second time by (row, column) of its direction about O = the origin of its own frame
//     (-mean after centring): row = bin of zeta = z / rho (sine of the elevation), column = bin of the pseudo-azimuth
//     p in [0, 4) (the "diamond angle" y / (|x| + |y|) unfolded over the quadrants: monotone in the azimuth, one division,
//     no trigonometry); `tab` maps (row, column) -> first position, `rowz` holds every row's zeta range;
//   * a searching lane turns its ball into a zeta range and, per row that range touches, a column window narrowed by the
//     chord of the cone at that row's elevation; the reference points of the window are one contiguous run of the SoA
//     copy, which the lane evaluates four at a time with the packed-pair arithmetic of the tile kernel (12 v_pk ops per
//     4 candidates).  For a spinning lidar seen from its own origin a row is a ring or empty, and the rows between the
//     rings are skipped on their zeta range alone: median 11 / mean 15 candidates per searching query on the benchmark
//     pair (devtools/sim_cone.py), a wave runs ~14 steps of 4 candidates where the broadcast search runs ~55.
// Exactness: every inequality below is an upper bound of the exact spherical relation (derivations inline), widened
// by margins two orders of magnitude above the float rounding of the quantities involved; every reference point
// within R of q is inside one of q's windows, points outside the ball that happen to be in a window are real reference
// points too (evaluating them cannot hurt).  Same contract as k_knn_tile towards the rest of the loop: warm start,
// keep / far skips, search `gap` beyond the current bound, lower bound on "every other point", smallest Morton index
// on exact ties (canonical_tie), share of the predicted / committed select.  Lanes the index cannot serve (|q - O| < 2R,
// a cone that reaches the polar axis, a window of more than kConeMaxWin points) search the voxel grid themselves
// (lane_ball_search) and are counted as stragglers: the host stops using this kernel for an align in which they are
// not rare.
#pragma once
#include "lsgpu_knn.hip.h"

namespace lsgpu {

constexpr int kConePad = 16;            // far points behind the direction-sorted arrays (group loads run to a multiple of 4)
constexpr uint32_t kConeMaxWin = 512;   // longest run (points) a lane takes from one row
constexpr int kConeRowBatch = 4;        // rows a lane looks up per round of table probes

struct ConeDev {
  float ox, oy, oz;        // O in the reference-mean frame
  float z0, rs;            // row = floor((zeta - z0) * rs), clamped to [0, rows)
  float cs;                // column = floor(p * cs), cs = cols / 4
  int rows, cols;
  const float* x; const float* y; const float* z;   // direction-sorted SoA copy of the reference (+ kConePad far points)
  const uint32_t* map;     // direction-sorted position -> index in the Morton-sorted reference (pts)
  const uint32_t* tab;     // rows * cols + 1: first position whose key is >= row * cols + column
  const float2* rowz;      // per row: {min, max} of zeta over its points; empty row: min > max
};

// direction of v = point - O: zeta = v.z / |v|, pseudo-azimuth pa in [0, 4), inv_rho = 1 / |v|, rxy = |v.xy|,
// inv_h = 1 / (|v.x| + |v.y|).  Hardware reciprocal / square root (1 ulp): the bins of the reference and the windows of
// the queries come from this one function, and the windows carry margins 20x larger than its rounding.
__device__ __forceinline__ void cone_dir(float vx, float vy, float vz, float& inv_rho, float& zeta, float& pa, float& rxy,
                                         float& inv_h) {
  const float r2 = __fmaf_rn(vy, vy, vx * vx);
  inv_rho = __builtin_amdgcn_rsqf(__fmaf_rn(vz, vz, r2));
  rxy = __builtin_amdgcn_sqrtf(r2);
  zeta = vz * inv_rho;
  inv_h = __builtin_amdgcn_rcpf(fabsf(vx) + fabsf(vy));
  const float t = vy * inv_h;
  pa = vx >= 0.f ? (vy >= 0.f ? t : 4.f + t) : 2.f - t;
}

__device__ __forceinline__ uint32_t cone_clampu(float t, int hi) {   // floor already applied; NaN -> 0
  return (uint32_t)fminf(fmaxf(t, 0.f), (float)hi);
}

// ---------------------------------------------------------------- build
// keys of the Morton-sorted, centred reference: (row, column) of every point's direction
__global__ __launch_bounds__(256) void k_cone_keys(const float4* __restrict__ pts, int64_t n, ConeDev c,
                                                   uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                   uint32_t* __restrict__ rowz_bits) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < 2 * (int64_t)c.rows) rowz_bits[i] = (i & 1) ? 0u : 0xFFFFFFFFu;   // {min, max} in ordered-integer form
  if (i >= n) return;
  const float4 p = pts[i];
  float inv_rho, zeta, pa, rxy, inv_h;
  cone_dir(p.x - c.ox, p.y - c.oy, p.z - c.oz, inv_rho, zeta, pa, rxy, inv_h);
  if (!(fabsf(zeta) <= 1.5f)) zeta = 0.f;   // the point IS O (0 * inf): any bin (a query that close to O does not use the index)
  if (!(pa >= 0.f && pa <= 4.f)) pa = 0.f;  // on the polar axis: any column (a cone that reaches the axis does not use the index)
  const uint32_t row = cone_clampu(floorf((zeta - c.z0) * c.rs), c.rows - 1);
  const uint32_t col = cone_clampu(floorf(pa * c.cs), c.cols - 1);
  keys[i] = (uint64_t)row * (uint32_t)c.cols + col;
  vals[i] = (uint32_t)i;
}

__device__ __forceinline__ uint32_t ord_of_float(float f) {   // order-preserving float -> uint32
  const uint32_t b = __float_as_uint(f);
  return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float float_of_ord(uint32_t o) {
  return __uint_as_float(o ^ ((o >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

// after the stable sort by key: SoA copy + position map, the (row, column) -> first position table (every thread fills
// the table entries between its predecessor's key and its own; long gaps -- empty rows -- by the whole wave), the zeta
// range of every row
__global__ __launch_bounds__(256) void k_cone_gather(const float4* __restrict__ pts, const uint32_t* __restrict__ perm,
                                                     const uint64_t* __restrict__ keys, int64_t n, ConeDev c,
                                                     float* __restrict__ x, float* __restrict__ y, float* __restrict__ z,
                                                     uint32_t* __restrict__ map, uint32_t* __restrict__ tab,
                                                     uint32_t* __restrict__ rowz_bits) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int64_t npad = ((n + 3) & ~(int64_t)3) + kConePad;
  const uint32_t nkeys = (uint32_t)c.rows * (uint32_t)c.cols;
  uint32_t gap_lo = 0, gap_hi = 0;   // this thread writes tab[gap_lo .. gap_hi) = its own position
  uint32_t pos = 0;
  bool valid = j < n;
  uint32_t row = 0xFFFFFFFFu;
  float zeta = 0.f;
  if (valid) {
    const uint32_t i = perm[j];
    const float4 p = pts[i];
    x[j] = p.x; y[j] = p.y; z[j] = p.z; map[j] = i;
    const uint32_t key = (uint32_t)keys[j];
    gap_lo = j > 0 ? (uint32_t)keys[j - 1] + 1u : 0u;
    gap_hi = key + 1u;
    pos = (uint32_t)j;
    row = key / (uint32_t)c.cols;
    float inv_rho, pa, rxy, inv_h;
    cone_dir(p.x - c.ox, p.y - c.oy, p.z - c.oz, inv_rho, zeta, pa, rxy, inv_h);
    if (!(fabsf(zeta) <= 1.5f)) zeta = 0.f;
  } else if (j < npad) {
    x[j] = kPadCoord; y[j] = kPadCoord; z[j] = kPadCoord; map[j] = 0u;
    if (j == n) { gap_lo = (uint32_t)keys[n - 1] + 1u; gap_hi = nkeys + 1u; pos = (uint32_t)n; }   // everything behind the last key
  }
  // ---- table: short gaps per thread, long ones by the wave
  const uint32_t glen = gap_hi - gap_lo;
  if (glen <= 16u) for (uint32_t k = gap_lo; k < gap_hi; ++k) tab[k] = pos;
  unsigned long long big = __ballot(glen > 16u);
  while (big) {
    const int src = __ffsll((long long)big) - 1;
    big &= big - 1;
    const uint32_t lo = rl_u(gap_lo, src), hi = rl_u(gap_hi, src), ps = rl_u(pos, src);
    for (uint32_t k = lo + (uint32_t)lane; k < hi; k += 64u) tab[k] = ps;
  }
  // ---- zeta range per row: one pair of atomics per wave where the wave is inside one row (almost always)
  const uint32_t r0 = rl_u(row, 0);
  const bool same = __ballot(valid && row == r0) == __ballot(valid) && __ballot(valid);
  if (same) {
    const float mn = wave_min(valid ? zeta : INFINITY), mx = wave_max(valid ? zeta : -INFINITY);
    if (lane == 0) { atomicMin(&rowz_bits[2u * r0], ord_of_float(mn)); atomicMax(&rowz_bits[2u * r0 + 1u], ord_of_float(mx)); }
  } else if (valid) {
    atomicMin(&rowz_bits[2u * row], ord_of_float(zeta));
    atomicMax(&rowz_bits[2u * row + 1u], ord_of_float(zeta));
  }
}

__global__ __launch_bounds__(256) void k_cone_rowz(const uint32_t* __restrict__ rowz_bits, int rows, float2* __restrict__ rowz) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  const uint32_t lo = rowz_bits[2 * r], hi = rowz_bits[2 * r + 1];
  rowz[r] = lo == 0xFFFFFFFFu ? make_float2(INFINITY, -INFINITY) : make_float2(float_of_ord(lo), float_of_ord(hi));
}

// ---------------------------------------------------------------- search
// One window of one lane: groups [g0, g0 + len) of four consecutive direction-sorted points.  All lanes run the wave's
// longest window; a lane past its own end sits the step out (no load is issued for it).
__device__ __forceinline__ void cone_eval_window(const ConeDev& c, uint32_t g0, uint32_t len, float qx, float qy, float qz,
                                                 float& best, float& sec, uint32_t& bgrp) {
  const uint32_t trip = wave_max_u32(len);
  const float4* __restrict__ x4 = reinterpret_cast<const float4*>(c.x);
  const float4* __restrict__ y4 = reinterpret_cast<const float4*>(c.y);
  const float4* __restrict__ z4 = reinterpret_cast<const float4*>(c.z);
  const f32x2 q2x = {qx, qx}, q2y = {qy, qy}, q2z = {qz, qz};
#ifdef LSGPU_CONE_PIPE
  // the next group's three loads are in flight while this one is evaluated
  float4 X = make_float4(0.f, 0.f, 0.f, 0.f), Y = X, Z = X;
  if (len) { X = x4[g0]; Y = y4[g0]; Z = z4[g0]; }
  for (uint32_t t = 0; t < trip; ++t) {
    float4 Xn = X, Yn = Y, Zn = Z;
    if (t + 1u < len) { Xn = x4[g0 + t + 1u]; Yn = y4[g0 + t + 1u]; Zn = z4[g0 + t + 1u]; }
    if (t < len) {
      const f32x2 d0 = dist2_pair(q2x, q2y, q2z, f32x2{X.x, X.y}, f32x2{Y.x, Y.y}, f32x2{Z.x, Z.y});
      const f32x2 d1 = dist2_pair(q2x, q2y, q2z, f32x2{X.z, X.w}, f32x2{Y.z, Y.w}, f32x2{Z.z, Z.w});
      const float m4 = fminf(fminf(fminf(d0.x, d0.y), d1.x), d1.y);
      sec = __builtin_amdgcn_fmed3f(best, m4, sec);
      if (m4 < best) { best = m4; bgrp = g0 + t; }
    }
    X = Xn; Y = Yn; Z = Zn;
  }
#else
  for (uint32_t t = 0; t < trip; ++t) {
    if (t < len) {
      const uint32_t g = g0 + t;
      const float4 X = x4[g], Y = y4[g], Z = z4[g];
      const f32x2 d0 = dist2_pair(q2x, q2y, q2z, f32x2{X.x, X.y}, f32x2{Y.x, Y.y}, f32x2{Z.x, Z.y});
      const f32x2 d1 = dist2_pair(q2x, q2y, q2z, f32x2{X.z, X.w}, f32x2{Y.z, Y.w}, f32x2{Z.z, Z.w});
      const float m4 = fminf(fminf(fminf(d0.x, d0.y), d1.x), d1.y);
      sec = __builtin_amdgcn_fmed3f(best, m4, sec);   // second smallest group minimum (best <= sec always)
      if (m4 < best) { best = m4; bgrp = g; }
    }
  }
#endif
}

#ifndef LSGPU_CONE_OCC
#define LSGPU_CONE_OCC 8
#endif
__global__ __launch_bounds__(64, LSGPU_CONE_OCC) void k_knn_cone(KnnArgs a, ConeDev c) {
  const int lane = threadIdx.x;
  const uint32_t tile = blockIdx.x;
  const int j = (int)(tile * 64u) + lane;
  const bool act = j < a.nq;
  float4 rraw = make_float4(0.f, 0.f, 0.f, 0.f), mp = rraw;
  float lb_in = 0.f;
  if (act) {
    rraw = a.rdq[j];
    mp = a.prev[j];
    lb_in = a.lb[j];
  }
  const int id_in = __float_as_int(mp.w);
  Mat34 T; float cap2;
  if (!iter_params(a.st, a.T, a.cap2, 1, T, cap2)) return;
  const float cap2s = cap2 * kCapSearchMargin2;
  const float gap = a.gap;
  // ---- the query, its warm-start distance, the keep / far skips: as k_knn_tile
  float qx = 0.f, qy = 0.f, qz = 0.f, ub = 0.f, lbn = 0.f;
  bool skip = false;
  if (act) {
    const float3 q = xform(T, rraw.x, rraw.y, rraw.z);
    qx = q.x; qy = q.y; qz = q.z;
    ub = dist2(qx - mp.x, qy - mp.y, qz - mp.z);
    Mat34 To;
#pragma unroll
    for (int i = 0; i < 12; ++i) To.m[i] = a.st->T_rows_prev[i];
    const float3 qo = xform(To, rraw.x, rraw.y, rraw.z);
    const float ddx = qx - qo.x, ddy = qy - qo.y, ddz = qz - qo.z;
    const float delta = sqrtf(ddx * ddx + ddy * ddy + ddz * ddz) * (1.0f + 1e-5f) + 1e-7f;
    lbn = fmaxf(lb_in * (1.0f - 1e-6f) - delta, 0.f);
    const float lb2 = lbn * lbn;
    const bool keep = ub * (1.0f + 1e-5f) < lb2;
    const bool far = fminf(ub, lb2) > cap2 * (1.0f + 1e-5f);
    skip = keep || far;
  }
  const float lim0 = prune_lim(ub, gap, cap2s);                 // squared search radius: every point inside is evaluated
  const float R = sqrtf(lim0) * (1.0f + 1e-5f) + 1e-7f;
  const bool ing = act && !skip;
  bool fb = false;                 // this lane searches the voxel grid instead
  float best = INFINITY, sec = INFINITY;
  uint32_t bgrp = 0xFFFFFFFFu;     // group of four direction-sorted points that holds the evaluated minimum
  if (__ballot(ing)) {
    // ---- the lane's cone.  sin(alpha) = R / rho.  A point within R of q is seen from O under an angle <= alpha from q.
    float inv_rho, zeta, pa, rxy, inv_h;
    cone_dir(qx - c.ox, qy - c.oy, qz - c.oz, inv_rho, zeta, pa, rxy, inv_h);
    const float s = R * inv_rho * (1.0f + 1e-5f) + 4e-6f;      // (margin: rounding of q - O and of the points' own directions)
    const float ce = rxy * inv_rho;                             // cos(elevation of q)
    bool cone = ing && s <= 0.5f && ce > 0.f && ce <= 1.5f;     // (NaN-safe: |q - O| == 0 fails)
    const float alpha = s * (1.0f + 0.2f * s * s) + 2e-6f;      // >= asin(s) for s <= 0.5, + rounding of the zetas
    // zeta of such a point: |sin(e_p) - sin(e_q)| <= |sin e_q| (1 - cos alpha) + cos e_q sin alpha <= |zeta| s^2 + ce s
    const float dz = __fmaf_rn(ce, s, fabsf(zeta) * s * s) + 2e-6f;
    const float gq = (rxy * inv_h) * (rxy * inv_h) * (1.0f + 1e-6f);   // d(pa) / d(azimuth) at q = rho_xy^2 / (|x| + |y|)^2, in [1/2, 1]
    const uint32_t r_lo = cone_clampu(floorf((zeta - dz - c.z0) * c.rs), c.rows - 1);
    const uint32_t r_hi = cone_clampu(floorf((zeta + dz - c.z0) * c.rs), c.rows - 1);
    fb = ing && !cone;
    const uint32_t nrows = cone ? r_hi - r_lo + 1u : 0u;
    const uint32_t maxrows = wave_max_u32(nrows);
    for (uint32_t rb = 0; rb < maxrows; rb += (uint32_t)kConeRowBatch) {
      // ---- per row of the batch: is the row inside the cone's zeta range, and which columns
      int clo[kConeRowBatch], chi[kConeRowBatch];
      uint32_t rowk[kConeRowBatch];
      bool in[kConeRowBatch];
      float2 rz[kConeRowBatch];
#pragma unroll
      for (int i = 0; i < kConeRowBatch; ++i) {
        rowk[i] = r_lo + rb + (uint32_t)i;
        in[i] = cone && rowk[i] <= r_hi;
        rz[i] = make_float2(INFINITY, -INFINITY);
        if (in[i]) rz[i] = c.rowz[rowk[i]];
      }
      bool wraps = false;
#pragma unroll
      for (int i = 0; i < kConeRowBatch; ++i) {
        clo[i] = 0; chi[i] = -1;
        // distance in zeta between q and the row's points: a LOWER bound of their difference in elevation
        // (|sin a - sin b| <= |a - b|)
        const float dzr = fmaxf(fmaxf(fmaxf(rz[i].x - zeta, zeta - rz[i].y), 0.f) - 1e-6f, 0.f);
        in[i] = in[i] && rz[i].x <= rz[i].y && dzr <= alpha;
        // azimuth: cos(theta) = cos(de) - cos(e_q) cos(e_p) (1 - cos(da)), theta <= alpha  =>
        //   sin^2(da / 2) <= (cos(de) - cos(alpha)) / (2 ce ce_p) <= (alpha^2 - de^2) / (4 ce ce_p)
        const float zm = fmaxf(fabsf(rz[i].x), fabsf(rz[i].y));
        const float cep = sqrtf(fmaxf(1.f - zm * zm, 0.f)) * (1.0f - 1e-5f);   // smallest cos(elevation) in the row
        const float u2 = __fmaf_rn(alpha, alpha, -dzr * dzr) / (4.f * ce * cep);
        const bool polar = in[i] && !(u2 <= 0.25f);            // the cone reaches (or nears) the polar axis at this row
        const float u = sqrtf(fmaxf(u2, 0.f));
        const float da = 2.f * u * (1.0f + 0.2f * u * u);       // >= 2 asin(u)
        // pseudo-azimuth: |pa' - gq| <= 2 sqrt 2 |da| (pa' is Lipschitz), so |d pa| <= gq da + 1.42 da^2
        const float dp = __fmaf_rn(gq, da, 1.42f * da * da) + 4e-6f;
        if (in[i] && !polar) {
          clo[i] = (int)floorf((pa - dp) * c.cs);
          chi[i] = (int)floorf((pa + dp) * c.cs);
        }
        const bool wide = in[i] && !polar && chi[i] - clo[i] >= (c.cols >> 2);
        if (polar || wide) { fb = true; cone = false; }
        in[i] = in[i] && !polar && !wide;
        wraps = wraps || (in[i] && (clo[i] < 0 || chi[i] >= c.cols));
      }
      // ---- pass 0: the part of every window inside [0, cols); pass 1: the wrapped part of the windows that have one
      for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1 && !__ballot(wraps && cone)) break;
        uint32_t st[kConeRowBatch], en[kConeRowBatch];
#pragma unroll
        for (int i = 0; i < kConeRowBatch; ++i) {
          st[i] = 0u; en[i] = 0u;
          int a0, a1;
          bool w;
          if (pass == 0) {
            a0 = max(clo[i], 0); a1 = min(chi[i], c.cols - 1); w = in[i] && cone;
          } else {
            const bool wl = clo[i] < 0, wh = chi[i] >= c.cols;
            a0 = wl ? clo[i] + c.cols : 0; a1 = wl ? c.cols - 1 : chi[i] - c.cols; w = in[i] && cone && (wl || wh);
          }
          if (w) {
            const uint32_t base = rowk[i] * (uint32_t)c.cols;
            st[i] = c.tab[base + (uint32_t)a0];
            en[i] = c.tab[base + (uint32_t)a1 + 1u];
          }
        }
#pragma unroll
        for (int i = 0; i < kConeRowBatch; ++i) {
          if (en[i] - st[i] > kConeMaxWin) { fb = true; cone = false; }
        }
#pragma unroll
        for (int i = 0; i < kConeRowBatch; ++i) {
          const bool w = cone && en[i] > st[i];
          const uint32_t g0 = st[i] >> 2, len = w ? ((en[i] + 3u) >> 2) - g0 : 0u;
          if (__ballot(len != 0u)) cone_eval_window(c, g0, len, qx, qy, qz, best, sec, bgrp);
        }
      }
    }
  }
  // ---- lanes the index could not serve: the voxel grid (same search as a spread wave's lanes in k_knn_tile)
  int bi = id_in;
  const unsigned long long fbm = __ballot(fb);
  if (fbm) {
    if (fb) {
      best = ub;
      lane_ball_search(a, cap2s, qx, qy, qz, best, bi);
      if (bi != id_in) { const float4 p = a.pts[bi]; mp = make_float4(p.x, p.y, p.z, __int_as_float(bi)); }
    }
    if (lane == 0) atomicAdd(a.strag_count, (uint32_t)__popcll(fbm));   // (counted, not listed: they are done)
  }
  if (act) {
    float nb;  // new lower bound on the distance to every point other than the (new) match
    if (fb) {
      nb = best <= cap2s ? sqrtf(best) * (1.0f - 1e-6f) : fmaxf(lbn, sqrtf(cap2s) * (1.0f - 1e-5f));
    } else if (!ing) {
      best = ub;   // keep / far
      nb = lbn;
    } else if (bgrp != 0xFFFFFFFFu && best <= ub) {
      // the evaluated minimum (the warm-start point itself unless something beat it): which point of its group, and
      // the runner-up inside the group (the group minima of all other groups are in `sec` already)
      const float4 X = reinterpret_cast<const float4*>(c.x)[bgrp], Y = reinterpret_cast<const float4*>(c.y)[bgrp];
      const float4 Z = reinterpret_cast<const float4*>(c.z)[bgrp];
      const uint4 M = reinterpret_cast<const uint4*>(c.map)[bgrp];
      const float e0 = dist2(qx - X.x, qy - Y.x, qz - Z.x), e1 = dist2(qx - X.y, qy - Y.y, qz - Z.y);
      const float e2 = dist2(qx - X.z, qy - Y.z, qz - Z.z), e3 = dist2(qx - X.w, qy - Y.w, qz - Z.w);
      if (e3 == best) mp = make_float4(X.w, Y.w, Z.w, __uint_as_float(M.w));
      if (e2 == best) mp = make_float4(X.z, Y.z, Z.z, __uint_as_float(M.z));
      if (e1 == best) mp = make_float4(X.y, Y.y, Z.y, __uint_as_float(M.y));
      if (e0 == best) mp = make_float4(X.x, Y.x, Z.x, __uint_as_float(M.x));
      const float s4 = fminf(fmaxf(fminf(e0, e1), fminf(e2, e3)), fminf(fmaxf(e0, e1), fmaxf(e2, e3)));
      sec = fminf(sec, s4);
      float others = fminf(sec, lim0);   // every point that was not evaluated lies beyond the search radius
      bool same = __float_as_int(mp.w) == id_in;
      if (sec == best || (!same && best == ub)) {  // a second point at exactly this distance: smallest Morton index
        mp = canonical_tie(a, qx, qy, qz, best, mp);
        same = __float_as_int(mp.w) == id_in;
      }
      if (!same) others = fminf(others, ub);  // (covers a warm-start point outside the windows)
      nb = sqrtf(others) * (1.0f - 1e-5f);
      if (same) nb = fmaxf(nb, lbn);
    } else {  // the warm-start point lies beyond the cap and nothing closer exists: the match stands
      nb = fmaxf(sqrtf(fminf(best, lim0)) * (1.0f - 1e-5f), lbn);
      best = ub;
    }
    if (a.write_all || __float_as_int(mp.w) != id_in) {
      a.ids[j] = __float_as_int(mp.w);
      a.prev[j] = mp;
    }
    a.d2[j] = best;
    a.lb[j] = nb;
  }
  if (a.sel_below && (a.st->sel_mode || a.sel_force)) {   // first two passes of the trimmed-distance select (as k_knn_tile)
    const uint32_t bits = __float_as_uint(best), top = bits >> 20, b1 = a.st->sel_bin1;
    const unsigned long long below = __ballot(act && top < b1);
    if (act && top == b1) sel_count_inside(a, bits);
    if (lane == 0 && below) atomicAdd(&a.sel_below[(tile & (kSelBelowSlots - 1)) * kSelBelowStride], (uint32_t)__popcll(below));
  }
}

}  // namespace lsgpu
