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

// float4s per group of four points: {x0..x3}{y0..y3}{z0..z3}, 48 bytes, + a separate position map.  (Round 6 measured 64-byte
// groups that carry the Morton indices as a fourth quarter -- one aligned line per group, no map: the settled launch 30.2 ->
// 31.2-31.7 us, the probing one 108 -> 121-123: a third more index bytes for the caches, and the map read never was a dependent
// read -- it goes out with the re-read of the winning group.  rocprofv3 averages over 476 launches, devtools/prof_variants.sh.)
constexpr int kConeGF4 = 3;
constexpr int kConePad = 16;            // far points behind the direction-sorted arrays (group loads run to a multiple of 4)
constexpr uint32_t kConeMaxWin = 512;   // longest run (points) a lane takes from one row

struct ConeDev {
  float ox, oy, oz;        // O in the reference-mean frame
  float z0, rs;            // row = floor((zeta - z0) * rs), clamped to [0, rows)
  float cs;                // column = floor(p * cs), cs = cols / 4
  int rows, cols;
  const float4* soa;       // direction-sorted copy of the reference in groups of four points: group g = {x0..x3}{y0..y3}{z0..z3}
                           // (48 bytes, one cache line for all three loads of an evaluation step); + kConePad far points
  const uint32_t* map;     // direction-sorted position -> index in the Morton-sorted reference (pts)
  const uint32_t* tab;     // rows * cols + 1: first position whose key is >= row * cols + column
  const float4* rowz;      // per row: {min zeta, max zeta, 1 / (4 min cos(elevation)), -} over its points; empty row: min > max
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
                                                   uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
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

// after the stable sort by key: SoA copy + position map, the (row, column) -> first position table (every thread fills
// the table entries between its predecessor's key and its own; long gaps -- empty rows -- by the whole wave).  No atomics:
// the zeta range of every row and the number of occupied bins are k_cone_rows' (a row is ~8 k consecutive points, i.e. 128
// consecutive waves: their two atomics per wave on the row's two words serialised in the L2 and made this kernel 190 us
// long -- 16 k waves waiting 15 us each for 50 MB of traffic)
__global__ __launch_bounds__(256) void k_cone_gather(const float4* __restrict__ pts, const uint32_t* __restrict__ perm,
                                                     const uint64_t* __restrict__ keys, int64_t n, ConeDev c,
                                                     float* __restrict__ soa, uint32_t* __restrict__ map,
                                                     uint32_t* __restrict__ tab) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int64_t npad = ((n + 3) & ~(int64_t)3) + kConePad;
  const uint32_t nkeys = (uint32_t)c.rows * (uint32_t)c.cols;
  uint32_t gap_lo = 0, gap_hi = 0;   // this thread writes tab[gap_lo .. gap_hi) = its own position
  uint32_t pos = 0;
  const bool valid = j < n;
  if (valid) {
    const uint32_t i = perm[j];
    const float4 p = pts[i];
    float* g = soa + 4 * kConeGF4 * (j >> 2) + (j & 3);
    g[0] = p.x; g[4] = p.y; g[8] = p.z; map[j] = i;
    const uint32_t key = (uint32_t)keys[j];
    gap_lo = j > 0 ? (uint32_t)keys[j - 1] + 1u : 0u;
    gap_hi = key + 1u;
    pos = (uint32_t)j;
  } else if (j < npad) {
    float* g = soa + 4 * kConeGF4 * (j >> 2) + (j & 3);
    g[0] = kPadCoord; g[4] = kPadCoord; g[8] = kPadCoord; map[j] = 0u;
    if (j == n) { gap_lo = (uint32_t)keys[n - 1] + 1u; gap_hi = nkeys + 1u; pos = (uint32_t)n; }   // everything behind the last key
  }
  // ---- table: a thread whose key follows its predecessor's closely writes the few entries in between itself; longer
  // gaps -- sparse directions, empty rows -- by the whole wave
  const uint32_t glen = gap_hi - gap_lo;
  if (glen >= 1u && glen <= 6u) {   // (half of the bins of a 1 M-point scan are empty: most gaps are two or three entries)
    tab[gap_lo] = pos;
    for (uint32_t k = 1u; k < glen; ++k) tab[gap_lo + k] = pos;
  }
  unsigned long long big = __ballot(glen > 6u);
  while (big) {
    const int src = __ffsll((long long)big) - 1;
    big &= big - 1;
    const uint32_t lo = rl_u(gap_lo, src), hi = rl_u(gap_hi, src), ps = rl_u(pos, src);
    for (uint32_t k = lo + (uint32_t)lane; k < hi; k += 64u) tab[k] = ps;
  }
}

// One workgroup per row, behind k_cone_gather: the zeta range of the row's points (a contiguous run of the sorted copy),
// 1 / (4 min cos(elevation)), and the row's occupied bins (the host decides from points per occupied bin whether this
// index or the voxel grid serves the settled searches: a local map of many scans is several times denser in direction
// than one scan) -- one add per row.
__global__ __launch_bounds__(256) void k_cone_rows(ConeDev c, float4* __restrict__ rowz, uint32_t* __restrict__ occupied) {
  __shared__ float mn_sh[4], mx_sh[4];
  __shared__ uint32_t cnt_sh[4];
  const uint32_t row = blockIdx.x, base = row * (uint32_t)c.cols;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t st = c.tab[base], en = c.tab[base + (uint32_t)c.cols];
  const float* __restrict__ soa = reinterpret_cast<const float*>(c.soa);
  float mn = INFINITY, mx = -INFINITY;
  for (uint32_t p = st + threadIdx.x; p < en; p += 256u) {
    const float* g = soa + (uint32_t)(4 * kConeGF4) * (size_t)(p >> 2) + (p & 3u);
    float inv_rho, zeta, pa, rxy, inv_h;
    cone_dir(g[0] - c.ox, g[4] - c.oy, g[8] - c.oz, inv_rho, zeta, pa, rxy, inv_h);
    if (!(fabsf(zeta) <= 1.5f)) zeta = 0.f;   // (as k_cone_keys)
    mn = fminf(mn, zeta); mx = fmaxf(mx, zeta);
  }
  uint32_t cnt = 0;
  for (uint32_t b = threadIdx.x; b < (uint32_t)c.cols; b += 256u) cnt += c.tab[base + b + 1u] > c.tab[base + b] ? 1u : 0u;
  mn = wave_min(mn); mx = wave_max(mx); cnt = wave_sum_u32(cnt);
  if (lane == 0) { mn_sh[w] = mn; mx_sh[w] = mx; cnt_sh[w] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    mn = fminf(fminf(mn_sh[0], mn_sh[1]), fminf(mn_sh[2], mn_sh[3]));
    mx = fmaxf(fmaxf(mx_sh[0], mx_sh[1]), fmaxf(mx_sh[2], mx_sh[3]));
    float4 o = make_float4(INFINITY, -INFINITY, INFINITY, 0.f);
    if (en > st) {
      o.x = mn; o.y = mx;
      const float zm = fmaxf(fabsf(mn), fabsf(mx));
      const float cep = sqrtf(fmaxf(1.f - zm * zm, 0.f)) * (1.0f - 1e-5f);   // smallest cos(elevation) in the row, rounded down
      o.z = 1.0f / (4.f * cep) * (1.0f + 1e-6f);                              // (a row that touches the polar axis: inf)
    }
    rowz[row] = o;
    const uint32_t total = cnt_sh[0] + cnt_sh[1] + cnt_sh[2] + cnt_sh[3];
    if (total) atomicAdd(occupied, total);
  }
}

// ---------------------------------------------------------------- search
#ifdef LSGPU_KNN_STATS
// stats build (devtools/cone_phases.py): one 16-word record per wave and ICP iteration of k_knn_cone (no atomics: sixteen
// thousand waves adding to the same words made a 30 us launch 1.5 ms long)
__device__ uint32_t* g_cone_rec;   // [kConeRecIters][ntiles][16], set by lsgpu_dev_cone_phases
constexpr int kConeRecIters = 48;
struct ConeStat { long long t_eval = 0, t_rows = 0; uint32_t steps = 0, items = 0, batches = 0; };
#define CONE_STAT_ARG , ConeStat& cs
#define CONE_STAT_PASS , cs
#else
#define CONE_STAT_ARG
#define CONE_STAT_PASS
#endif

// One window of one lane, evaluated by that lane alone: groups [g0, g0 + len) of four consecutive direction-sorted points
// (the probe of the first search through the index; the search proper shares its windows out over the wave, below).  All
// lanes run until the wave's longest window is through; a lane past its own end sits the step out.
__device__ __forceinline__ void cone_eval_window(const ConeDev& c, uint32_t g0, uint32_t len, float qx, float qy, float qz,
                                                 float& best, float& sec) {
  const f32x2 q2x = {qx, qx}, q2y = {qy, qy}, q2z = {qz, qz};
  // two groups per step, all six loads issued before the first is waited for: one memory round trip for eight candidates
  for (uint32_t t = 0; __ballot(t < len); t += 2u) {
    if (t < len) {
      const float4* __restrict__ p = c.soa + (uint32_t)kConeGF4 * (size_t)(g0 + t);
      const float4* __restrict__ p1 = t + 1u < len ? p + kConeGF4 : p;   // (no second group: the first once more, it changes nothing)
      const float4 X = p[0], Y = p[1], Z = p[2];
      const float4 X1 = p1[0], Y1 = p1[1], Z1 = p1[2];
      const f32x2 d0 = dist2_pair(q2x, q2y, q2z, f32x2{X.x, X.y}, f32x2{Y.x, Y.y}, f32x2{Z.x, Z.y});
      const f32x2 d1 = dist2_pair(q2x, q2y, q2z, f32x2{X.z, X.w}, f32x2{Y.z, Y.w}, f32x2{Z.z, Z.w});
      const f32x2 e0 = dist2_pair(q2x, q2y, q2z, f32x2{X1.x, X1.y}, f32x2{Y1.x, Y1.y}, f32x2{Z1.x, Z1.y});
      const f32x2 e1 = dist2_pair(q2x, q2y, q2z, f32x2{X1.z, X1.w}, f32x2{Y1.z, Y1.w}, f32x2{Z1.z, Z1.w});
      const float m4 = fminf(fminf(fminf(d0.x, d0.y), d1.x), d1.y);
      const float n4 = fminf(fminf(fminf(e0.x, e0.y), e1.x), e1.y);
      sec = __builtin_amdgcn_fmed3f(best, m4, sec);
      best = fminf(best, m4);
      sec = __builtin_amdgcn_fmed3f(best, n4, sec);
      best = fminf(best, n4);
    }
  }
}

// ---- the kernel.  A workgroup takes WAVES consecutive tiles of 64 queries.  Phase 1, every wave for its own tile: the query,
// its warm-start distance, the keep / far skips (as k_knn_tile); lanes that do not search are finished here.  The
// searching lanes of all WAVES tiles are then packed through LDS (in query order) and phase 2 -- cone, rows, windows,
// evaluation, results -- runs on full waves of searching lanes only: what a wave pays per row and per window it pays
// for 64 lanes that need it, and the waves left without lanes exit.  Per-lane windows do not care who the neighbours in
// the wave are.
// Round 6, from per-wave records of where the cycles go (devtools/cone_phases.py, profiles/r06_cone_phases_*.txt: a
// searching wave lived 29 k cycles in a 29 us launch whose vector units idled 70 % of the time -- a chain of ~15
// dependent memory round trips of 1.6-2.5 k cycles each, and the launch ends with its slowest wave):
//   * the barriers of the pack wait for the LDS only -- __syncthreads() also waited for the stores and select atomics of
//     the lanes that were done (4.3 k -> 2 k cycles between the loads' arrival and the pack);
//   * the loop state's words arrive in one batch of scalar loads; the rows' zeta ranges travel to LDS beside the queries,
//     with one occupancy bit per row: empty rows (between the rings of a spinning lidar) cost a shift, not a load;
//   * the table reads of the NEXT row are in flight while the current row's windows are evaluated (a table round trip per
//     row was 3.5 k cycles);
//   * both groups of an evaluation step are loaded before the first is waited for (the conditional second load used to
//     follow the first one's wait: two round trips per step);
//   (64-byte groups carrying the Morton indices, instead of 48-byte groups + a position map, were measured slower and are not
//   used: kConeGF4 above.)
// Built and measured on the way, and dropped (same results, slower launches): every lane's windows cut into groups and
// shared out over the wave through an LDS list with per-owner LDS minima (balanced lanes, three round trips per wave --
// but 27 KB of LDS per workgroup held until its slowest wave ends: 35-38 us per late launch against 25-28); the union of
// a row's windows staged into LDS with LDS-DMA and every lane stepping through its window there (coalesced loads, a few
// dozen cache lines per wave instead of hundreds -- but 64 lanes gathering 48 bytes each per step make the LDS the
// bottleneck: 800 cycles per step, 42-50 us per late launch).
#ifndef LSGPU_CONE_WAVES
#define LSGPU_CONE_WAVES 4
#endif
#ifndef LSGPU_CONE_OCC
#define LSGPU_CONE_OCC 8   // waves per SIMD the register budget is cut for
#endif
constexpr int kConeRowSlots = 4;           // occupied rows a wave looks up per table round trip

struct ConeRec { float qx, qy, qz, ub, lbn; int id_in, j; uint32_t pad; };   // 32 B: one searching query on its way to phase 2

__device__ __forceinline__ void cone_barrier_lds() {   // workgroup barrier that waits for the LDS only: the stores and
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // atomics of phase 1 need not have landed
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) { return ~wave_max_u32(~v); }

template <int WAVES, bool PROBE = false>
__global__ __launch_bounds__(WAVES * 64, LSGPU_CONE_OCC) void k_knn_cone(KnnArgs a, ConeDev c) {
  __shared__ ConeRec rec[WAVES * 64];          // the searching queries of the block's tiles, packed
  __shared__ unsigned long long ne_mask[16];   // occupied rows, one bit each (<= 1024 rows)
  __shared__ uint32_t wcount[WAVES], wbelow[WAVES];
  extern __shared__ float4 rowz_sh[];          // the rows' records {min zeta, max zeta, 1 / (4 min cos e), -}
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#ifdef LSGPU_KNN_STATS
  const long long ts0 = clock64();
  const uint32_t wall0 = (uint32_t)wall_clock64();   // (100 MHz, one counter for the whole device: the launch's timeline)
  long long t_rows = 0, t_stage = 0, t_eval = 0;
  uint32_t n_chunks = 0, n_steps = 0, n_groups = 0, n_rowslots = 0;
#endif
  // (tiles go to the XCDs round robin, in dispatch order.  One contiguous eighth of the tiles per XCD -- whose L2 would then
  // hold just the stretch of the index its tiles read -- was measured slower, 39-55 us per launch against 30-42: the tiles
  // of near range cost several times those of the far field, and the launch ends with the slowest XCD.)
  const uint32_t tile = blockIdx.x * (uint32_t)WAVES + (uint32_t)w;
  int j = (int)(tile * 64u) + lane;
  const bool act = j < a.nq;
  float4 rraw = make_float4(0.f, 0.f, 0.f, 0.f), mp = rraw;
  float lb_in = 0.f;
  if (act) {
    rraw = a.rdq[j];
    mp = a.prev[j];
    lb_in = a.lb[j];
  }
  // the rows' records and their occupancy mask go to LDS beside those loads
  const uint32_t nrows = (uint32_t)c.rows;
  for (uint32_t r0 = 0; r0 < nrows; r0 += (uint32_t)(WAVES * 64)) {
    const uint32_t r = r0 + threadIdx.x;
    float4 rz = make_float4(INFINITY, -INFINITY, INFINITY, 0.f);
    if (r < nrows) { rz = c.rowz[r]; rowz_sh[r] = rz; }
    const unsigned long long occ = __ballot(rz.x <= rz.y);
    if (lane == 0 && (r0 >> 6) + (uint32_t)w < 16u) ne_mask[(r0 >> 6) + (uint32_t)w] = occ;
  }
  int id_in = __float_as_int(mp.w);
  // the loop state: everything this launch reads of it, in one go
  const IcpState* __restrict__ st = a.st;
  if (st->done) return;   // (the same answer in every wave of the block)
  Mat34 T, To;
#pragma unroll
  for (int i = 0; i < 12; ++i) { T.m[i] = st->T_rows[i]; To.m[i] = st->T_rows_prev[i]; }
  const float cap2 = st->cap2;
  const bool sel_on = a.sel_below && (st->sel_mode || a.sel_force);
  const uint32_t sel_lo = sel_on ? st->sel_lo : 0u, sel_span = sel_on ? st->sel_span : 0u;
  const uint32_t sel_sh = sel_on ? (uint32_t)st->sel_shift : 0u;
  const uint32_t sel_b2 = sel_on ? st->sel_bin2 : 0u;
  const float cap2s = cap2 * kCapSearchMargin2;
  const float gap = a.gap;
#ifdef LSGPU_KNN_STATS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long ts1 = clock64();
  uint32_t* const dbg = g_cone_rec ? g_cone_rec + ((size_t)min(max(st->iter, 0), kConeRecIters - 1) * (size_t)a.ntiles + tile) * 16u : nullptr;
#endif
  // one final distance's share of the predicted / committed select (as sel_count_inside, the state's words already here)
  auto sel_inside = [&](uint32_t bits) {
    const uint32_t bin2 = (bits - sel_lo) >> sel_sh;
    atomicAdd(&a.sel_hist2[bin2], 1u);
    if (a.sel_hist3w) {
      const uint32_t d = bin2 - sel_b2 + (uint32_t)kSelWinHalf;
      if (d < (uint32_t)kSelWinRows) atomicAdd(&a.sel_hist3w[d * 512u + (bits & 0x1FFu)], 1u);
    }
  };
  // ---- phase 1
  float qx = 0.f, qy = 0.f, qz = 0.f, ub = 0.f, lbn = 0.f;
  bool skip = false;
  if (act) {
    const float3 q = xform(T, rraw.x, rraw.y, rraw.z);
    qx = q.x; qy = q.y; qz = q.z;
    ub = dist2(qx - mp.x, qy - mp.y, qz - mp.z);
    const float3 qo = xform(To, rraw.x, rraw.y, rraw.z);
    const float ddx = qx - qo.x, ddy = qy - qo.y, ddz = qz - qo.z;
    const float delta = __builtin_amdgcn_sqrtf(ddx * ddx + ddy * ddy + ddz * ddz) * (1.0f + 1e-5f) + 1e-7f;   // (hardware square root, 1 ulp: the factor covers it)
    lbn = fmaxf(lb_in * (1.0f - 1e-6f) - delta, 0.f);
    const float lb2 = lbn * lbn;
    const bool keep = ub * (1.0f + 1e-5f) < lb2;
    const bool far = fminf(ub, lb2) > cap2 * (1.0f + 1e-5f);
    skip = keep || far;
    if (skip) {   // the match stands, only its distance moved
      a.d2[j] = ub;
      a.lb[j] = lbn;
    }
  }
  uint32_t nbelow1 = 0u;
  if (sel_on) {   // the skipped lanes' share of the trimmed-distance select (first two passes, as k_knn_tile)
    const uint32_t bits = __float_as_uint(ub);
    const unsigned long long below = __ballot(act && skip && bits < sel_lo);
    if (act && skip && bits >= sel_lo && bits - sel_lo < sel_span) sel_inside(bits);
    nbelow1 = (uint32_t)__popcll(below);   // (one atomic per workgroup, behind the pack's first barrier)
  }
  // ---- pack the searching lanes of the block: slot s of the block goes to wave s / 64, entry s % 64 of its area
#ifdef LSGPU_KNN_STATS
  const long long ts2 = clock64();
#endif
  const unsigned long long sm = __ballot(act && !skip);
  if (lane == 0) { wcount[w] = (uint32_t)__popcll(sm); wbelow[w] = nbelow1; }
  cone_barrier_lds();
  uint32_t before = 0, total = 0;
#pragma unroll
  for (int k = 0; k < WAVES; ++k) { const uint32_t n = wcount[k]; before += k < w ? n : 0u; total += n; }
  if (sel_on && threadIdx.x == 0) {
    uint32_t nb = 0u;
#pragma unroll
    for (int k = 0; k < WAVES; ++k) nb += wbelow[k];
    if (nb) atomicAdd(&a.sel_below[(blockIdx.x & (kSelBelowSlots - 1)) * kSelBelowStride], nb);
  }
  if (act && !skip) {
    ConeRec r;
    r.qx = qx; r.qy = qy; r.qz = qz; r.ub = ub; r.lbn = lbn; r.id_in = id_in; r.j = j; r.pad = 0u;
    const uint32_t at = before + (uint32_t)__popcll(sm & ((1ull << lane) - 1ull));
    rec[at] = r;
  }
  cone_barrier_lds();
#ifdef LSGPU_KNN_STATS
  const long long ts3 = clock64();
  if (lane == 0 && dbg) {
    dbg[0] = (uint32_t)(ts1 - ts0); dbg[1] = (uint32_t)(ts2 - ts1); dbg[2] = (uint32_t)(ts3 - ts2);
    dbg[3] = 1u; dbg[10] = (uint32_t)(ts3 - ts0); dbg[11] = wall0; dbg[12] = (uint32_t)wall_clock64();
    dbg[13] = (uint32_t)__popcll(sm);
  }
#endif
  if ((uint32_t)(w * 64) >= total) return;
  const bool ing = (uint32_t)(w * 64 + lane) < total;
  {
    const ConeRec r = rec[w * 64 + lane];   // (beyond `total`: stale or uninitialised words, never used)
    qx = r.qx; qy = r.qy; qz = r.qz; ub = r.ub; lbn = r.lbn; id_in = r.id_in; j = r.j;
  }
  // ---- phase 2: the searching lanes
  float inv_rho, zeta, pa, rxy, inv_h;
  cone_dir(qx - c.ox, qy - c.oy, qz - c.oz, inv_rho, zeta, pa, rxy, inv_h);
  float ubs = ub;                  // upper bound of the nearest-neighbour distance the search starts from
  if (PROBE) {
    // Launches whose balls are still as wide as the last ICP step (the warm-start point is the match of a transform that
    // has since moved by centimetres): the reference points in the query's OWN direction -- its column and the two next
    // to it, in the occupied rows within two of its own -- are real points, so the nearest of them is an upper bound of
    // the nearest-neighbour distance, usually a far tighter one.  A dozen candidates per lane; the search proper then
    // runs with that radius (and meets these points again: the probe keeps nothing but the bound).
    const bool pr = ing && fabsf(zeta) <= 1.5f && pa >= 0.f && pa <= 4.f;
    const int rq = (int)cone_clampu(floorf((zeta - c.z0) * c.rs), c.rows - 1);
    const int cq = (int)cone_clampu(floorf(pa * c.cs), c.cols - 1);
    const uint32_t p_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_u32(pr ? ~(uint32_t)max(rq - 2, 0) : 0u));
    const uint32_t p_last = wave_max_u32(pr ? (uint32_t)min(rq + 2, c.rows - 1) + 1u : 0u);
    float pb = INFINITY, ps = INFINITY;
    for (uint32_t row = ~p_first; row < p_last; ++row) {
      const float4 rz = rowz_sh[row];
      if (!(rz.x <= rz.y)) continue;
      const bool in = pr && (int)row >= rq - 2 && (int)row <= rq + 2;
      const uint32_t base = row * (uint32_t)c.cols;
      uint32_t st_ = 0u, en_ = 0u;
      if (in) {
        st_ = c.tab[base + (uint32_t)max(cq - 1, 0)];
        en_ = c.tab[base + (uint32_t)min(cq + 1, c.cols - 1) + 1u];
      }
      const uint32_t g0 = st_ >> 2;
      const uint32_t len = en_ > st_ ? min(((en_ + 3u) >> 2) - g0, 4u) : 0u;   // (a bound needs no more than a few groups)
      cone_eval_window(c, g0, len, qx, qy, qz, pb, ps);
    }
    ubs = fminf(ub, pb);
  }
  const float lim0 = prune_lim(ubs, gap, cap2s);                // squared search radius: every point inside is evaluated
  const float R = __builtin_amdgcn_sqrtf(lim0) * (1.0f + 1e-5f) + 1e-7f;
  bool fb = false;                 // this lane searches the voxel grid instead
  float best = INFINITY, sec = INFINITY, bs4 = INFINITY;   // evaluated minimum, second smallest group minimum, runner-up inside the leading group
  float4 bpt = make_float4(0.f, 0.f, 0.f, 0.f);            // the point at the minimum {x, y, z, Morton index}
  bool found = false;
  uint32_t bgrp = 0xFFFFFFFFu;     // group of four direction-sorted points that holds the evaluated minimum
  {
    // ---- the lane's cone.  sin(alpha) = R / rho.  A point within R of q is seen from O under an angle <= alpha from q.
    const float s = R * inv_rho * (1.0f + 1e-5f) + 4e-6f;      // (margin: rounding of q - O and of the points' own directions)
    const float ce = rxy * inv_rho;                             // cos(elevation of q)
    bool cone = ing && s <= 0.5f && ce > 0.f && ce <= 1.5f;     // (NaN-safe: |q - O| == 0 fails)
    const float alpha = s * (1.0f + 0.2f * s * s) + 2e-6f;      // >= asin(s) for s <= 0.5, + rounding of the zetas
    const float alpha2 = alpha * alpha;
    // zeta of such a point: |sin(e_p) - sin(e_q)| <= |sin e_q| (1 - cos alpha) + cos e_q sin alpha <= |zeta| s^2 + ce s
    const float dz = __fmaf_rn(ce, s, fabsf(zeta) * s * s) + 2e-6f;
    const float gq = (rxy * inv_h) * (rxy * inv_h) * (1.0f + 1e-6f);   // d(pa) / d(azimuth) at q = rho_xy^2 / (|x| + |y|)^2, in [1/2, 1]
    const float inv_ce = __builtin_amdgcn_rcpf(ce) * (1.0f + 1e-6f);
    const uint32_t r_lo = cone_clampu(floorf((zeta - dz - c.z0) * c.rs), c.rows - 1);
    const uint32_t r_hi = cone_clampu(floorf((zeta + dz - c.z0) * c.rs), c.rows - 1);
    fb = ing && !cone;
    const int cols = c.cols;
    const f32x2 q2x = {qx, qx}, q2y = {qy, qy}, q2z = {qz, qz};
    // rows of the wave, one after the other (wave-uniform; the occupancy mask skips the empty ones between the rings of a
    // spinning lidar).  The table reads of the NEXT row some lane's cone reaches are issued before the current row's run
    // is staged and evaluated, so only the first row's table round trip is waited for; a row with windows that cross
    // column 0 comes twice, the second time for the wrapped parts.
    uint32_t row = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_min_u32(cone ? r_lo : 0xFFFFFFFFu));
    const uint32_t row_last = wave_max_u32(cone ? r_hi + 1u : 0u);
    int pass = 0;
    uint32_t n_st = 0u, n_en = 0u;   // the next slot's window, its loads in flight
    bool n_in = false;
    auto next_slot = [&]() -> bool {
      n_st = 0u; n_en = 0u; n_in = false;
      while (row < row_last) {
        if (pass == 0) {
          const unsigned long long wd = ne_mask[row >> 6] >> (row & 63u);
          if (!wd) { row = (row | 63u) + 1u; continue; }
          row += (uint32_t)__ffsll((long long)wd) - 1u;
          if (row >= row_last) break;
        }
        const float4 rz = rowz_sh[row];
        // distance in zeta between q and the row's points: a LOWER bound of their difference in elevation
        // (|sin a - sin b| <= |a - b|)
        const float dzr = fmaxf(fmaxf(fmaxf(rz.x - zeta, zeta - rz.y), 0.f) - 1e-6f, 0.f);
        bool in = cone && row >= r_lo && row <= r_hi && dzr <= alpha;
        if (!__ballot(in)) { ++row; pass = 0; continue; }
        // azimuth: cos(theta) = cos(de) - cos(e_q) cos(e_p) (1 - cos(da)), theta <= alpha  =>
        //   sin^2(da / 2) <= (cos(de) - cos(alpha)) / (2 ce ce_p) <= (alpha^2 - de^2) / (4 ce ce_p)
        const float u2 = __fmaf_rn(-dzr, dzr, alpha2) * inv_ce * rz.z;
        const bool polar = in && !(u2 <= 0.25f);                  // the cone reaches (or nears) the polar axis at this row
        const float u = __builtin_amdgcn_sqrtf(fmaxf(u2, 0.f)) * (1.0f + 1e-6f);
        const float da = 2.f * u * (1.0f + 0.2f * u * u);         // >= 2 asin(u)
        // pseudo-azimuth: its derivative is Lipschitz (2 sqrt 2), so |d pa| <= gq da + 1.42 da^2
        const float dp = __fmaf_rn(gq, da, 1.42f * da * da) + 4e-6f;
        const int clo = (int)floorf((pa - dp) * c.cs), chi = (int)floorf((pa + dp) * c.cs);
        const bool wide = in && !polar && chi - clo >= (cols >> 2);
        if (polar || wide) { fb = true; cone = false; }
        in = in && cone;
        // pass 0: the part of the window inside [0, cols); pass 1: the wrapped part of the windows that have one
        int a0 = max(clo, 0), a1 = min(chi, cols - 1);
        const bool wraps = in && (clo < 0 || chi >= cols);
        if (pass == 1) {
          const bool wl = clo < 0;
          a0 = wl ? clo + cols : 0; a1 = wl ? cols - 1 : chi - cols;
          in = wraps;
        }
        const uint32_t base = row * (uint32_t)cols;
        if (in) {
          n_st = c.tab[base + (uint32_t)a0];
          n_en = c.tab[base + (uint32_t)a1 + 1u];
        }
        n_in = in;
        if (pass == 0 && __ballot(wraps)) pass = 1;   // the same row once more, for the wrapped parts
        else { pass = 0; ++row; }
        return true;
      }
      return false;
    };
#ifdef LSGPU_KNN_STATS
    const long long t_r0 = clock64();
#endif
    bool got = next_slot();
    while (got) {
      const uint32_t w_st = n_st, w_en = n_en;
      const bool w_in = n_in;
      got = next_slot();
#ifdef LSGPU_KNN_STATS
      if (!n_rowslots) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t_rows += clock64() - t_r0; }
#endif
      // ---- every lane steps through its own window, two groups per step: all six loads of a step are issued before the
      // first is waited for (one memory round trip for eight candidates), a lane past its own end sits the step out
      if (w_in && w_en - w_st > kConeMaxWin) { fb = true; cone = false; }
      const bool wv = w_in && cone && w_en > w_st;
      const uint32_t g0 = w_st >> 2, g1 = wv ? (w_en + 3u) >> 2 : 0u;   // this lane's groups [g0, g1)
      if (!__ballot(wv)) continue;
#ifdef LSGPU_KNN_STATS
      ++n_rowslots; n_groups += wave_sum_u32(wv ? g1 - g0 : 0u);
      const long long t_s1 = clock64();
#endif
      for (uint32_t t = g0; __ballot(wv && t < g1); t += 2u) {
#ifdef LSGPU_KNN_STATS
        ++n_steps;
#endif
        if (wv && t < g1) {
          const bool two = t + 1u < g1;
          const float4* __restrict__ p0 = c.soa + (uint32_t)kConeGF4 * (size_t)t;
          const float4* __restrict__ p1 = two ? p0 + kConeGF4 : p0;   // (no second group: the first once more -- its line is on its way
                                                                         //  anyway; one fixed line for all such lanes measured slower)   // (no second group: the first once more, left out of the minimum below)
          const float4 X = p0[0], Y = p0[1], Z = p0[2];
          const float4 X1 = p1[0], Y1 = p1[1], Z1 = p1[2];
          const f32x2 d0 = dist2_pair(q2x, q2y, q2z, f32x2{X.x, X.y}, f32x2{Y.x, Y.y}, f32x2{Z.x, Z.y});
          const f32x2 d1 = dist2_pair(q2x, q2y, q2z, f32x2{X.z, X.w}, f32x2{Y.z, Y.w}, f32x2{Z.z, Z.w});
          const f32x2 e0 = dist2_pair(q2x, q2y, q2z, f32x2{X1.x, X1.y}, f32x2{Y1.x, Y1.y}, f32x2{Z1.x, Z1.y});
          const f32x2 e1 = dist2_pair(q2x, q2y, q2z, f32x2{X1.z, X1.w}, f32x2{Y1.z, Y1.w}, f32x2{Z1.z, Z1.w});
          const float m4 = fminf(fminf(fminf(d0.x, d0.y), d1.x), d1.y);
          const float n4 = two ? fminf(fminf(fminf(e0.x, e0.y), e1.x), e1.y) : INFINITY;
          sec = __builtin_amdgcn_fmed3f(best, m4, sec);
          bgrp = m4 < best ? t : bgrp;
          best = fminf(best, m4);
          sec = __builtin_amdgcn_fmed3f(best, n4, sec);
          bgrp = n4 < best ? t + 1u : bgrp;
          best = fminf(best, n4);
        }
      }
#ifdef LSGPU_KNN_STATS
      t_eval += clock64() - t_s1;
#endif
    }
  }
  // ---- the group that holds the evaluated minimum, once more: which of its points, the runner-up inside the group, the
  // point's Morton index (the group minima of all other groups are in `sec` already)
  if (bgrp != 0xFFFFFFFFu) {
    const float4* __restrict__ p = c.soa + (uint32_t)kConeGF4 * (size_t)bgrp;
    const float4 X = p[0], Y = p[1], Z = p[2];
    const uint4 m = reinterpret_cast<const uint4*>(c.map)[bgrp];   // (goes out with the three loads above: no extra round trip)
    const float4 M = make_float4(__uint_as_float(m.x), __uint_as_float(m.y), __uint_as_float(m.z), __uint_as_float(m.w));
    const float e0 = dist2(qx - X.x, qy - Y.x, qz - Z.x), e1 = dist2(qx - X.y, qy - Y.y, qz - Z.y);
    const float e2 = dist2(qx - X.z, qy - Y.z, qz - Z.z), e3 = dist2(qx - X.w, qy - Y.w, qz - Z.w);
    if (e3 == best) bpt = make_float4(X.w, Y.w, Z.w, M.w);
    if (e2 == best) bpt = make_float4(X.z, Y.z, Z.z, M.z);
    if (e1 == best) bpt = make_float4(X.y, Y.y, Z.y, M.y);
    if (e0 == best) bpt = make_float4(X.x, Y.x, Z.x, M.x);
    bs4 = fminf(fmaxf(fminf(e0, e1), fminf(e2, e3)), fminf(fmaxf(e0, e1), fmaxf(e2, e3)));
    found = true;
  }
#ifdef LSGPU_KNN_STATS
  const long long ts5 = clock64();
#endif
  // ---- lanes the index could not serve: the voxel grid (same search as a spread wave's lanes in k_knn_tile)
  int bi = id_in;
  mp = make_float4(0.f, 0.f, 0.f, __int_as_float(id_in));   // (only the index of the old match matters from here on)
  const unsigned long long fbm = __ballot(fb);
  if (fbm) {
    if (fb) {
      best = ub;
      lane_ball_search(a, cap2s, qx, qy, qz, best, bi);
      if (bi != id_in) { const float4 p = a.pts[bi]; mp = make_float4(p.x, p.y, p.z, __int_as_float(bi)); }
    }
    if (lane == 0) atomicAdd(a.strag_count, (uint32_t)__popcll(fbm));   // (counted, not listed: they are done)
  }
  if (ing) {
    float nb;  // new lower bound on the distance to every point other than the (new) match
    if (fb) {
      nb = best <= cap2s ? sqrtf(best) * (1.0f - 1e-6f) : fmaxf(lbn, sqrtf(cap2s) * (1.0f - 1e-5f));
    } else if (found && best <= ub) {
      // the evaluated minimum (the warm-start point itself unless something beat it)
      mp = bpt;
      sec = fminf(sec, bs4);
      float others = fminf(sec, lim0);   // every point that was not evaluated lies beyond the search radius
      bool same = __float_as_int(mp.w) == id_in;
      if (sec == best || (!same && best == ub)) {  // a second point at exactly this distance: smallest Morton index
        mp = canonical_tie(a, qx, qy, qz, best, mp);
        same = __float_as_int(mp.w) == id_in;
      }
      if (!same) others = fminf(others, ub);  // (covers a warm-start point outside the windows)
      nb = __builtin_amdgcn_sqrtf(others) * (1.0f - 1e-5f);
      if (same) nb = fmaxf(nb, lbn);
    } else {  // the warm-start point lies beyond the cap and nothing closer exists: the match stands
      nb = fmaxf(__builtin_amdgcn_sqrtf(fminf(best, lim0)) * (1.0f - 1e-5f), lbn);
      best = ub;
    }
    if (a.write_all || __float_as_int(mp.w) != id_in) {
      a.ids[j] = __float_as_int(mp.w);
      a.prev[j] = mp;
    }
    a.d2[j] = best;
    a.lb[j] = nb;
  }
  if (sel_on) {   // the searching lanes' share of the select
    const uint32_t bits = __float_as_uint(best);
    const unsigned long long below = __ballot(ing && bits < sel_lo);
    if (ing && bits >= sel_lo && bits - sel_lo < sel_span) sel_inside(bits);
    if (lane == 0 && below) atomicAdd(&a.sel_below[((tile + 32u) & (kSelBelowSlots - 1)) * kSelBelowStride], (uint32_t)__popcll(below));
  }
#ifdef LSGPU_KNN_STATS
  {
    const long long ts9 = clock64();
    const uint32_t n_ing = (uint32_t)__popcll(__ballot(ing)), n_fb = (uint32_t)__popcll(fbm);
    if (lane == 0 && dbg) {
      dbg[3] = 2u | (n_ing << 8) | (n_fb << 16); dbg[4] = (uint32_t)t_rows; dbg[5] = (uint32_t)t_eval; dbg[6] = n_chunks;
      dbg[7] = n_steps; dbg[8] = n_groups; dbg[15] = (uint32_t)t_stage | 0u; dbg[13] = n_rowslots;
      dbg[9] = (uint32_t)(ts9 - ts5); dbg[10] = (uint32_t)(ts9 - ts0); dbg[12] = (uint32_t)wall_clock64(); dbg[14] = (uint32_t)(ts5 - ts3);
    }
  }
#endif
}

}  // namespace lsgpu
