// lsgpu_knn.hip.h -- exact 1-NN correspondence search: KDTreeMatcher::findClosests, knn 1,
// epsilon 0, squared distances (laser_slam/configurations/icp_default.yaml:9-12), run once per ICP
// iteration inside icp_.compute (laser_slam/src/laser_track.cpp:496).
//
// Scheme (all exact, no approximation):
//   * every query carries an upper bound: the distance to a known reference point (its match of
//     the previous iteration, or a seed from k_knn_seed).  The true NN lies inside that ball.
//   * k_knn_tile: one WAVE = 64 Morton-neighbouring queries.  The wave takes the bounding box of
//     its lanes' balls, picks the pyramid level at which that box spans <= 4x4x4 cells (one hash
//     lookup per lane), then walks the cells' chunks: 64 chunk AABBs are culled lane-parallel, a
//     surviving chunk is tested per lane against that lane's current best, and only if some lane
//     needs it are its <= 64 points staged through LDS and broadcast to all lanes.
//   * lanes whose ball is larger than r_cap go to k_knn_fallback: one wave per query, chunks culled
//     lane-parallel against the single ball, surviving chunks evaluated one point per lane.
//   * (k_knn_lane, one LANE per query with lane_ball_search, was measured slower and lives in the -DLSGPU_EXPERIMENTS
//     build only; lane_ball_search itself serves the rare tie and late-spread paths.)
//   * TrimmedDistOutlierFilter (yaml:14-16) gives every pair beyond the trim limit weight 0, so in
//     the ICP loop a lane needs an exact neighbour only if it is closer than cap = sqrt(cap2), a
//     bound the host derives from the previous iteration's limit and verifies after the select
//     (a violated bound repeats the iteration uncapped).  Lanes with nothing inside the cap keep an
//     upper bound > cap2 ("far"); their order among themselves never matters.
//   * Lower bounds.  Every query also keeps a LOWER bound lb on its NN distance (exact distance after
//     an exact search, the verified radius otherwise).  Between two iterations a query moves by
//     delta = |T_new r - T_old r|, so its new NN distance is >= lb - delta (triangle inequality): if
//     that already exceeds the cap the lane is "far" without searching at all.
// Ties in distance: libnabo's order is implementation defined, so any nearest point is a valid answer; this
// library returns the one with the SMALLEST index in its Morton-sorted reference, whichever kernel or wave
// composition found it (the broadcast evaluations record the first group that reaches the minimum and detect a
// second group at exactly the same distance through the runner-up they track anyway; the rare tie is then
// settled by canonical_tie).  Results therefore do not depend on how queries are grouped into waves.
#pragma once
#include "lsgpu_common.hip.h"

namespace lsgpu {

constexpr float kPruneShrink = 1.0f - 2e-6f;  // a box is skipped only if mind2 * this > best
constexpr float kCapSearchMargin2 = 1.05f * 1.05f;  // (search radius / cap radius)^2 in capped launches
constexpr float kPadCoord = 3e18f;            // LDS pad point: squared distance ~2.7e37, never best

// What k_knn_cone would pay for a lane: rows of its cone x columns at the query's own elevation x reference points per bin,
// in evaluation steps of four candidates (the quantities of lsgpu_cone.hip.h's phase 2 without their margins).  A spinning
// lidar's cloud is uniform in DIRECTION whatever the range, so the density is one number per reference.
constexpr int kPriceSlots = 64, kPriceStride = 32;
struct ConePrice {
  float ox, oy, oz, rs, cs, dens4;
  float heavy;              // a lane above this many steps is "heavy" (so is one the index cannot serve)
  uint32_t* count;          // kPriceSlots x {heavy lanes, searching lanes}, one pair per 128-byte line (hashed by tile: atomics
                            // on one line serialise in the L2 -- two counters for 16 k waves cost 100 us)
};
__device__ __forceinline__ float cone_price(const ConePrice& c, float qx, float qy, float qz, float R) {
  const float vx = qx - c.ox, vy = qy - c.oy, vz = qz - c.oz;
  const float r2 = __fmaf_rn(vy, vy, vx * vx);
  const float inv_rho = __builtin_amdgcn_rsqf(__fmaf_rn(vz, vz, r2));
  const float rxy = __builtin_amdgcn_sqrtf(r2);
  const float sn = R * inv_rho, ce = rxy * inv_rho, zeta = fabsf(vz * inv_rho);
  const float g = rxy * __builtin_amdgcn_rcpf(fabsf(vx) + fabsf(vy));
  const float rows = 2.f * (ce * sn + zeta * sn * sn) * c.rs + 1.f;
  const float cols = 2.f * (g * g) * sn * __builtin_amdgcn_rcpf(ce) * c.cs + 1.f;
  const bool served = sn <= 0.5f && ce > 0.05f && ce <= 1.5f;
  return served ? rows * __fmaf_rn(cols, c.dens4, 0.5f) : INFINITY;
}

struct KnnArgs {
  const float4* rdq;        // sorted reading (already moved by T_refMean_dataIn), w = caller index
  int nq;
  Mat34 T;                  // T_iter, applied on load (RigidTransformation, yaml default)
  GridDev g;
  const float4* pts;        // Morton-sorted centred reference
  const ChunkDesc* chunks;
  const ChunkDesc* cgroups;  // bounding box of every kChunkGroup consecutive chunks (k_chunk_cnt4)
  const float4* soa;        // chunk-blocked SoA copy of pts for the broadcast evaluation: chunk c = x[cnt4] y[cnt4] z[cnt4]
  const uint32_t* chunk_soa;  //   float4 index of chunk c's block (cnt4 = count rounded up to 4, pads far away)
  int* ids;                 // out: sorted-reference index of the NN
  float* d2;                // out: squared distance
  float4* prev;             // in/out: warm start = the query's current match {x,y,z, sorted index bits}
  uint32_t* strag;          // out: straggler list
  uint32_t* strag_count;
  int route_heavy_max;   // heavy tiles (see k_knn_tile) handed to the wave-per-query pass per launch; < 0: all of them (no ticket)
  float r_cap;              // lanes with a larger ball go to the fallback
  float group_r;            // half extent of one search group inside a wave
  float cap2;               // only neighbours with d2 <= cap2 must be exact (INF: all)
  float* lb;                // per-query lower bound on the distance to every point OTHER than prev (nullable)
  float gap;                // capped launches search `gap` metres beyond the current best (keep-match bound)
  float spread_route_r;     // > 0: a spread wave whose largest ball exceeds this hands its lanes to k_knn_fallback
  int route_chunks;         // (with spread_route_r > 0) so does any wave whose cell block holds more chunks than this
  int route_dense;          // (with spread_route_r > 0) and a SPREAD wave whose cell block holds more chunks than this, whatever its balls
#ifdef LSGPU_EXPERIMENTS
  int sparse_lanes;         // > 0: a wave with at most this many searching lanes hands them to k_knn_rowq
  int xcd_swizzle;          // 1: block b -> XCD (b % 8) gets a contiguous eighth of the tiles
  uint32_t* work;           // compacted list of the queries that have to search (k_knn_classify -> k_knn_rows)
  uint32_t* work_count;     //   its length; re-armed by the last block of k_normal_eq_loop
#endif
  // front rows (settled launches): tiles found spread in an earlier launch are searched row-wise by the first
  // `front_blocks` workgroups of the tile kernel itself (8 per tile) -- no hand-over list, no second launch
  uint32_t* spread_flag;    // per tile: on the list (nullable)
  uint32_t* spread_list;    // tiles found spread so far
  uint32_t* spread_cnt;     // [0] entries the front rows may use (committed by k_normal_eq_loop), [1] entries appended
  int front_blocks;
  // the search before the first one through the direction index (lsgpu_cone.hip.h) prices that index for this align
  ConePrice price;          // price.count == nullptr: not this launch
  uint32_t* sel_hist2;      // predicted select (IcpState::sel_mode): 2048-slice histogram of the distances inside the window
  uint32_t* sel_below;      //   kSelBelowSlots counters of distances below the bin (nullable: launch without prediction)
  uint32_t* sel_hist3w;     // committed select: kSelWinRows x 512 histogram of bits [8:0] around the last limit (nullable)
  int sel_force;            //   1: histogram against the last limit's bins whatever IcpState::sel_mode says
  int write_all;            // 1: store index + warm start of every query (first search of an align, kernel-level API);
                            // 0: only where the match changed (they were stored by an earlier launch)
  const IcpState* st;       // loop state (nullable): overrides T (and cap2 if use_state_cap)
  int use_state_cap;
  unsigned long long* dbg;  // optional counters (LSGPU_KNN_STATS builds only)
  uint4* dbg_wave;          // optional per-wave {cycles, chunk evals, proxy survivors, groups<<8|level}
  int ntiles;               // number of 64-query tiles
  int pad_index;            // index of the first far pad point behind pts (= Nr)
  int chunk_budget;         // a wide wave whose region holds more chunks than this searches per lane
  uint2* cell_cache;        // per tile: the 64 (chunk_start, chunk_end) probe results of its cell block
  ulonglong2* cell_tags;    // per tile: which block (generation, level, origin, extent) the cache holds
  uint32_t cache_gen;       // bumped by every set_reference / align: older entries never match
  int dbg_flags;            // LSGPU_KNN_STATS builds: ablation switches (1 no eval, 2 no refine, 4 no search)
};

#ifdef LSGPU_KNN_STATS
#define KNN_COUNT(slot, v) do { if (lane == 0 && a.dbg) atomicAdd(&a.dbg[slot], (unsigned long long)(v)); } while (0)
#else
#define KNN_COUNT(slot, v) do { } while (0)
#endif

__device__ __forceinline__ float box_dist2(float lx, float ly, float lz, float hx, float hy,
                                           float hz, float qx, float qy, float qz) {
  const float dx = fmaxf(fmaxf(lx - qx, qx - hx), 0.f);
  const float dy = fmaxf(fmaxf(ly - qy, qy - hy), 0.f);
  const float dz = fmaxf(fmaxf(lz - qz, qz - hz), 0.f);
  return dx * dx + dy * dy + dz * dz;
}

// The predicted / committed / fused select counts against a window of the distances' bit patterns
// [sel_lo, sel_lo + sel_span) in slices of 2^sel_shift (IcpState; the aligned mode's window is the last limit's 12-bit
// float bin in 2048 slices of 512, the fused mode's [0.7, 1.1] x the last limit).  One final distance INSIDE the window:
// the slice histogram, and the third-level window table where the launch carries one (aligned mode only).
__device__ __forceinline__ void sel_count_inside(const KnnArgs& a, uint32_t bits) {
  const uint32_t bin2 = (bits - a.st->sel_lo) >> a.st->sel_shift;
  atomicAdd(&a.sel_hist2[bin2], 1u);
  if (a.sel_hist3w) {
    const uint32_t d = bin2 - a.st->sel_bin2 + (uint32_t)kSelWinHalf;
    if (d < (uint32_t)kSelWinRows) atomicAdd(&a.sel_hist3w[d * 512u + (bits & 0x1FFu)], 1u);
  }
}

// ---------------------------------------------------------------- seed
// Any reference point near the query: climb the pyramid from level 0 until the cell holding the
// query (clamped into the grid) exists, take the best of the first 8 points of its first chunk.
__global__ __launch_bounds__(256) void k_knn_seed(KnnArgs a) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= a.nq) return;
  Mat34 T; float cap2;
  if (!iter_params(a.st, a.T, a.cap2, a.use_state_cap, T, cap2)) return;
  const float4 r = a.rdq[j];
  const float3 q = xform(T, r.x, r.y, r.z);
  const GridDev& g = a.g;
  const int lim = (1 << (g.bits + g.fine)) - 1;
  const int fx = fine_coord(q.x, g.ox, g.inv_hf, lim);
  const int fy = fine_coord(q.y, g.oy, g.inv_hf, lim);
  const int fz = fine_coord(q.z, g.oz, g.inv_hf, lim);
  int bi = 0;
  float4 bp = a.pts[0];
  float bd = INFINITY;
  for (int l = 0; l <= g.bits; ++l) {
    const int sh = g.fine + l;
    uint32_t cs, ce;
    if (!grid_lookup(g, l, (uint32_t)(fx >> sh), (uint32_t)(fy >> sh), (uint32_t)(fz >> sh), cs, ce))
      continue;
    // level 0: every point of the cell (tight seed where the reference is dense); coarser levels only
    // serve isolated queries: one chunk is enough for a bound
    const uint32_t p0 = a.chunks[cs].start;
    const ChunkDesc dl = a.chunks[l == 0 ? ce - 1 : cs];
    uint32_t p1 = dl.start + dl.count;
    if (p1 - p0 > 256u) p1 = p0 + 256u;
    float best = INFINITY;
    for (uint32_t t = p0; t < p1; t += 4) {  // pts is padded: running a few points past p1 is harmless
      const float4 c0 = a.pts[t], c1 = a.pts[t + 1], c2 = a.pts[t + 2], c3 = a.pts[t + 3];
      const float d0 = dist2(q.x - c0.x, q.y - c0.y, q.z - c0.z), d1 = dist2(q.x - c1.x, q.y - c1.y, q.z - c1.z);
      const float d2 = dist2(q.x - c2.x, q.y - c2.y, q.z - c2.z), d3 = dist2(q.x - c3.x, q.y - c3.y, q.z - c3.z);
      if (d0 < best) { best = d0; bi = (int)t; bp = c0; }
      if (d1 < best) { best = d1; bi = (int)t + 1; bp = c1; }
      if (d2 < best) { best = d2; bi = (int)t + 2; bp = c2; }
      if (d3 < best) { best = d3; bi = (int)t + 3; bp = c3; }
    }
    bd = best;
    break;
  }
  a.prev[j] = make_float4(bp.x, bp.y, bp.z, __int_as_float(bi));
  // the seed distance bounds the nearest-neighbour distance from above, query by query, hence so does every order
  // statistic: the trim quantile of the seed distances is a guaranteed search cap for the first iteration
  a.d2[j] = bd;
  if (a.lb) a.lb[j] = 0.f;  // nothing is known yet about the other points: no keep / far skip in the first search
}

// ---------------------------------------------------------------- per-lane ball search
// Exact for every reference point with d2 <= min(best, cap2): looks up the <= 2x2x2 cells the ball
// touches at the level where it spans at most two cells per axis and walks their chunks.
// pts is padded by 8 far points, so 4-wide point loads may run past a chunk's end (the extra points
// are real reference points or pads: evaluating them is harmless).
__device__ __forceinline__ void lane_ball_search(const KnnArgs& a, float cap2, float qx, float qy, float qz,
                                                 float& best, int& bi) {
  const GridDev& g = a.g;
  float prune = fminf(best, cap2);
  const float R = sqrtf(prune) * (1.0f + 1e-5f) + 1e-7f + kFineSlack * g.hf;
  const int lim = (1 << (g.bits + g.fine)) - 1;
  const int flx = fine_coord(qx - R, g.ox, g.inv_hf, lim), fhx = fine_coord(qx + R, g.ox, g.inv_hf, lim);
  const int fly = fine_coord(qy - R, g.oy, g.inv_hf, lim), fhy = fine_coord(qy + R, g.oy, g.inv_hf, lim);
  const int flz = fine_coord(qz - R, g.oz, g.inv_hf, lim), fhz = fine_coord(qz + R, g.oz, g.inv_hf, lim);
  int l = 0, sh = g.fine;
  for (; l < g.bits; ++l, ++sh)
    if ((fhx >> sh) - (flx >> sh) < 2 && (fhy >> sh) - (fly >> sh) < 2 && (fhz >> sh) - (flz >> sh) < 2)
      break;
  sh = g.fine + l;
  const int x0 = flx >> sh, y0 = fly >> sh, z0 = flz >> sh;
  const int x1 = fhx >> sh, y1 = fhy >> sh, z1 = fhz >> sh;
  // all (up to 8) first probes in flight together: one table latency instead of eight
  const uint32_t mask = g.mask[l];
  const uint4* tab = reinterpret_cast<const uint4*>(g.tab[l]);
  uint4 en[8];
  uint32_t slot[8];
  bool want[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int cx = x0 + (c & 1), cy = y0 + ((c >> 1) & 1), cz = z0 + (c >> 2);
    want[c] = cx <= x1 && cy <= y1 && cz <= z1;
    slot[c] = cell_hash((uint32_t)cx, (uint32_t)cy, (uint32_t)cz) & mask;
    en[c] = make_uint4(kEmpty, 0u, 0u, 0u);
    if (want[c]) en[c] = tab[slot[c]];
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (!want[c]) continue;
    const uint32_t cx = (uint32_t)(x0 + (c & 1)), cy = (uint32_t)(y0 + ((c >> 1) & 1)), cz = (uint32_t)(z0 + (c >> 2));
    const uint32_t xy = cx | (cy << 16);
    uint4 e = en[c];
    uint32_t sl = slot[c];
    while (!(((e.x ^ xy) | (e.y ^ cz)) == 0u) && e.x != kEmpty) {  // collision chain (rare)
      sl = (sl + 1) & mask;
      e = tab[sl];
    }
    if (e.x == kEmpty) continue;
    for (uint32_t ch = e.z; ch < e.w; ++ch) {
      // a cell of hundreds of chunks (dense near-range geometry: 550 chunks in one 12.5 cm cell next to a wall): sixteen
      // chunk boxes are skipped with one test of their common box
      if ((ch & (uint32_t)(kChunkGroup - 1)) == 0u && ch + (uint32_t)kChunkGroup <= e.w) {
        const float4* gd = reinterpret_cast<const float4*>(a.cgroups + ch / (uint32_t)kChunkGroup);
        const float4 g0 = gd[0], g1 = gd[1];
        if (!(box_dist2(g0.x, g0.y, g0.z, g1.x, g1.y, g1.z, qx, qy, qz) * kPruneShrink <= prune)) { ch += (uint32_t)kChunkGroup - 1u; continue; }
      }
      const float4* cd = reinterpret_cast<const float4*>(a.chunks + ch);
      const float4 b0 = cd[0], b1 = cd[1];
      if (!(box_dist2(b0.x, b0.y, b0.z, b1.x, b1.y, b1.z, qx, qy, qz) * kPruneShrink <= prune)) continue;
      const uint32_t st = __float_as_uint(b0.w), cnt = __float_as_uint(b1.w);
      for (uint32_t t = 0; t < cnt; t += 4) {
        const float4 p0 = a.pts[st + t], p1 = a.pts[st + t + 1], p2 = a.pts[st + t + 2], p3 = a.pts[st + t + 3];
        const float d0 = dist2(qx - p0.x, qy - p0.y, qz - p0.z);
        const float d1 = dist2(qx - p1.x, qy - p1.y, qz - p1.z);
        const float d2 = dist2(qx - p2.x, qy - p2.y, qz - p2.z);
        const float d3 = dist2(qx - p3.x, qy - p3.y, qz - p3.z);
        // equal distances: the smaller index wins, whatever the visiting order (canonical ties, see header)
        if (d0 < best || (d0 == best && (int)(st + t) < bi)) { best = d0; bi = (int)(st + t); }
        if (d1 < best || (d1 == best && (int)(st + t + 1) < bi)) { best = d1; bi = (int)(st + t + 1); }
        if (d2 < best || (d2 == best && (int)(st + t + 2) < bi)) { best = d2; bi = (int)(st + t + 2); }
        if (d3 < best || (d3 == best && (int)(st + t + 3) < bi)) { best = d3; bi = (int)(st + t + 3); }
      }
      prune = fminf(best, cap2);
    }
  }
}

// Two different reference points at exactly the distance `best`: the smaller index wins (rare path).
__device__ __forceinline__ float4 canonical_tie(const KnnArgs& a, float qx, float qy, float qz, float best, float4 mp) {
  int bi = __float_as_int(mp.w);
  const int before = bi;
  float b = best;
  lane_ball_search(a, INFINITY, qx, qy, qz, b, bi);  // nothing is closer than best; equal distance + smaller index wins
  if (bi != before) { const float4 p = a.pts[bi]; mp = make_float4(p.x, p.y, p.z, __int_as_float(bi)); }
  return mp;
}

// ---------------------------------------------------------------- tile search
constexpr int kListCap = 128;        // chunk ids queued per wave (LDS)
constexpr uint32_t kChunkBudget = 1024;  // a whole-wave group is accepted up to this many chunks

struct TileLds {
  float4 slot[4][64];            // 4 chunks in flight: filled by LDS-DMA (global_load_lds, 16 B per lane)
  uint32_t list[kListCap];       // flattened chunk ids of the region's cells
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

// d2 of two candidates at once; each half is exactly fma(dz,dz,fma(dy,dy,dx*dx)) (v_pk_add/mul/fma_f32)
__device__ __forceinline__ f32x2 dist2_pair(f32x2 qx, f32x2 qy, f32x2 qz, f32x2 cx, f32x2 cy, f32x2 cz) {
  const f32x2 dx = qx - cx, dy = qy - cy, dz = qz - cz;
  f32x2 d = dx * dx;
  d = __builtin_elementwise_fma(dy, dy, d);
  d = __builtin_elementwise_fma(dz, dz, d);
  return d;
}

// Broadcast-evaluate one staged chunk: 4 candidates per step -- 12 packed-pair ops for the distances,
// min3 + min, then ONE compare/select pair that records the group of 4 holding the new best; the exact
// index is resolved once at the end of the kernel (tile_resolve_match).
// Squared search radius of a lane: `gap` beyond its current bound b (so that the search also proves
// that no OTHER point lies within sqrt(b) + gap), never beyond the cap.  gap == 0: min(b, cap2).
__device__ __forceinline__ float prune_lim(float b, float gap, float cap2) {
  return fminf(__fmaf_rn(2.f * gap, __builtin_amdgcn_sqrtf(b), b) + gap * gap, cap2);
}

__device__ __forceinline__ void tile_eval_slot(const float4* __restrict__ slot, uint32_t st, uint32_t cnt,
                                               float qx, float qy, float qz, float& best, float& sec, int& grp) {
  // the slot holds the chunk's SoA block: x[cnt4] y[cnt4] z[cnt4].  One ds_read_b128 per coordinate fetches four
  // candidates as two register pairs, exactly the operands of the packed-pair arithmetic (no shuffles).
  const uint32_t c4 = (cnt + 3u) >> 2;  // float4s per coordinate
  const f32x2 q2x = {qx, qx}, q2y = {qy, qy}, q2z = {qz, qz};
#pragma unroll 2
  for (uint32_t t = 0; t < c4; ++t) {
    const float4 x = slot[t], y = slot[c4 + t], z = slot[2u * c4 + t];
    const f32x2 d0 = dist2_pair(q2x, q2y, q2z, f32x2{x.x, x.y}, f32x2{y.x, y.y}, f32x2{z.x, z.y});
    const f32x2 d1 = dist2_pair(q2x, q2y, q2z, f32x2{x.z, x.w}, f32x2{y.z, y.w}, f32x2{z.z, z.w});
    const float m4 = fminf(fminf(fminf(d0.x, d0.y), d1.x), d1.y);
    sec = __builtin_amdgcn_fmed3f(best, m4, sec);  // second smallest group minimum (best <= sec always)
    if (m4 < best) { best = m4; grp = (int)(st + 4u * t); }
  }
}

// ---- lane split (k_knn_tile<1, false, true>: the settled launches of the voxel search -- dense local maps, scans whose
// direction index rests).  A settled wave has 22-26 searching lanes of 64 (keep / far lanes have left), yet the broadcast
// evaluation above runs every candidate through all 64 lanes.  With no more than 32 (16, 8) searching lanes the wave is cut
// into 2 (4, 8) PARTS of S = 32 (16, 8) lanes: lane L serves the (L mod S)-th searching query and takes every ways-th group
// of four of each staged chunk, so a chunk costs a half (quarter, eighth) of the steps.  The parts' partial results {smallest
// group minimum, second smallest, its group} are merged over the parts by xor-shuffles and handed to the query's own lane at
// the end of every batch -- the same (min, second, group) algebra as one more step of tile_eval_slot; two parts that found the
// same minimum leave second == best, which sends the lane to canonical_tie as any tie does.  Results are those of the
// plain evaluation (the order candidates are looked at does not enter them).
struct SplitState {
  int ways;          // 1: plain evaluation (wave-uniform)
  float qx, qy, qz;  // the served query
  float best, sec;   // partial result of this lane's share of the batch
  int grp;
};

__device__ __forceinline__ void split_setup(SplitState& sp, unsigned long long ing_mask, int lane, float qx, float qy, float qz) {
  const int nin = __popcll(ing_mask);
  sp.ways = nin <= 8 ? 8 : nin <= 16 ? 4 : nin <= 32 ? 2 : 1;
  int own = lane; sp.qx = 0.f; sp.qy = 0.f; sp.qz = 0.f; sp.best = INFINITY; sp.sec = INFINITY; sp.grp = -1;
  if (sp.ways == 1) return;
  const int slot = lane & (64 / sp.ways - 1);
  unsigned long long mm = ing_mask;
  for (int idx = 0; mm; ++idx) {
    const int k = __ffsll((long long)mm) - 1;
    mm &= mm - 1;
    if (slot == idx) own = k;
  }
  sp.qx = __shfl(qx, own, 64); sp.qy = __shfl(qy, own, 64); sp.qz = __shfl(qz, own, 64);   // (a lane without a query to serve keeps its own: harmless)
}

// this lane's share of one staged chunk: groups part, part + ways, ... for the served query
__device__ __forceinline__ void tile_eval_slot_split(const float4* __restrict__ slot, uint32_t st, uint32_t cnt, int lane,
                                                     SplitState& sp) {
  const uint32_t c4 = (cnt + 3u) >> 2;
  const uint32_t ways = (uint32_t)sp.ways, part = (uint32_t)lane / (64u / ways);
  const f32x2 q2x = {sp.qx, sp.qx}, q2y = {sp.qy, sp.qy}, q2z = {sp.qz, sp.qz};
  for (uint32_t t = part; t < c4; t += ways) {
    const float4 x = slot[t], y = slot[c4 + t], z = slot[2u * c4 + t];
    const f32x2 d0 = dist2_pair(q2x, q2y, q2z, f32x2{x.x, x.y}, f32x2{y.x, y.y}, f32x2{z.x, z.y});
    const f32x2 d1 = dist2_pair(q2x, q2y, q2z, f32x2{x.z, x.w}, f32x2{y.z, y.w}, f32x2{z.z, z.w});
    const float m4 = fminf(fminf(fminf(d0.x, d0.y), d1.x), d1.y);
    sp.sec = __builtin_amdgcn_fmed3f(sp.best, m4, sp.sec);
    if (m4 < sp.best) { sp.best = m4; sp.grp = (int)(st + 4u * t); }
  }
}

// end of a batch: parts -> one partial result per served query -> the query's own lane; the parts start afresh
__device__ __forceinline__ void split_merge(SplitState& sp, bool ing, unsigned long long ing_mask, int lane, float& best, float& sec, int& grp) {
  for (int o = 32; o >= 64 / sp.ways; o >>= 1) {
    const float ob = __shfl_xor(sp.best, o, 64), os = __shfl_xor(sp.sec, o, 64);
    const int og = __shfl_xor(sp.grp, o, 64);
    sp.sec = fminf(fmaxf(sp.best, ob), fminf(sp.sec, os));
    if (ob < sp.best) { sp.best = ob; sp.grp = og; }   // (equal minima: sec == best now, either group will do)
  }
  const int rank = __popcll(ing_mask & ((1ull << lane) - 1ull));   // this lane's position among the searching lanes = the slot that served it
  const float mb = __shfl(sp.best, rank, 64), ms = __shfl(sp.sec, rank, 64);
  const int mg = __shfl(sp.grp, rank, 64);
  if (ing) {
    sec = fminf(fmaxf(best, mb), fminf(sec, ms));
    if (mb < best) { best = mb; grp = mg; }
  }
  sp.best = INFINITY; sp.sec = INFINITY; sp.grp = -1;
}

// Cull 64 queued chunks (one per lane) against the group's query box, test the survivors per lane
// against each lane's own bound, then fetch the needed chunks FOUR AT A TIME with LDS-DMA (one memory
// latency per four chunks, no staging registers) and broadcast-evaluate them.
// LAZY (launches whose balls are still WIDE -- the first iterations of an align: a ball is as large as the last ICP step,
// 15 cm in the median of iteration 1, while the neighbour sits 2-5 cm from the query): the chunks that overlap the tile's own
// query box are fetched first, and every chunk is RE-TESTED against the lanes' current bounds right before it is fetched --
// once the near chunks are evaluated most of the others are needed by nobody.  Exact: the bounds are upper bounds of the
// final distances, a skipped chunk lies beyond the final search radius.  Worth 35-40 us in each of the first two launches;
// compiled into its own instantiation of the kernel, because the same code in the settled launches costs them 8 % (it
// lengthens the live ranges of a kernel that sits at its register budget; profiles/r03_knn_variants.txt).
template <bool LAZY, bool SPLIT>
__device__ __forceinline__ void tile_process_batch(const KnnArgs& a, float cap2, TileLds& lds, int lane, bool valid,
                                                   uint32_t ch, bool ing, float qx, float qy, float qz,
                                                   float tlx, float tly, float tlz, float thx, float thy,
                                                   float thz, float& maxbest, float ub, float gap, float& best,
                                                   float& sec, int& grp, uint32_t& n_eval, uint32_t& n_surv,
                                                   uint32_t& c_eval /* stats builds: cycles spent fetching + evaluating */,
                                                   SplitState& sp) {
  float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
  uint32_t sbase = 0;
  bool pass = false, near = false;
  float gdl = INFINITY;   // (LAZY, heavy tiles) distance of this lane's chunk box to the box of the tile's queries
  if (valid) {
    const float4* cd = reinterpret_cast<const float4*>(a.chunks + ch);
    b0 = cd[0]; b1 = cd[1];
    sbase = a.chunk_soa[ch];
    const float gx = fmaxf(fmaxf(b0.x - thx, tlx - b1.x), 0.f);
    const float gy = fmaxf(fmaxf(b0.y - thy, tly - b1.y), 0.f);
    const float gz = fmaxf(fmaxf(b0.z - thz, tlz - b1.z), 0.f);
    const float gd = gx * gx + gy * gy + gz * gz;
    pass = gd * kPruneShrink <= maxbest;
    if (LAZY) near = gd == 0.f;   // the chunk's box overlaps the box of the tile's own queries
    if (LAZY) gdl = gd;
  }
  unsigned long long m = __ballot(pass);
  if (!m) return;
  const unsigned long long nearm = LAZY ? __ballot(pass && near) : 0ull;
#ifdef LSGPU_KNN_STATS
  if (a.dbg_flags & 128) { n_surv += __popcll(m); return; }
#endif
  const float lim = ing ? prune_lim(fminf(best, ub), gap, cap2) : 0.f;  // bounds as of now; they only tighten
  bool refined = false;
  if (__popcll(m) > kRefineMin) {
    // Many boxes passed the group-level test: refine lane-parallel (lane = chunk) against every
    // query's own bound before the serial walk, which costs a broadcast + branch per chunk.
    bool needed = false;
    unsigned long long qm = __ballot(ing);
    while (qm) {
      const int u = __ffsll((long long)qm) - 1;
      qm &= qm - 1;
      const float ux = rl_f(qx, u), uy = rl_f(qy, u), uz = rl_f(qz, u);
      const float ul = rl_f(lim, u);
      needed = needed || (box_dist2(b0.x, b0.y, b0.z, b1.x, b1.y, b1.z, ux, uy, uz) * kPruneShrink <= ul);
    }
    m &= __ballot(pass && needed);
    if (!m) return;
    refined = true;
  }
  n_surv += __popcll(m);
  // ---- which survivors does any lane need (bounds as of now; they only tighten later)
  unsigned long long needm = refined ? m : 0;
  if (!refined) {
    unsigned long long mm = m;
    while (mm) {
      const int k = __ffsll((long long)mm) - 1;
      mm &= mm - 1;
      const float lx = rl_f(b0.x, k), ly = rl_f(b0.y, k), lz = rl_f(b0.z, k);
      const float hx = rl_f(b1.x, k), hy = rl_f(b1.y, k), hz = rl_f(b1.z, k);
      const bool need = ing && box_dist2(lx, ly, lz, hx, hy, hz, qx, qy, qz) * kPruneShrink <= lim;
      if (__ballot(need)) needm |= 1ull << k;
    }
  }
#ifdef LSGPU_KNN_STATS
  if (a.dbg_flags & (1 | 256)) { n_eval += __popcll(needm); return; }
#endif
  // ---- fetch + evaluate: two chunks per round, double buffered -- the LDS-DMA of the next pair is in
  // flight while the current pair is evaluated (slots 0,1 <-> 2,3)
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
#ifdef LSGPU_KNN_STATS
  const long long t_ev0 = clock64();
#endif
  uint32_t sa0 = 0, sa1 = 0, ca0 = 0, ca1 = 0, sb0 = 0, sb1 = 0, cb0 = 0, cb1 = 0;
  auto issue = [&](int slot, uint32_t& st, uint32_t& cnt) -> int {
    int k;
    if (LAZY) {
      for (;;) {
        unsigned long long pick = needm & nearm;
        if (!pick) pick = needm;
        if (!pick) return 0;
        k = __ffsll((long long)pick) - 1;
        if (sp.ways < 0 && !(needm & nearm)) {
          // a HEAVY tile (cell block above route_chunks, wide balls -- a large guess on an aggregated map evaluated 64 chunks per
          // tile in cell order): around the tile's own box the chunk NEAREST to it goes first -- the sooner the lanes' bounds
          // shrink, the more of the farther chunks the re-test below drops (32 per tile)
          const unsigned long long key = ((needm >> lane) & 1ull) ? (((unsigned long long)__float_as_uint(gdl) << 32) | (unsigned long long)lane) : ~0ull;
          k = __builtin_amdgcn_readfirstlane((int)(wave_min_u64(key) & 63ull));
        }
        needm &= ~(1ull << k);
        const float lx = rl_f(b0.x, k), ly = rl_f(b0.y, k), lz = rl_f(b0.z, k);
        const float hx = rl_f(b1.x, k), hy = rl_f(b1.y, k), hz = rl_f(b1.z, k);
        const bool need = ing && box_dist2(lx, ly, lz, hx, hy, hz, qx, qy, qz) * kPruneShrink <= prune_lim(fminf(best, ub), gap, cap2);
        if (__ballot(need)) break;
      }
    } else {
      if (!needm) return 0;
      k = __ffsll((long long)needm) - 1;
      needm &= needm - 1;
    }
    st = rl_u(__float_as_uint(b0.w), k);
    cnt = rl_u(__float_as_uint(b1.w), k);
    // the chunk's SoA block is 3 * cnt4 / 4 float4s (<= 48): one per lane, the other lanes stay out of it (nothing
    // reads the slot beyond the block)
    const uint32_t nf4 = 3u * ((cnt + 3u) >> 2);
    const float4* src = a.soa + rl_u(sbase, k) + (uint32_t)lane;
    if ((uint32_t)lane < nf4) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)&lds.slot[slot][0], 16, 0, 0);
    return 1;
  };
  int na = issue(0, sa0, ca0);
  na += issue(1, sa1, ca1);
  while (na) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // slots 2,3 are no longer being read
    int nb = issue(2, sb0, cb0);
    nb += issue(3, sb1, cb1);
    if (nb == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (nb == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    n_eval += na;
    if (SPLIT && sp.ways > 1) {
      tile_eval_slot_split(lds.slot[0], sa0, ca0, lane, sp);
      if (na > 1) tile_eval_slot_split(lds.slot[1], sa1, ca1, lane, sp);
    } else {
      tile_eval_slot(lds.slot[0], sa0, ca0, qx, qy, qz, best, sec, grp);
      if (na > 1) tile_eval_slot(lds.slot[1], sa1, ca1, qx, qy, qz, best, sec, grp);
    }
    if (!nb) break;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // slots 0,1 are no longer being read
    na = issue(0, sa0, ca0);
    na += issue(1, sa1, ca1);
    if (na == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (na == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    n_eval += nb;
    if (SPLIT && sp.ways > 1) {
      tile_eval_slot_split(lds.slot[2], sb0, cb0, lane, sp);
      if (nb > 1) tile_eval_slot_split(lds.slot[3], sb1, cb1, lane, sp);
    } else {
      tile_eval_slot(lds.slot[2], sb0, cb0, qx, qy, qz, best, sec, grp);
      if (nb > 1) tile_eval_slot(lds.slot[3], sb1, cb1, qx, qy, qz, best, sec, grp);
    }
  }
  if (SPLIT && sp.ways > 1) split_merge(sp, ing, __ballot(ing), lane, best, sec, grp);
  maxbest = wave_max(ing ? prune_lim(fminf(best, ub), gap, cap2) : 0.f);
#ifdef LSGPU_KNN_STATS
  c_eval += (uint32_t)(clock64() - t_ev0);
#else
  (void)c_eval;
#endif
}

// Which point of the recorded group of 4 is at distance `best` (first one; pts is padded, and a point
// of the following chunk at exactly the same distance would be an equally valid nearest neighbour).
// `s4` receives the second smallest distance inside that group (points of the following chunk that the
// group may run into are real reference points too, so the value stays a valid bound on "every other point").
__device__ __forceinline__ float4 tile_resolve_match(const KnnArgs& a, int grp, float qx, float qy, float qz,
                                                     float best, float4 mp, float& s4) {
  if (grp >= 0) {
    const float4 p0 = a.pts[grp], p1 = a.pts[grp + 1], p2 = a.pts[grp + 2], p3 = a.pts[grp + 3];
    const float e0 = dist2(qx - p0.x, qy - p0.y, qz - p0.z), e1 = dist2(qx - p1.x, qy - p1.y, qz - p1.z);
    const float e2 = dist2(qx - p2.x, qy - p2.y, qz - p2.z), e3 = dist2(qx - p3.x, qy - p3.y, qz - p3.z);
    if (e3 == best) mp = make_float4(p3.x, p3.y, p3.z, __int_as_float(grp + 3));
    if (e2 == best) mp = make_float4(p2.x, p2.y, p2.z, __int_as_float(grp + 2));
    if (e1 == best) mp = make_float4(p1.x, p1.y, p1.z, __int_as_float(grp + 1));
    if (e0 == best) mp = make_float4(p0.x, p0.y, p0.z, __int_as_float(grp));
    s4 = fminf(fmaxf(fminf(e0, e1), fminf(e2, e3)), fminf(fmaxf(e0, e1), fmaxf(e2, e3)));
  }
  return mp;
}

// ---------------------------------------------------------------- row-per-query search
// One DPP row of 16 lanes takes one query -- lanes 0..7 probe the <= 2x2x2 cells the ball touches, the cells' chunks
// are culled 16 at a time (lane = chunk) against the ball itself, a surviving chunk is evaluated one point per lane --
// and a wave runs four queries side by side.  Exact nearest point inside the cap, smallest index on ties; searches
// `gap` beyond the current bound like the tile search, so that the second smallest distance found (or the search
// radius) bounds "every other point" for the next iterations' keep test.  All 16 lanes of a row hold the same inputs.
constexpr int kRowqList = 64;  // chunk ids staged per row and window

__device__ __forceinline__ void rowq_search(const KnnArgs& a, float cap2s, float gap, uint32_t* list, int row, int k16,
                                            bool have, float qx, float qy, float qz, float ub, int id_in,
                                            unsigned long long& bestp_out, float& sec_out) {
  const GridDev& g = a.g;
  const int lim = (1 << (g.bits + g.fine)) - 1;
  unsigned long long bestp = have ? (((unsigned long long)__float_as_uint(ub) << 32) | (uint32_t)id_in) : ~0ull;
  float bcur = ub;          // the row's smallest distance so far
  float sec = INFINITY;     // this lane: smallest distance among the points it evaluated other than its own best
  float best = prune_lim(bcur, gap, cap2s);  // squared search radius
  // ---- the ball's cells: the level at which it spans at most two cells per axis
  uint32_t cs = 0, ce = 0;
  {
    const float B = sqrtf(best) * (1.0f + 1e-5f) + 1e-7f + kFineSlack * g.hf;
    const int flx = fine_coord(qx - B, g.ox, g.inv_hf, lim), fhx = fine_coord(qx + B, g.ox, g.inv_hf, lim);
    const int fly = fine_coord(qy - B, g.oy, g.inv_hf, lim), fhy = fine_coord(qy + B, g.oy, g.inv_hf, lim);
    const int flz = fine_coord(qz - B, g.oz, g.inv_hf, lim), fhz = fine_coord(qz + B, g.oz, g.inv_hf, lim);
    int l = 0;
    for (int sh = g.fine; l < g.bits; ++l, ++sh)
      if ((fhx >> sh) - (flx >> sh) < 2 && (fhy >> sh) - (fly >> sh) < 2 && (fhz >> sh) - (flz >> sh) < 2) break;
    unsigned long long todo = __ballot(have);
    while (todo) {  // rows may sit on different levels: one pass per distinct level keeps table base / mask scalar
      const int L = __builtin_amdgcn_readlane(l, __ffsll((long long)todo) - 1);
      const bool mine = have && l == L;
      todo &= ~__ballot(mine);
      if (mine && k16 < 8) {
        const int sh = g.fine + L;
        const int cx = (flx >> sh) + (k16 & 1), cy = (fly >> sh) + ((k16 >> 1) & 1), cz = (flz >> sh) + (k16 >> 2);
        if (cx <= (fhx >> sh) && cy <= (fhy >> sh) && cz <= (fhz >> sh))
          if (!grid_lookup(g, L, (uint32_t)cx, (uint32_t)cy, (uint32_t)cz, cs, ce)) { cs = 0; ce = 0; }
      }
    }
  }
  const uint32_t nch = ce - cs;
  const uint32_t incl = row_scan_incl_u32(nch), excl = incl - nch;
  const uint32_t tot = row_sum_u32(nch);
  const uint32_t totmax = wave_max_u32(tot);
  for (uint32_t wbase = 0; wbase < totmax; wbase += (uint32_t)kRowqList) {  // (one window unless the ball is huge)
    // this lane's chunks whose list position falls into the window
    for (uint32_t c = 0; c < nch; ++c) {
      const uint32_t pos = excl + c;
      if (pos >= wbase && pos < wbase + (uint32_t)kRowqList) list[pos - wbase] = cs + c;
    }
    const uint32_t wlen = tot > wbase ? (tot - wbase < (uint32_t)kRowqList ? tot - wbase : (uint32_t)kRowqList) : 0u;
    const uint32_t wmax = wave_max_u32(wlen);
    for (uint32_t e0 = 0; e0 < wmax; e0 += 16u) {
      const uint32_t e = e0 + (uint32_t)k16;
      float bd = INFINITY;
      uint32_t st = 0, cnt = 0;
      if (e < wlen) {
        const float4* cd = reinterpret_cast<const float4*>(a.chunks + list[e]);
        const float4 b0 = cd[0], b1 = cd[1];
        bd = box_dist2(b0.x, b0.y, b0.z, b1.x, b1.y, b1.z, qx, qy, qz) * kPruneShrink;
        st = __float_as_uint(b0.w); cnt = __float_as_uint(b1.w);
      }
      uint32_t m16 = (uint32_t)(__ballot(bd <= best) >> (row * 16)) & 0xFFFFu;
      while (__ballot(m16 != 0u)) {
        const bool has = m16 != 0u;
        const int src = row * 16 + (has ? __ffs((int)m16) - 1 : 0);
        m16 &= m16 - 1u;
        const float cbd = __shfl(bd, src, 64);
        const uint32_t cst = (uint32_t)__shfl((int)st, src, 64), ccnt = (uint32_t)__shfl((int)cnt, src, 64);
        float dmin = INFINITY;
        if (has && cbd <= best) {  // (the bound may have shrunk since the cull)
          for (uint32_t o = (uint32_t)k16; o < ccnt; o += 16u) {
            const float4 p = a.pts[cst + o];
            const float d = dist2(qx - p.x, qy - p.y, qz - p.z);
            const unsigned long long pk = ((unsigned long long)__float_as_uint(d) << 32) | (cst + o);
            if (pk < bestp) { sec = fminf(sec, __uint_as_float((uint32_t)(bestp >> 32))); bestp = pk; }
            else if (pk != bestp) sec = fminf(sec, d);   // (pk == bestp: the warm-start point itself)
            dmin = fminf(dmin, d);
          }
        }
        bcur = fminf(bcur, row_min(dmin));
        best = prune_lim(bcur, gap, cap2s);
      }
    }
  }
  // ---- the row's answer: smallest (distance, index) pair over its 16 lanes; every other lane's best is an "other"
  unsigned long long gb = bestp;
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    const unsigned long long w = __shfl_xor(gb, o, 64);
    gb = w < gb ? w : gb;
  }
  if (bestp != gb) sec = fminf(sec, __uint_as_float((uint32_t)(bestp >> 32)));
  sec_out = row_min(sec);
  bestp_out = gb;
}

__device__ __forceinline__ void sel_count_query(const KnnArgs& a, int j, uint32_t bits) {
  if (a.sel_below && (a.st->sel_mode || a.sel_force)) {  // predicted select: this query's share (see k_knn_tile)
    const uint32_t lo = a.st->sel_lo;
    if (bits < lo) atomicAdd(&a.sel_below[(j & (kSelBelowSlots - 1)) * kSelBelowStride], 1u);
    else if (bits - lo < a.st->sel_span) sel_count_inside(a, bits);
  }
}

// one lane per row: write the query's result, its share of the predicted select, its new lower bound
__device__ __forceinline__ void rowq_store(const KnnArgs& a, float cap2s, float gap, int j, unsigned long long bestp,
                                           float sec, int id_in, float lb_carried, bool write_all) {
  const int id = (int)(uint32_t)(bestp & 0xFFFFFFFFull);
  const float fd = __uint_as_float((uint32_t)(bestp >> 32));
  if (write_all || id != id_in) {
    const float4 p = a.pts[id];
    a.ids[j] = id;
    a.prev[j] = make_float4(p.x, p.y, p.z, __int_as_float(id));
  }
  a.d2[j] = fd;
  sel_count_query(a, j, (uint32_t)(bestp >> 32));
  if (a.lb) {  // every unevaluated point lies beyond the final search radius
    float nb = sqrtf(fminf(sec, prune_lim(fd, gap, cap2s))) * (1.0f - 1e-5f);
    if (id == id_in) nb = fmaxf(nb, lb_carried);
    a.lb[j] = nb;
  }
}

// Front rows: workgroup b of the first `front_blocks` takes a quarter (16 queries) of tile spread_list[b / 4] -- the
// whole per-query work of the tile kernel (transform, keep / far test, search, results) for the tiles whose queries
// share no candidates.  Every row first runs the prologue of its four queries (loads in flight together); the queries
// that have to search are packed into an LDS list and searched four at a time, one per row: a quarter with five
// searching queries costs two rounds of dependent round trips (about 8 us each), not four.  These waves start first,
// so their latency overlaps the rest of the launch instead of following it as a separate pass.
constexpr int kFrontMax = 8192;   // tiles the list holds
#ifndef LSGPU_FRONT_PER_TILE
#define LSGPU_FRONT_PER_TILE 4
#endif
// workgroups at the front of the grid per listed tile; measured on the benchmark pair: 1: 191, 2: 212, 4: 212, 8: 210,
// 16: 208 scans/s
constexpr int kFrontPerTile = LSGPU_FRONT_PER_TILE;
constexpr int kFrontRowQ = 16 / kFrontPerTile;       // queries per row of such a workgroup

__device__ __forceinline__ void tile_front_rows(const KnnArgs& a, uint32_t* lds_words /* >= 4 * kRowqList + 16 * 8 */,
                                                int lane, uint32_t b) {
  const uint32_t entry = b / (uint32_t)kFrontPerTile, sub = b % (uint32_t)kFrontPerTile;
  if (entry >= a.spread_cnt[0]) return;   // (entry < front_blocks / kFrontPerTile by construction)
  Mat34 T; float cap2;
  if (!iter_params(a.st, a.T, a.cap2, a.use_state_cap, T, cap2)) return;
  const float cap2s = cap2 * kCapSearchMargin2;
  const float gap = a.use_state_cap ? a.gap : 0.f;
  const int row = lane >> 4, k16 = lane & 15;
  uint32_t* list = lds_words + row * kRowqList;
  uint32_t* packed = lds_words + 4 * kRowqList;   // 16 entries x {j, qx, qy, qz, ub, id, lbn, -}
  const uint32_t tile = a.spread_list[entry];
  Mat34 To;
#pragma unroll
  for (int i = 0; i < 12; ++i) To.m[i] = a.st->T_rows_prev[i];
  // ---- prologue of the row's four queries (same arithmetic as k_knn_tile's), one after the other: this code shares
  // its register budget with the broadcast search (72 VGPRs, 7 waves per SIMD), arrays of four queries spilled there
  const int j0 = (int)(tile * 64u + sub * (uint32_t)(4 * kFrontRowQ)) + row * kFrontRowQ;
  uint32_t mine = 0;   // searching queries of this row: packed[row * 4 + 0 .. mine)
#pragma unroll 1
  for (int i = 0; i < kFrontRowQ; ++i) {
    const int j = j0 + i;
    if (j >= a.nq) break;
    const float4 rraw = a.rdq[j], mp = a.prev[j];
    const float lb_in = a.lb[j];
    const float3 q = xform(T, rraw.x, rraw.y, rraw.z);
    const float ub = dist2(q.x - mp.x, q.y - mp.y, q.z - mp.z);
    const float3 qo = xform(To, rraw.x, rraw.y, rraw.z);
    const float ddx = q.x - qo.x, ddy = q.y - qo.y, ddz = q.z - qo.z;
    const float delta = sqrtf(ddx * ddx + ddy * ddy + ddz * ddz) * (1.0f + 1e-5f) + 1e-7f;
    const float lbn = fmaxf(lb_in * (1.0f - 1e-6f) - delta, 0.f);
    const float lb2 = lbn * lbn;
    const bool keep = ub * (1.0f + 1e-5f) < lb2;
    const bool far = fminf(ub, lb2) > cap2 * (1.0f + 1e-5f);
    if (keep || far) {   // the match stands, only its distance moved
      if (k16 == 0) {
        a.d2[j] = ub;
        a.lb[j] = lbn;
        sel_count_query(a, j, __float_as_uint(ub));
      }
    } else {
      if (k16 == 0) {
        uint32_t* e = packed + ((uint32_t)row * (uint32_t)kFrontRowQ + mine) * 8u;
        e[0] = (uint32_t)j; e[1] = __float_as_uint(q.x); e[2] = __float_as_uint(q.y); e[3] = __float_as_uint(q.z);
        e[4] = __float_as_uint(ub); e[5] = (uint32_t)__float_as_int(mp.w); e[6] = __float_as_uint(lbn);
      }
      ++mine;
    }
  }
  // ---- the searching queries in (row, query) order: entry e lives in the segment of the row whose prefix range holds it
  const uint32_t c0 = rl_u(mine, 0), c1 = rl_u(mine, 16), c2 = rl_u(mine, 32), c3 = rl_u(mine, 48);
  const uint32_t total = c0 + c1 + c2 + c3;
  __syncthreads();   // (the workgroup is this one wave)
  for (uint32_t r0 = 0; r0 < total; r0 += 4u) {
    const uint32_t e_idx = r0 + (uint32_t)row;
    const bool have = e_idx < total;
    constexpr uint32_t Q = (uint32_t)kFrontRowQ;
    const uint32_t slot = e_idx < c0 ? e_idx : e_idx < c0 + c1 ? Q + (e_idx - c0)
                        : e_idx < c0 + c1 + c2 ? 2u * Q + (e_idx - c0 - c1) : 3u * Q + (e_idx - c0 - c1 - c2);
    const uint32_t* e = packed + (have ? slot : 0u) * 8u;
    const int j = (int)e[0];
    const float sx = __uint_as_float(e[1]), sy = __uint_as_float(e[2]), sz = __uint_as_float(e[3]);
    const float sub_ = __uint_as_float(e[4]);
    const int id_in = (int)e[5];
    const float slbn = __uint_as_float(e[6]);
    unsigned long long bestp; float sec;
    rowq_search(a, cap2s, gap, list, row, k16, have, sx, sy, sz, sub_, id_in, bestp, sec);
    if (have && k16 == 0) rowq_store(a, cap2s, gap, j, bestp, sec, id_in, slbn, a.write_all != 0);
  }
}

#ifndef LSGPU_TILE_OCC
#define LSGPU_TILE_OCC 7   // waves per SIMD the register budget is cut for (7: 72 VGPRs, no spills; 8 spills 48 B per lane)
#endif
#ifndef LSGPU_TILE_OCC_SPLIT
#define LSGPU_TILE_OCC_SPLIT 6   // the lane-split instantiation carries nine more live registers through the batch loop
#endif
template <int WAVES, bool LAZY = false, bool SPLIT = false>
__global__ __launch_bounds__(WAVES * 64, SPLIT ? LSGPU_TILE_OCC_SPLIT : LSGPU_TILE_OCC) void k_knn_tile(KnnArgs a) {
  static_assert(!(LAZY && SPLIT), "the lazy instantiation re-tests chunks against bounds the parts have not merged yet");
  __shared__ TileLds lds_all[WAVES];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  TileLds& lds = lds_all[w];
#ifdef LSGPU_KNN_STATS
  const long long t_begin = clock64();
#endif
  // ---- which tile.  Workgroup b runs on XCD b % 8 (observed dispatch order; speed only): give each
  // XCD a contiguous range of the Morton-ordered tiles so that neighbouring tiles, which read the
  // same reference chunks and hash entries, share one L2.
  const uint32_t wpb = blockDim.x >> 6;
  uint32_t blk = blockIdx.x;
  if (WAVES == 1 && a.front_blocks > 0) {
    if (blk < (uint32_t)a.front_blocks) {
      tile_front_rows(a, reinterpret_cast<uint32_t*>(lds.slot), lane, blk);
      return;
    }
    blk -= (uint32_t)a.front_blocks;
  }
#ifdef LSGPU_EXPERIMENTS
  else if (a.xcd_swizzle > 1) {
    // XCD x takes runs of `xcd_swizzle` consecutive blocks: run index = (i / S) * 8 + x.  Neighbouring
    // tiles share an L2 inside a run, while every XCD still gets an even mix of the whole scan.
    const uint32_t S = (uint32_t)a.xcd_swizzle, nb = gridDim.x, x = blk & 7u, i = blk >> 3;
    const uint32_t cand = ((i / S) * 8u + x) * S + (i % S);
    const uint32_t full = (nb / (8u * S)) * (8u * S);  // blocks beyond the last complete round keep their id
    if (blk < full) blk = cand;
  } else if (a.xcd_swizzle == 1) {
    const uint32_t nb = gridDim.x, q = nb >> 3, r = nb & 7u, x = blk & 7u, i = blk >> 3;
    blk = (x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q) + i;  // bijective for any grid size
  }
#endif
  const uint32_t tile = blk * wpb + w;
  if (tile >= (uint32_t)a.ntiles) return;
  // the wave's three coalesced loads go out before anything waits on the loop state (scalar loads + early exit)
  const int j = (int)(tile * 64u) + lane;
  const bool act = j < a.nq;
  float4 rraw = make_float4(0.f, 0.f, 0.f, 0.f), mp = rraw;  // query (own frame), current match (point + index)
  float lb_in = 0.f;
  if (act) {
    rraw = a.rdq[j];
    mp = a.prev[j];  // coalesced: no dependent gather of pts[prev]
    if (a.lb) lb_in = a.lb[j];
  }
  const int id_in = __float_as_int(mp.w);  // the match this query came in with
  const uint32_t on_list = a.spread_flag ? a.spread_flag[tile] : 0u;   // list position + 1
  if (on_list && a.front_blocks > 0) {   // the front rows own this tile if its entry is committed and inside this launch's front
    const uint32_t covered = min(a.spread_cnt[0], (uint32_t)a.front_blocks / (uint32_t)kFrontPerTile);
    if (on_list - 1u < covered) return;
  }
  // the tile's cached cell block (tag + 64 probe results) travels with the same round trip: whether it still fits
  // is only known after the reductions below, but waiting until then cost two more dependent loads (40 % of a
  // settled wave's time was this prologue)
  ulonglong2 tag_pre = make_ulonglong2(0ull, 0ull);
  uint2 cc_pre = make_uint2(0u, 0u);
  if (a.cell_cache) {
    tag_pre = a.cell_tags[tile];
    cc_pre = a.cell_cache[(size_t)tile * 64 + lane];
  }
  Mat34 T; float cap2;
  if (!iter_params(a.st, a.T, a.cap2, a.use_state_cap, T, cap2)) return;
  // Search / verification radius 5 % beyond the cap: a lane verified to have nothing inside keeps a
  // lower bound ABOVE the next iterations' caps, so it is skipped (farskip) instead of re-verified.
  const float cap2s = cap2 * kCapSearchMargin2;
#ifdef LSGPU_KNN_STATS
  if ((a.dbg_flags & 1024) && (tile & 1u)) return;
  if ((a.dbg_flags & 2048) && (tile & 3u)) return;
#endif
  const GridDev& g = a.g;
  uint32_t n_eval = 0, n_surv = 0, n_grp = 0, lvl_max = 0, c_eval = 0;

  // ub: distance to the warm-start point (prev match); best / sec: smallest and second smallest distance
  // among the points this search evaluates (the warm-start point is one of them whenever its chunk is)
  float qx = 0.f, qy = 0.f, qz = 0.f, ub = 0.f, best = INFINITY, sec = INFINITY;
  int bi = -1, grp = -1;
  if (act) {
    const float3 q = xform(T, rraw.x, rraw.y, rraw.z);
    qx = q.x; qy = q.y; qz = q.z;
    bi = __float_as_int(mp.w);
    ub = dist2(qx - mp.x, qy - mp.y, qz - mp.z);
  }
#ifdef LSGPU_KNN_STATS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t_loaded = clock64();
  long long t_red = t_loaded, t_look = t_loaded;
#endif
  // lb = lower bound on the distance to every reference point other than the warm-start point, carried
  // over from the previous iteration and reduced by this query's displacement (triangle inequality).
  //   keep: the warm-start point is provably still the unique nearest neighbour -> no search
  //   far : every point, the warm-start one included, is provably beyond the cap -> weight 0, no search
  float lbn = 0.f;
  bool skip = false;
  const float gap = a.use_state_cap ? a.gap : 0.f;
  if (act && a.lb && a.st && a.use_state_cap) {
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
  // only neighbours closer than the lane's search radius can matter
  const float R = sqrtf(prune_lim(ub, gap, cap2s)) * (1.0f + 1e-5f) + 1e-7f;
  const bool straggler = act && !skip && !(R <= a.r_cap);
  // (Tried and measured slower, 82 -> 84..88 us: letting a tile that has to search take its keep / far lanes along
  // with a margin sized from the displacement, so that whole tiles would be skipped in between -- the slack a lane keeps
  // is limited by the distance between its nearest and second nearest reference point, a few millimetres on densely
  // sampled surfaces, not by the search margin: most lanes have to search again after one or two iterations anyway.)
  const bool ing = act && !straggler && !skip;
  bool routed = false;
  const unsigned long long ing_mask = __ballot(ing);
  if (a.price.count) {   // (one launch per align)
    const unsigned long long hv = __ballot(ing && !(cone_price(a.price, qx, qy, qz, R) <= a.price.heavy));
    if (lane == 0 && ing_mask) {
      uint32_t* slot = a.price.count + (tile & (uint32_t)(kPriceSlots - 1)) * (uint32_t)kPriceStride;
      if (hv) atomicAdd(slot, (uint32_t)__popcll(hv));
      atomicAdd(slot + 1, (uint32_t)__popcll(ing_mask));
    }
  }
#ifdef LSGPU_EXPERIMENTS
  // Experiment (LSGPU_SPARSE_LANES): a wave with few searching lanes evaluates every candidate of its region for all 64
  // lanes (about 1400 vector instructions whatever the number of lanes that need them), so hand those lanes to the
  // row-per-query pass.  Measured: the tile kernel does get shorter (79 -> 57 us with <= 48 lanes handed over), but the
  // row pass pays ~11 us of dependent round trips per query and row, and thousands of waves appending to one list
  // counter serialise in the L2 -- a net loss at every threshold (kNN section).
  const bool sparse = a.sparse_lanes > 0 && __popcll(ing_mask) <= a.sparse_lanes;
  if (sparse) {
    routed = ing;
  } else
#endif
#ifdef LSGPU_KNN_STATS
  if (ing_mask && !(a.dbg_flags & 4)) {
#else
  if (ing_mask) {
#endif
    // query box, largest ball, largest bound of the wave
    const float tlx = wave_min(ing ? qx : INFINITY), thx = wave_max(ing ? qx : -INFINITY);
    const float tly = wave_min(ing ? qy : INFINITY), thy = wave_max(ing ? qy : -INFINITY);
    const float tlz = wave_min(ing ? qz : INFINITY), thz = wave_max(ing ? qz : -INFINITY);
    const float Rmax = wave_max(ing ? R : 0.f);
    float maxbest = wave_max(ing ? prune_lim(ub, gap, cap2s) : 0.f);
    const int lim = (1 << (g.bits + g.fine)) - 1;
    // fine-key box of the region (every lane's ball lies inside), widened by the rounding slack
    const float pad = Rmax + kFineSlack * g.hf;
    const int flx = __builtin_amdgcn_readfirstlane(fine_coord(tlx - pad, g.ox, g.inv_hf, lim));
    const int fly = __builtin_amdgcn_readfirstlane(fine_coord(tly - pad, g.oy, g.inv_hf, lim));
    const int flz = __builtin_amdgcn_readfirstlane(fine_coord(tlz - pad, g.oz, g.inv_hf, lim));
    const int fhx = __builtin_amdgcn_readfirstlane(fine_coord(thx + pad, g.ox, g.inv_hf, lim));
    const int fhy = __builtin_amdgcn_readfirstlane(fine_coord(thy + pad, g.oy, g.inv_hf, lim));
    const int fhz = __builtin_amdgcn_readfirstlane(fine_coord(thz + pad, g.oz, g.inv_hf, lim));
#ifdef LSGPU_KNN_STATS
    t_red = clock64();
#endif
    int l = 0, sh = g.fine;
    for (; l < g.bits; ++l, ++sh)
      if ((fhx >> sh) - (flx >> sh) < 4 && (fhy >> sh) - (fly >> sh) < 4 && (fhz >> sh) - (flz >> sh) < 4)
        break;
    sh = g.fine + l;
    const int x0 = flx >> sh, y0 = fly >> sh, z0 = flz >> sh;
    const int nx = (fhx >> sh) - x0 + 1, ny = (fhy >> sh) - y0 + 1, nz = (fhz >> sh) - z0 + 1;
#ifdef LSGPU_KNN_STATS
    if (a.dbg_flags & 512) { if (lane == 0 && nx + ny + nz == -7) a.d2[0] = 0.f; return; }
#endif
    // ---- one cell per lane.  A tile's cell block rarely changes between iterations (the queries move
    // by far less than a cell once ICP converges), so the 64 probe results are kept per tile, tagged
    // with the block they belong to: a hit replaces up to 64 random table probes by one coalesced read.
    uint32_t cs = 0, ce = 0;
    {
      const unsigned long long tag0 = ((unsigned long long)a.cache_gen << 32) | ((unsigned long long)l << 24) |
                                      ((unsigned long long)nx << 16) | ((unsigned long long)ny << 8) | (unsigned long long)nz;
      const unsigned long long tag1 = (unsigned long long)x0 | ((unsigned long long)y0 << 21) | ((unsigned long long)z0 << 42);
      const bool hit = a.cell_cache && tag_pre.x == tag0 && tag_pre.y == tag1;
      if (hit) {
        cs = cc_pre.x; ce = cc_pre.y;
      } else {
        const int cx = lane & 3, cy = (lane >> 2) & 3, cz = lane >> 4;
        if (cx < nx && cy < ny && cz < nz) {
          if (!grid_lookup(g, l, (uint32_t)(x0 + cx), (uint32_t)(y0 + cy), (uint32_t)(z0 + cz), cs, ce)) {
            cs = 0; ce = 0;
          }
        }
        if (a.cell_cache) {
          a.cell_cache[(size_t)tile * 64 + lane] = make_uint2(cs, ce);
          if (lane == 0) a.cell_tags[tile] = make_ulonglong2(tag0, tag1);
        }
      }
    }
#ifdef LSGPU_KNN_STATS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t_look = clock64();
#endif
    // A wave whose queries are spread far wider than their balls (sparse far field, Morton jumps)
    // shares no candidates: its lanes search on their own.
    const float ext = fmaxf(fmaxf(thx - tlx, thy - tly), thz - tlz);
    const uint32_t block_chunks = wave_sum_u32(ce - cs);
    const bool spread = ext > fmaxf(a.group_r, 4.f * Rmax) && block_chunks > (uint32_t)a.chunk_budget;
    if (spread && a.spread_list && !on_list && lane == 0) {   // remembered: the settled launches search it row-wise, up front
      const uint32_t idx = atomicAdd(&a.spread_cnt[1], 1u);
      if (idx < (uint32_t)kFrontMax) { a.spread_list[idx] = tile; a.spread_flag[tile] = idx + 1u; }
    }
    // a tile with wide balls and a heavy cell block (more than route_chunks chunks): a few such tiles are the tail of a launch
    // (one wave culling thousands of chunk boxes batch by batch) and go to the wave-per-query pass; when a launch has
    // thousands of them -- a 0.3 m guess on an aggregated local map -- that pass pays the map's density once per query
    // (3.8 ms for 468 k queries) where this kernel pays it once per 64, so only the first route_heavy_max tiles to ask go there
    bool heavy = Rmax > a.spread_route_r && !spread && block_chunks > (uint32_t)a.route_chunks;
    if (a.spread_route_r > 0.f && heavy && a.route_heavy_max >= 0) {
      uint32_t t = 0u;
      if (lane == 0) t = atomicAdd(a.strag_count + 2, 1u);   // (third word of the loop's per-iteration counters: zeroed by k_align_init, re-armed by k_normal_eq_loop)
      heavy = (uint32_t)__builtin_amdgcn_readfirstlane((int)t) < (uint32_t)a.route_heavy_max;
    }
#ifdef LSGPU_KNN_STATS
    if (spread && (a.dbg_flags & 32)) { /* ablation: drop spread waves */ } else
#endif
    if (a.spread_route_r > 0.f && (heavy || (spread && (Rmax > a.spread_route_r || block_chunks > (uint32_t)a.route_dense)))) {
      // wide balls and no shared candidates: 64 divergent per-lane searches would hold this wave for up
      // to a millisecond (the tail of the first launches); one wave per query (k_knn_fallback) instead
      routed = ing;
    } else if (spread) {
      if (ing) {  // tracks the exact index itself
        const int before = bi;
        best = ub;
        lane_ball_search(a, cap2s, qx, qy, qz, best, bi);
        if (bi != before) { const float4 p = a.pts[bi]; mp = make_float4(p.x, p.y, p.z, __int_as_float(bi)); }
      }
      n_grp = 64;
    } else {
      n_grp = 1; lvl_max = l;
#ifdef LSGPU_KNN_STATS
      if (a.dbg_flags & 64) return;
#endif
      SplitState sp;
      sp.ways = 1;
      if (SPLIT) split_setup(sp, ing_mask, lane, qx, qy, qz);
      if (LAZY && block_chunks > (uint32_t)a.route_chunks) sp.ways = -1;   // (the lazy instantiation has no lane split: the field marks a heavy tile)
      // ---- flatten the cells' chunk ranges into the LDS list, 64 at a time into the cull
      unsigned long long cells = __ballot(ce > cs);
      uint32_t fill = 0;
      while (cells) {
        const int c = __ffsll((long long)cells) - 1;
        cells &= cells - 1;
        const uint32_t ccs = rl_u(cs, c), cce = rl_u(ce, c);
        for (uint32_t base = ccs; base < cce; base += 64) {
          const uint32_t n = (cce - base) < 64u ? (cce - base) : 64u;
          if ((uint32_t)lane < n) lds.list[fill + lane] = base + lane;
          fill += n;
          while (fill >= 64u) {  // a full batch is ready
            fill -= 64u;
            const uint32_t ch = lds.list[fill + lane];
            tile_process_batch<LAZY, SPLIT>(a, cap2s, lds, lane, true, ch, ing, qx, qy, qz, tlx, tly, tlz, thx, thy, thz,
                               maxbest, ub, gap, best, sec, grp, n_eval, n_surv, c_eval, sp);
          }
        }
      }
      if (fill) {
        const bool v = (uint32_t)lane < fill;
        const uint32_t ch = v ? lds.list[lane] : 0u;
        tile_process_batch<LAZY, SPLIT>(a, cap2s, lds, lane, v, ch, ing, qx, qy, qz, tlx, tly, tlz, thx, thy, thz,
                           maxbest, ub, gap, best, sec, grp, n_eval, n_surv, c_eval, sp);
      }
    }
  }
#ifdef LSGPU_KNN_STATS
  if (a.dbg_flags & (64 | 128 | 256 | 512)) return;
  const long long t_loop_end = clock64();
#endif
  if (act) {
    float nb;  // new lower bound on the distance to every point other than the (new) match
    if (routed) {
      best = ub;   // the wave-per-query pass starts from the warm-start point and overwrites this result
      nb = lbn;
    } else if (n_grp == 64 && ing) {
      // per-lane search: exact neighbour inside the cap, or nothing there (match unchanged)
      nb = best <= cap2s ? sqrtf(best) * (1.0f - 1e-6f) : fmaxf(lbn, sqrtf(cap2s) * (1.0f - 1e-5f));
    } else if (!ing) {
      best = ub;   // keep / far / straggler (the fallback overwrites a straggler's result)
      nb = lbn;
    } else {
      const float lim_f = prune_lim(fminf(best, ub), gap, cap2s);  // every unevaluated point is beyond this
      if (best <= ub) {  // the evaluated minimum (the warm-start point itself unless something beat it)
        float s4 = INFINITY;
        mp = tile_resolve_match(a, grp, qx, qy, qz, best, mp, s4);
        float others = fminf(fminf(sec, s4), lim_f);
        bool same = __float_as_int(mp.w) == bi;
        if (sec == best || (!same && best == ub)) {  // a second point at exactly this distance: smallest index
          mp = canonical_tie(a, qx, qy, qz, best, mp);
          same = __float_as_int(mp.w) == bi;
        }
        if (!same) others = fminf(others, ub);  // (covers a warm-start point whose chunk was not needed)
        nb = sqrtf(others) * (1.0f - 1e-5f);
        if (same) nb = fmaxf(nb, lbn);
      } else {  // the warm-start point's chunk was beyond the cap and nothing closer exists
        nb = fmaxf(sqrtf(fminf(best, lim_f)) * (1.0f - 1e-5f), lbn);
        best = ub;
      }
    }
    // a settled iteration leaves most matches where they were: 20 of the 28 bytes a query used to write per launch
    // were its unchanged index and warm-start point
    if (a.write_all || __float_as_int(mp.w) != id_in) {
      a.ids[j] = __float_as_int(mp.w);
      a.prev[j] = mp;
    }
    a.d2[j] = best;
    if (a.lb) a.lb[j] = nb;
  }
  {  // hand-over list: one atomic per wave, the wave's queries stay neighbours in the list
    const unsigned long long hm = __ballot(act && (straggler || routed));
    if (hm) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(a.strag_count, (uint32_t)__popcll(hm));
      base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
      if (act && (straggler || routed)) a.strag[base + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull))] = (uint32_t)j;
    }
  }
  if (a.sel_below && (a.st->sel_mode || a.sel_force)) {
    // first two passes of the trimmed-distance select, folded into this kernel (every distance of the launch is
    // final here; the lanes routed to the wave-per-query pass are counted there)
    // (lanes handed to the wave-per-query pass get their final distance, and their count, there)
    const bool fin = act && !routed && !straggler;
    const uint32_t bits = __float_as_uint(best), lo = a.st->sel_lo;
    const unsigned long long below = __ballot(fin && bits < lo);
    if (fin && bits >= lo && bits - lo < a.st->sel_span) sel_count_inside(a, bits);
    if (lane == 0 && below) atomicAdd(&a.sel_below[(tile & (kSelBelowSlots - 1)) * kSelBelowStride], (uint32_t)__popcll(below));
  }
#ifdef LSGPU_KNN_STATS
  const uint32_t n_act = (uint32_t)__popcll(__ballot(ing));
  if (lane == 0 && a.dbg) {
    atomicAdd(&a.dbg[1], (unsigned long long)(t_loaded - t_begin)); atomicAdd(&a.dbg[2], (unsigned long long)(t_red - t_loaded));
    atomicAdd(&a.dbg[5], (unsigned long long)(t_look - t_red)); atomicAdd(&a.dbg[6], (unsigned long long)(clock64() - t_look));
    atomicAdd(&a.dbg[0], (unsigned long long)n_grp); atomicAdd(&a.dbg[3], (unsigned long long)n_surv);
    atomicAdd(&a.dbg[4], (unsigned long long)n_eval);
  }
  if (lane == 0 && a.dbg_wave) {
    const long long t_end = clock64();
    a.dbg_wave[tile] = make_uint4((uint32_t)(t_end - t_begin), n_eval, n_surv, (n_act << 16) | (n_grp << 8) | lvl_max);
    if (a.dbg_flags & 4096)  // second record per tile (the buffer holds 2 x ntiles records): where the cycles went
      a.dbg_wave[(uint32_t)a.ntiles + tile] = make_uint4((uint32_t)(t_look - t_begin), (uint32_t)(t_loop_end - t_look), c_eval,
                                                          (uint32_t)(t_end - t_loop_end));
  }
#else
  (void)n_grp; (void)lvl_max; (void)c_eval;
#endif
}

#ifdef LSGPU_EXPERIMENTS
// ---------------------------------------------------------------- lane-per-query search
__global__ __launch_bounds__(256) void k_knn_lane(KnnArgs a) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= a.nq) return;
  Mat34 T; float cap2;
  if (!iter_params(a.st, a.T, a.cap2, a.use_state_cap, T, cap2)) return;
  const float4 r = a.rdq[j];
  const float3 q = xform(T, r.x, r.y, r.z);
  float4 mp = a.prev[j];
  int bi = __float_as_int(mp.w);
  const int before = bi;
  float best = dist2(q.x - mp.x, q.y - mp.y, q.z - mp.z);
  lane_ball_search(a, cap2, q.x, q.y, q.z, best, bi);
  if (bi != before) { const float4 p = a.pts[bi]; mp = make_float4(p.x, p.y, p.z, __int_as_float(bi)); }
  a.ids[j] = bi;
  a.d2[j] = best;
  a.prev[j] = mp;
  if (a.lb) a.lb[j] = best <= cap2 ? sqrtf(best) * (1.0f - 1e-6f) : sqrtf(cap2) * (1.0f - 1e-5f);
}

#endif  // LSGPU_EXPERIMENTS

// ---------------------------------------------------------------- exact fallback, one wave per query
__global__ __launch_bounds__(256) void k_knn_fallback(KnnArgs a) {
  const int lane = threadIdx.x & 63;
  const uint32_t nw = gridDim.x * 4u;
  Mat34 T; float cap2;
  if (!iter_params(a.st, a.T, a.cap2, a.use_state_cap, T, cap2)) return;
  const float cap2s = cap2 * kCapSearchMargin2;  // (INF in uncapped launches)
  const uint32_t count = *a.strag_count;
  const GridDev& g = a.g;
  const int lim = (1 << (g.bits + g.fine)) - 1;
  for (uint32_t s = blockIdx.x * 4u + (threadIdx.x >> 6); s < count; s += nw) {
    const uint32_t j = a.strag[s];
    const float4 r = a.rdq[j];
    const float3 q = xform(T, r.x, r.y, r.z);
    const float ub = a.d2[j];  // finite: distance to the warm-start point (possibly improved)
    unsigned long long bestp =
        ((unsigned long long)__float_as_uint(ub) << 32) | (uint32_t)a.ids[j];
    float best = fminf(ub, cap2s);  // only neighbours inside the cap must be exact
    const float B = sqrtf(best) * (1.0f + 1e-5f) + 1e-7f + kFineSlack * g.hf;
    const int flx = fine_coord(q.x - B, g.ox, g.inv_hf, lim), fhx = fine_coord(q.x + B, g.ox, g.inv_hf, lim);
    const int fly = fine_coord(q.y - B, g.oy, g.inv_hf, lim), fhy = fine_coord(q.y + B, g.oy, g.inv_hf, lim);
    const int flz = fine_coord(q.z - B, g.oz, g.inv_hf, lim), fhz = fine_coord(q.z + B, g.oz, g.inv_hf, lim);
    int l = 0, sh = g.fine;
    for (; l < g.bits; ++l, ++sh)
      if ((fhx >> sh) - (flx >> sh) < 4 && (fhy >> sh) - (fly >> sh) < 4 && (fhz >> sh) - (flz >> sh) < 4)
        break;
    sh = g.fine + l;
    const int x0 = flx >> sh, y0 = fly >> sh, z0 = flz >> sh;
    const int nx = (fhx >> sh) - x0 + 1, ny = (fhy >> sh) - y0 + 1, nz = (fhz >> sh) - z0 + 1;
    uint32_t cs = 0, ce = 0;
    {
      const int cx = lane & 3, cy = (lane >> 2) & 3, cz = lane >> 4;
      if (cx < nx && cy < ny && cz < nz) {
        if (!grid_lookup(g, l, (uint32_t)(x0 + cx), (uint32_t)(y0 + cy), (uint32_t)(z0 + cz), cs, ce)) {
          cs = 0; ce = 0;
        }
      }
    }
    unsigned long long cells = __ballot(ce > cs);
    while (cells) {
      const int c = __ffsll((long long)cells) - 1;
      cells &= cells - 1;
      const uint32_t ccs = rl_u(cs, c), cce = rl_u(ce, c);
      for (uint32_t base = ccs; base < cce; base += 64) {
        const uint32_t ch = base + lane;
        float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
        float bd = INFINITY;   // (shrunk) distance from the query to this lane's chunk box
        bool live = ch < cce;  // this lane's chunk has not been evaluated yet
        if (live) {
          const float4* cd = reinterpret_cast<const float4*>(a.chunks + ch);
          b0 = cd[0]; b1 = cd[1];
          bd = box_dist2(b0.x, b0.y, b0.z, b1.x, b1.y, b1.z, q.x, q.y, q.z) * kPruneShrink;
        }
        // NEAREST BOX FIRST: every evaluated chunk is one dependent round trip, and the bound only shrinks when a closer
        // point turns up -- in cell order a 25 cm ball on an aggregated map walked ~100 chunks per query (6.6 ns per
        // query over a launch); the nearest box usually holds the neighbour and the re-test drops most of the others
        unsigned long long m = __ballot(live && bd <= best);
        while (m) {
          const unsigned long long key = (live && bd <= best) ? (((unsigned long long)__float_as_uint(bd) << 32) | (unsigned long long)lane) : ~0ull;
          const int k = __builtin_amdgcn_readfirstlane((int)(wave_min_u64(key) & 63ull));
          const uint32_t st = rl_u(__float_as_uint(b0.w), k);
          const uint32_t cnt = rl_u(__float_as_uint(b1.w), k);
          float d = INFINITY;
          if ((uint32_t)lane < cnt) {
            const float4 p = a.pts[st + lane];
            d = dist2(q.x - p.x, q.y - p.y, q.z - p.z);
            const unsigned long long pk = ((unsigned long long)__float_as_uint(d) << 32) | (st + lane);
            bestp = pk < bestp ? pk : bestp;
          }
          best = fminf(best, wave_min(d));
          if (lane == k) live = false;
          m = __ballot(live && bd <= best);   // (boxes at exactly the bound stay in: a point there at the same distance may carry the smaller index)
        }
      }
    }
    bestp = wave_min_u64(bestp);
    if (lane == 0) {
      const int id = (int)(uint32_t)(bestp & 0xFFFFFFFFull);
      const float4 p = a.pts[id];
      a.ids[j] = id;
      a.d2[j] = __uint_as_float((uint32_t)(bestp >> 32));
      if (a.sel_below && (a.st->sel_mode || a.sel_force)) {  // predicted select: this query's share (see k_knn_tile)
        const uint32_t bits = (uint32_t)(bestp >> 32), lo = a.st->sel_lo;
        if (bits < lo) atomicAdd(&a.sel_below[(j & (kSelBelowSlots - 1)) * kSelBelowStride], 1u);
        else if (bits - lo < a.st->sel_span) sel_count_inside(a, bits);
      }
      a.prev[j] = make_float4(p.x, p.y, p.z, __int_as_float(id));
      if (a.lb) {
        // every other point is at least as far as the neighbour found, or beyond the verified radius
        // (nothing inside the cap: the match did not change, its old bound still holds)
        const float fd = __uint_as_float((uint32_t)(bestp >> 32));
        a.lb[j] = fd <= cap2s ? sqrtf(fd) * (1.0f - 1e-6f) : fmaxf(a.lb[j], sqrtf(cap2s) * (1.0f - 1e-5f));
      }
    }
  }
}

// ---------------------------------------------------------------- row-per-query pass (settled launches)
// The queries a settled launch hands over (lanes of spread waves: a few thousand, small balls) do not need a whole
// wave each: one DPP row of 16 lanes takes one query -- lanes 0..7 probe the <= 2x2x2 cells the ball touches, the
// cells' chunks are culled 16 at a time (lane = chunk) against the ball itself, a surviving chunk is evaluated one
// point per lane -- and a wave runs four queries side by side.  Same results as k_knn_fallback (exact nearest point
// inside the cap, smallest index on ties, lower bound for the next iterations); three to five dependent memory
// round trips per query instead of one per chunk and per reduction of a 64-lane wave.
__global__ __launch_bounds__(256) void k_knn_rowq(KnnArgs a) {
  __shared__ uint32_t list_sh[16][kRowqList];
  const int lane = threadIdx.x & 63, row = lane >> 4, k16 = lane & 15, wave = threadIdx.x >> 6;
  uint32_t* list = list_sh[wave * 4 + row];
  Mat34 T; float cap2;
  if (!iter_params(a.st, a.T, a.cap2, a.use_state_cap, T, cap2)) return;
  const float cap2s = cap2 * kCapSearchMargin2;
  const float gap = a.use_state_cap ? a.gap : 0.f;
  const uint32_t count = *a.strag_count;
  for (uint32_t base = (blockIdx.x * 4u + (uint32_t)wave) * 4u; base < count; base += gridDim.x * 16u) {
    const uint32_t s = base + (uint32_t)row;
    const bool have = s < count;
    const uint32_t j = have ? a.strag[s] : 0u;
    float qx = 0.f, qy = 0.f, qz = 0.f, ub = 0.f;
    int id_in = -1;
    if (have) {
      const float4 r = a.rdq[j];
      const float3 q = xform(T, r.x, r.y, r.z);
      qx = q.x; qy = q.y; qz = q.z;
      ub = a.d2[j];  // distance to the warm-start point
      id_in = a.ids[j];
    }
    unsigned long long bestp; float sec;
    rowq_search(a, cap2s, gap, list, row, k16, have, qx, qy, qz, ub, id_in, bestp, sec);
    if (have && k16 == 0)
      rowq_store(a, cap2s, gap, (int)j, bestp, sec, id_in, a.lb ? a.lb[j] : 0.f /* the tile kernel left the carried bound there */, true);
  }
}

// ids (sorted-reference order, sorted-query order) -> caller order
__global__ __launch_bounds__(256) void k_knn_unpermute(const float4* __restrict__ rdq, int nq,
                                                       const int* __restrict__ ids,
                                                       const float* __restrict__ d2,
                                                       const float4* __restrict__ pts,
                                                       int* __restrict__ ids_out,
                                                       float* __restrict__ d2_out) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= nq) return;
  const uint32_t o = __float_as_uint(rdq[j].w);
  const int id = ids[j];
  ids_out[o] = id < 0 ? -1 : (int)__float_as_uint(pts[id].w);
  d2_out[o] = d2[j];
}

}  // namespace lsgpu
