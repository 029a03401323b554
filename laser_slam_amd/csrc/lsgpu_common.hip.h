// lsgpu_common.hip.h -- shared device helpers of the gfx950 ICP hot path (wave64, CDNA4).
//
// Kernel map (module chain of laser_slam/configurations/icp_default.yaml, executed by
// PointMatcher::ICP::compute at laser_slam/src/laser_track.cpp:496):
//   lsgpu_grid.hip.h   k_ref_* / k_chunk_* / k_cells_*   KDTreeMatcher::init          (yaml:9-12)
//   lsgpu_knn.hip.h    k_knn_seed / k_knn_tile / k_knn_fallback  ...::findClosests (knn 1, eps 0)
//   lsgpu_solve.hip.h  k_hist* / find_bin                TrimmedDistOutlierFilter     (yaml:14-16)
//                      k_normal_eq / k_ne_final          PointToPlaneErrorMinimizer   (yaml:18-19)
//   lsgpu_grid.hip.h   k_transform                       RigidTransformation::compute
//
// Shared arithmetic definitions (the CPU oracle uses the same, so ids / d2 / weights are
// bit-comparable):
//   transform : x' = fma(m02,z, fma(m01,y, fma(m00,x, m03)))
//   dist^2    : fma(dz,dz, fma(dy,dy, dx*dx))
// The whole TU is compiled with -ffp-contract=off: every fused op is written out.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lsgpu {

constexpr int kMaxLevels = 17;       // bits per axis <= 16
constexpr uint32_t kEmpty = 0xFFFFFFFFu;
constexpr int kHistBins = 2048;
#ifndef LSGPU_CHUNK_MAX
#define LSGPU_CHUNK_MAX 64
#endif
constexpr int kChunkMax = LSGPU_CHUNK_MAX;  // points per chunk (power of two, <= 64)
constexpr int kChunkGroup = 16;             // consecutive chunks that share one more bounding box (power of two)

struct HashEntry {  // 16 B: one dwordx4 per probe
  uint32_t xy;      // cell x | y << 16
  uint32_t z;
  uint32_t start;   // first chunk of the cell
  uint32_t end;     // one past its last chunk
};

// A chunk = up to 64 consecutive points of the Morton-sorted reference that lie in ONE level-0 cell,
// with their bounding box.  32 B: two dwordx4.
struct ChunkDesc {
  float lox, loy, loz;
  uint32_t start;   // first point
  float hix, hiy, hiz;
  uint32_t count;   // 1..64
};

// Voxel-hash pyramid over the Morton-sorted reference.  Keys are quantised at hf = h0 / 2^fine;
// level l (0..bits) has cell edge h0 * 2^l and a level-l cell is a contiguous range of chunks.
// Level `bits` is one cell holding every point.
struct GridDev {
  float ox, oy, oz;   // origin (reference-mean frame)
  float inv_hf, hf;   // key quantisation step
  float h0;           // level-0 cell edge
  int fine;           // key bits per axis below level 0
  int bits;           // level-0 cells per axis = 2^bits
  const HashEntry* tab[kMaxLevels];
  uint32_t mask[kMaxLevels];
};

// slack (in fine-key units) that covers float rounding of (c - o) * inv_hf at 16-bit magnitudes
constexpr float kFineSlack = 0.0625f;

// tile kNN: batches with more surviving boxes than this get the lane-parallel per-query refinement
#ifndef LSGPU_REFINE_MIN
#define LSGPU_REFINE_MIN 48
#endif
constexpr int kRefineMin = LSGPU_REFINE_MIN;

__device__ __forceinline__ int fine_coord(float c, float o, float inv_hf, int lim) {
  const float t = floorf((c - o) * inv_hf);
  return (int)fminf(fmaxf(t, 0.f), (float)lim);
}

struct Mat34 {  // rows of a rigid transform
  float m[12];  // m[r*4+c]
};

// Loop state of one align, resident in HBM: written by k_icp_update (one wave per iteration), read by
// every kernel of the following iteration, so that iterations are enqueued back to back with no
// host round trip.  done != 0 turns every later launch into an immediate exit.
struct IcpState {
  float T_iter[16];   // column major (PointMatcher TransformationParameters)
  float T_rows[12];   // the same transform as Mat34 rows (what the kernels load)
  float T_rows_prev[12];  // T_iter of the previous iteration (query displacement bound)
  float prev_limit;   // trim limit of the last completed iteration
  float cap2;         // search cap of the next capped kNN launch (INF: none)
  int iter;           // completed iterations
  int done;
  int status;         // 0 ok; 1 no convergence; 100 = cap prediction failed, host repeats uncapped
  int err_code;       // 1 no point to minimize, 2 normal matrix not positive definite, 3 NaN in checker
  int converged;      // stopped by the differential checker
  int counter, n_hist;  // checker state
  int cap_enabled;
  int max_iter, smooth;
  float lim_rot, lim_trans;
  unsigned long long stragglers;
  // predicted select: once the trim limit keeps its top 12 bits from one iteration to the next (sel_mode = 1) the
  // kNN kernel itself counts the distances below that bin and histograms the ones inside it, which replaces the
  // first two passes of the radix select; verified afterwards (k_hist_refine<3>), failure repeats the iteration
  uint32_t sel_bin1;  // top 12 bits of the last limit
  int sel_mode;
  // committed select: once the limit has stayed within kSelStreakBins second-level bins of its predecessor for a few
  // iterations, the host stops launching the select kernels altogether; the search kernels then also histogram the
  // last 9 bits of the distances that fall into a window of second-level bins around the last limit, and the normal-
  // equation kernel reads the order statistic from those tables in its prologue (verified; a miss repeats the
  // iteration's select in full)
  uint32_t sel_bin2;  // bits [19:9] of the last limit
  int sel_streak;     // consecutive iterations whose limit stayed in the same 12-bit bin and close in the second level
  // front rows of the tile kernel: tiles on the spread list as of the last completed iteration (the host sizes the
  // front of the grid from the copy it fetches with the rest of the state)
  uint32_t n_spread;
  // fused select (round 6, sel_wide = 1; not in the split-scan mode).  The distances' bit patterns are cut into SLICES of
  // 2^kSelSliceShift steps (a relative width of 1.2e-4 .. 2.4e-4).  The normal equations' sum is DEFINED as: every inlier
  // outside the slice that holds the limit, in block / thread order, plus the inliers inside that slice in query order
  // (at most kSelAmbCap of them; a fuller slice is summed in place) -- whichever way the limit was found.  So the
  // normal-equation kernel can find the limit itself: the search kernels count the distances below a window of
  // kSelSlices slices that starts at 0.7 x the last limit (one octave) and histogram those inside it by slice; the
  // kernel's prologue finds the slice that holds the order statistic, its main pass takes everything below that slice as an
  // inlier and sets the slice's few dozen distances aside with their contributions, its last block ranks them: the one of
  // the remaining rank is the limit, those up to it are added.  No select launch, no window table; valid while the limit
  // stays inside [0.7, 1.4] x its predecessor
  uint32_t sel_lo;    // bit pattern of the window's lower edge (aligned mode: sel_bin1 << 20)
  uint32_t sel_span;  // ... its width in bit steps (aligned mode: 2^20)
  int sel_shift;      // ... log2 of a slice's width (aligned mode: 9)
  int sel_wide;       // set by the host at the start of an align
  int sel_fails;      // iterations the fused / predicted select voided; after the second the alignment keeps to the select kernels
  // the differential checker's two smoothed changes of the last completed iteration (0: history still short): the host
  // estimates from them how many iterations are left and does not enqueue a full group of launches in front of the end
  float chk_rot, chk_trans;
  float chk_rot_prev, chk_trans_prev;   // ... and of the iteration before it (a first look has no earlier look to compare with)
  // the last completed iteration's trim limit and inlier count (the alignment's statistics; the per-iteration trace stays on
  // the device until somebody asks for it)
  float last_limit;
  uint32_t pad2_;
  long long last_used;
};
constexpr int kSelBelowSlots = 64;   // counters of "distance below the predicted bin", hashed by tile ...
constexpr int kSelBelowStride = 32;  // ... one per 128-byte line (atomics on one line serialise in L2)
constexpr int kSelFailFlag = kSelBelowSlots * kSelBelowStride;  // word index of the failure flag
constexpr int kSelWinRows = 128;     // committed select: second-level bins covered by the window table ...
constexpr int kSelWinHalf = 64;      // ... centred on the last limit's bin (relative width of a bin: 2^-14)
constexpr int kSelStreakBins = 40;   // a limit that moved less than this many second-level bins counts as "stayed"
constexpr int kSelSliceShift = 11;   // fused select: a slice is 2^11 bit steps of the distance ...
constexpr int kSelSlices = 2 * kHistBins;   // ... the window 4096 of them = one octave (counted into the select's second and third table)
constexpr int kSelAmbCap = 256;      // ... and at most this many distances of the limit's slice are set aside (a fuller slice: summed in place, select in full)
constexpr float kSelWideLo = 0.7f;   // the window starts at this x the last limit ...
constexpr float kSelArmLo = 0.75f, kSelArmHi = 1.2f;     // ... and is used once a limit has moved by no more than this from its predecessor
constexpr int kStatusCapFailed = 100;
constexpr int kStatusSelFailed = 101;  // predicted select missed: the host repeats select + normal equations only

// T and cap of this launch: from the kernel arguments, or from the loop state.  false => exit now.
__device__ __forceinline__ bool iter_params(const IcpState* __restrict__ st, const Mat34& T_arg,
                                            float cap2_arg, int use_state_cap, Mat34& T, float& cap2) {
  T = T_arg; cap2 = cap2_arg;
  if (!st) return true;
  if (st->done) return false;
#pragma unroll
  for (int i = 0; i < 12; ++i) T.m[i] = st->T_rows[i];
  if (use_state_cap) cap2 = st->cap2;
  return true;
}

__device__ __forceinline__ float3 xform(const Mat34& T, float x, float y, float z) {
  float3 o;
  o.x = __fmaf_rn(T.m[2], z, __fmaf_rn(T.m[1], y, __fmaf_rn(T.m[0], x, T.m[3])));
  o.y = __fmaf_rn(T.m[6], z, __fmaf_rn(T.m[5], y, __fmaf_rn(T.m[4], x, T.m[7])));
  o.z = __fmaf_rn(T.m[10], z, __fmaf_rn(T.m[9], y, __fmaf_rn(T.m[8], x, T.m[11])));
  return o;
}

__device__ __forceinline__ float dist2(float dx, float dy, float dz) {
  return __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, dx * dx));
}

// ---------------------------------------------------------------- Morton helpers (<= 21 bits/axis)
__host__ __device__ __forceinline__ uint64_t spread3(uint32_t v) {
  uint64_t x = v & 0x1FFFFFull;
  x = (x | x << 32) & 0x1F00000000FFFFull;
  x = (x | x << 16) & 0x1F0000FF0000FFull;
  x = (x | x << 8) & 0x100F00F00F00F00Full;
  x = (x | x << 4) & 0x10C30C30C30C30C3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}
__host__ __device__ __forceinline__ uint32_t compact3(uint64_t x) {
  x &= 0x1249249249249249ull;
  x = (x ^ (x >> 2)) & 0x10C30C30C30C30C3ull;
  x = (x ^ (x >> 4)) & 0x100F00F00F00F00Full;
  x = (x ^ (x >> 8)) & 0x1F0000FF0000FFull;
  x = (x ^ (x >> 16)) & 0x1F00000000FFFFull;
  x = (x ^ (x >> 32)) & 0x1FFFFFull;
  return (uint32_t)x;
}
__host__ __device__ __forceinline__ uint64_t morton3(uint32_t x, uint32_t y, uint32_t z) {
  return spread3(x) | (spread3(y) << 1) | (spread3(z) << 2);
}

// (A sum of products followed by two multiply-shift rounds.  The classic (x p1) ^ (y p2) ^ (z p3) is LINEAR in the low
// bits the table mask keeps, and the cells of a scan are surfaces: on the benchmark scan its probe chains at the third
// pyramid level reached 61 slots at a load of 0.4 (mean 2.85) -- k_cells_fill, whose inserts are one device-scope CAS
// round trip per probed slot, took 60 us for that level alone -- against 7 (mean 1.13) with this mix at a load of 0.2.)
__device__ __forceinline__ uint32_t cell_hash(uint32_t x, uint32_t y, uint32_t z) {
  uint32_t h = x * 0x9E3779B1u + y * 0x85EBCA77u + z * 0xC2B2AE3Du;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}

__device__ __forceinline__ bool grid_lookup(const GridDev& g, int l, uint32_t x, uint32_t y,
                                            uint32_t z, uint32_t& s, uint32_t& e) {
  const uint32_t xy = x | (y << 16);
  const uint32_t mask = g.mask[l];
  const uint4* t = reinterpret_cast<const uint4*>(g.tab[l]);
  uint32_t slot = cell_hash(x, y, z) & mask;
  for (;;) {
    const uint4 en = t[slot];  // one dwordx4; the combined compare keeps it a single load
    if (((en.x ^ xy) | (en.y ^ z)) == 0u) { s = en.z; e = en.w; return true; }
    if (en.x == kEmpty) return false;
    slot = (slot + 1) & mask;
  }
}

// ---------------------------------------------------------------- wave / block reductions
// DPP row reductions (no LDS traffic): quad_perm xor1, xor2, row_half_mirror, row_mirror leave the
// row result in all 16 lanes of each row; the 4 row results are combined through readlane.
#define LSGPU_DPP(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xF, 0xF, true)

__device__ __forceinline__ float rl_f(float v, int lane) {  // uniform-lane broadcast (v_readlane)
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ uint32_t rl_u(uint32_t v, int lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}

__device__ __forceinline__ float wave_min(float v) {
  v = fminf(v, __int_as_float(LSGPU_DPP(__float_as_int(v), 0xB1)));
  v = fminf(v, __int_as_float(LSGPU_DPP(__float_as_int(v), 0x4E)));
  v = fminf(v, __int_as_float(LSGPU_DPP(__float_as_int(v), 0x141)));
  v = fminf(v, __int_as_float(LSGPU_DPP(__float_as_int(v), 0x140)));
  return fminf(fminf(rl_f(v, 0), rl_f(v, 16)), fminf(rl_f(v, 32), rl_f(v, 48)));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, __int_as_float(LSGPU_DPP(__float_as_int(v), 0xB1)));
  v = fmaxf(v, __int_as_float(LSGPU_DPP(__float_as_int(v), 0x4E)));
  v = fmaxf(v, __int_as_float(LSGPU_DPP(__float_as_int(v), 0x141)));
  v = fmaxf(v, __int_as_float(LSGPU_DPP(__float_as_int(v), 0x140)));
  return fmaxf(fmaxf(rl_f(v, 0), rl_f(v, 16)), fmaxf(rl_f(v, 32), rl_f(v, 48)));
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
  v += (uint32_t)LSGPU_DPP((int)v, 0xB1);
  v += (uint32_t)LSGPU_DPP((int)v, 0x4E);
  v += (uint32_t)LSGPU_DPP((int)v, 0x141);
  v += (uint32_t)LSGPU_DPP((int)v, 0x140);
  return rl_u(v, 0) + rl_u(v, 16) + rl_u(v, 32) + rl_u(v, 48);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  v = max(v, (uint32_t)LSGPU_DPP((int)v, 0xB1));
  v = max(v, (uint32_t)LSGPU_DPP((int)v, 0x4E));
  v = max(v, (uint32_t)LSGPU_DPP((int)v, 0x141));
  v = max(v, (uint32_t)LSGPU_DPP((int)v, 0x140));
  return max(max(rl_u(v, 0), rl_u(v, 16)), max(rl_u(v, 32), rl_u(v, 48)));
}
// ---- reductions over one DPP row (16 lanes): the result is left in all 16 lanes of each row
__device__ __forceinline__ float row_min(float v) {
  v = fminf(v, __int_as_float(LSGPU_DPP(__float_as_int(v), 0xB1)));
  v = fminf(v, __int_as_float(LSGPU_DPP(__float_as_int(v), 0x4E)));
  v = fminf(v, __int_as_float(LSGPU_DPP(__float_as_int(v), 0x141)));
  return fminf(v, __int_as_float(LSGPU_DPP(__float_as_int(v), 0x140)));
}
__device__ __forceinline__ float row_max(float v) {
  v = fmaxf(v, __int_as_float(LSGPU_DPP(__float_as_int(v), 0xB1)));
  v = fmaxf(v, __int_as_float(LSGPU_DPP(__float_as_int(v), 0x4E)));
  v = fmaxf(v, __int_as_float(LSGPU_DPP(__float_as_int(v), 0x141)));
  return fmaxf(v, __int_as_float(LSGPU_DPP(__float_as_int(v), 0x140)));
}
__device__ __forceinline__ uint32_t row_sum_u32(uint32_t v) {
  v += (uint32_t)LSGPU_DPP((int)v, 0xB1);
  v += (uint32_t)LSGPU_DPP((int)v, 0x4E);
  v += (uint32_t)LSGPU_DPP((int)v, 0x141);
  return v + (uint32_t)LSGPU_DPP((int)v, 0x140);
}
// inclusive prefix sum inside each row (row_shr:n with zero fill for the lanes that have no source)
__device__ __forceinline__ uint32_t row_scan_incl_u32(uint32_t v) {
  v += (uint32_t)LSGPU_DPP((int)v, 0x111);
  v += (uint32_t)LSGPU_DPP((int)v, 0x112);
  v += (uint32_t)LSGPU_DPP((int)v, 0x114);
  return v + (uint32_t)LSGPU_DPP((int)v, 0x118);
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long w = __shfl_xor(v, o, 64);
    v = w < v ? w : v;
  }
  return v;
}

}  // namespace lsgpu
