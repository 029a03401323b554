// lsgpu_kernels.hip.h -- gfx950 device code of the ICP hot path (wave64, CDNA4).
//
// What each kernel stands in for (module chain of laser_slam/configurations/icp_default.yaml,
// executed by PointMatcher::ICP::compute at laser_slam/src/laser_track.cpp:496):
//   k_ref_stats / k_ref_keys / k_ref_gather / k_cells_*   KDTreeMatcher::init        (yaml:9-12)
//   k_knn_main / k_knn_fallback                           KDTreeMatcher::findClosests (knn 1, eps 0)
//   k_hist* / find_bin                                    TrimmedDistOutlierFilter    (yaml:14-16)
//   k_normal_eq / k_ne_final                              PointToPlaneErrorMinimizer  (yaml:18-19)
//   k_transform                                           RigidTransformation::compute
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

struct HashEntry {  // 16 B: one dwordx4 per probe
  uint32_t xy;      // cell x | y << 16
  uint32_t z;
  uint32_t start;   // first point (Morton-sorted reference order)
  uint32_t end;     // one past last
};

// Voxel-hash pyramid over the Morton-sorted reference.  Level l has cell edge h0 * 2^l; a level-l
// cell is a contiguous range of the sorted array.  Level `bits` is a single cell (all points).
struct GridDev {
  float ox, oy, oz;   // origin (reference-mean frame)
  float inv_h0, h0;
  int bits;           // cells per axis at level 0 = 2^bits
  const HashEntry* tab[kMaxLevels];
  uint32_t mask[kMaxLevels];
};

struct Mat34 {  // rows of a rigid transform
  float m[12];  // m[r*4+c]
};

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

__device__ __forceinline__ uint32_t cell_hash(uint32_t x, uint32_t y, uint32_t z) {
  return (x * 73856093u) ^ (y * 19349663u) ^ (z * 83492791u);
}

__device__ __forceinline__ bool grid_lookup(const GridDev& g, int l, uint32_t x, uint32_t y,
                                            uint32_t z, uint32_t& s, uint32_t& e) {
  const uint32_t xy = x | (y << 16);
  const uint32_t mask = g.mask[l];
  const uint4* t = reinterpret_cast<const uint4*>(g.tab[l]);
  uint32_t slot = cell_hash(x, y, z) & mask;
  for (;;) {
    const uint4 en = t[slot];
    if (en.x == xy && en.y == z) { s = en.z; e = en.w; return true; }
    if (en.x == kEmpty) return false;
    slot = (slot + 1) & mask;
  }
}

// ---------------------------------------------------------------- wave / block reductions
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
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

// ---------------------------------------------------------------- reference statistics
// partials[block] = {sum x, sum y, sum z (double), min xyz, max xyz (float bits in double slots)}
struct RefStats {
  double sum[3];
  float mn[3];
  float mx[3];
};

__global__ __launch_bounds__(256) void k_ref_stats(const float4* __restrict__ in, int64_t n,
                                                   RefStats* __restrict__ partials) {
  double sx = 0, sy = 0, sz = 0;
  float mnx = INFINITY, mny = INFINITY, mnz = INFINITY, mxx = -INFINITY, mxy = -INFINITY,
        mxz = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float4 p = in[i];
    sx += p.x; sy += p.y; sz += p.z;
    mnx = fminf(mnx, p.x); mny = fminf(mny, p.y); mnz = fminf(mnz, p.z);
    mxx = fmaxf(mxx, p.x); mxy = fmaxf(mxy, p.y); mxz = fmaxf(mxz, p.z);
  }
  sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
  mnx = wave_min(mnx); mny = wave_min(mny); mnz = wave_min(mnz);
  mxx = wave_max(mxx); mxy = wave_max(mxy); mxz = wave_max(mxz);
  __shared__ RefStats sh[4];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sh[w].sum[0] = sx; sh[w].sum[1] = sy; sh[w].sum[2] = sz;
    sh[w].mn[0] = mnx; sh[w].mn[1] = mny; sh[w].mn[2] = mnz;
    sh[w].mx[0] = mxx; sh[w].mx[1] = mxy; sh[w].mx[2] = mxz;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    RefStats r = sh[0];
    for (int k = 1; k < 4; ++k)
      for (int d = 0; d < 3; ++d) {
        r.sum[d] += sh[k].sum[d];
        r.mn[d] = fminf(r.mn[d], sh[k].mn[d]);
        r.mx[d] = fmaxf(r.mx[d], sh[k].mx[d]);
      }
    partials[blockIdx.x] = r;
  }
}

// One block; fixed order => deterministic mean.
__global__ __launch_bounds__(64) void k_ref_stats_final(const RefStats* __restrict__ partials,
                                                        int nblocks, RefStats* __restrict__ out) {
  if (threadIdx.x != 0) return;
  RefStats r = partials[0];
  for (int k = 1; k < nblocks; ++k)
    for (int d = 0; d < 3; ++d) {
      r.sum[d] += partials[k].sum[d];
      r.mn[d] = fminf(r.mn[d], partials[k].mn[d]);
      r.mx[d] = fmaxf(r.mx[d], partials[k].mx[d]);
    }
  *out = r;
}

// ---------------------------------------------------------------- keys
// Reference: centre on the mean, quantise to level-0 cells, Morton key.
__global__ __launch_bounds__(256) void k_ref_keys(const float4* __restrict__ in, int64_t n,
                                                  float mx, float my, float mz, GridDev g,
                                                  uint64_t* __restrict__ keys,
                                                  uint32_t* __restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 p = in[i];
  const float cx = p.x - mx, cy = p.y - my, cz = p.z - mz;
  const int lim = (1 << g.bits) - 1;
  int ix = (int)floorf((cx - g.ox) * g.inv_h0);
  int iy = (int)floorf((cy - g.oy) * g.inv_h0);
  int iz = (int)floorf((cz - g.oz) * g.inv_h0);
  ix = min(max(ix, 0), lim); iy = min(max(iy, 0), lim); iz = min(max(iz, 0), lim);
  keys[i] = morton3((uint32_t)ix, (uint32_t)iy, (uint32_t)iz);
  vals[i] = (uint32_t)i;
}

// Reading: coarse Morton key in its own frame, only to make consecutive lanes spatially coherent.
__global__ __launch_bounds__(256) void k_query_keys(const float4* __restrict__ in, int64_t n,
                                                    float inv_h, int bits,
                                                    uint64_t* __restrict__ keys,
                                                    uint32_t* __restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 p = in[i];
  const int lim = (1 << bits) - 1, half = 1 << (bits - 1);
  int ix = (int)floorf(p.x * inv_h) + half;
  int iy = (int)floorf(p.y * inv_h) + half;
  int iz = (int)floorf(p.z * inv_h) + half;
  ix = min(max(ix, 0), lim); iy = min(max(iy, 0), lim); iz = min(max(iz, 0), lim);
  keys[i] = morton3((uint32_t)ix, (uint32_t)iy, (uint32_t)iz);
  vals[i] = (uint32_t)i;
}

// sorted reference: pts[j] = {centred xyz, original index}, nrm[j] = {normal, 0}, inv[orig] = j
__global__ __launch_bounds__(256) void k_ref_gather(const float4* __restrict__ in,
                                                    const float* __restrict__ nrm_in, int64_t n,
                                                    const uint32_t* __restrict__ perm, float mx,
                                                    float my, float mz, float4* __restrict__ pts,
                                                    float4* __restrict__ nrm,
                                                    uint32_t* __restrict__ inv) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const uint32_t s = perm[j];
  const float4 p = in[s];
  pts[j] = make_float4(p.x - mx, p.y - my, p.z - mz, __uint_as_float(s));
  float4 nn = make_float4(0.f, 0.f, 0.f, 0.f);
  if (nrm_in) { nn.x = nrm_in[3 * (int64_t)s]; nn.y = nrm_in[3 * (int64_t)s + 1]; nn.z = nrm_in[3 * (int64_t)s + 2]; }
  nrm[j] = nn;
  inv[s] = (uint32_t)j;
}

// sorted reading moved by T (step 5 of ICP::compute); w = original index
__global__ __launch_bounds__(256) void k_query_gather(const float4* __restrict__ in, int64_t n,
                                                      const uint32_t* __restrict__ perm, Mat34 T,
                                                      float4* __restrict__ out) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const uint32_t s = perm[j];
  const float4 p = in[s];
  const float3 q = xform(T, p.x, p.y, p.z);
  out[j] = make_float4(q.x, q.y, q.z, __uint_as_float(s));
}

__global__ __launch_bounds__(256) void k_transform(const float4* __restrict__ in, int64_t n,
                                                   Mat34 T, float4* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 p = in[i];
  const float3 q = xform(T, p.x, p.y, p.z);
  out[i] = make_float4(q.x, q.y, q.z, p.w);
}

// ---------------------------------------------------------------- cell tables
// A level-l cell boundary sits between sorted points i-1 and i when their keys differ above bit 3l.
__device__ __forceinline__ int boundary_levels(uint64_t k, uint64_t kp) {
  const uint64_t x = k ^ kp;
  if (x == 0) return -1;
  return (63 - __clzll((long long)x)) / 3;  // levels 0..that have a boundary
}

__global__ __launch_bounds__(256) void k_cells_count(const uint64_t* __restrict__ keys, int64_t n,
                                                     int bits, uint32_t* __restrict__ counts) {
  __shared__ uint32_t sh[kMaxLevels];
  if (threadIdx.x < kMaxLevels) sh[threadIdx.x] = 0;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    int lv = (i == 0) ? bits : boundary_levels(keys[i], keys[i - 1]);
    if (lv > bits) lv = bits;
    for (int l = 0; l <= lv; ++l) atomicAdd(&sh[l], 1u);
  }
  __syncthreads();
  if (threadIdx.x <= bits && sh[threadIdx.x]) atomicAdd(&counts[threadIdx.x], sh[threadIdx.x]);
}

__device__ __forceinline__ HashEntry* table_slot(HashEntry* tab, uint32_t mask, uint32_t x,
                                                 uint32_t y, uint32_t z) {
  const uint32_t xy = x | (y << 16);
  const unsigned long long want = (unsigned long long)xy | ((unsigned long long)z << 32);
  uint32_t slot = cell_hash(x, y, z) & mask;
  for (;;) {
    unsigned long long* kp = reinterpret_cast<unsigned long long*>(&tab[slot]);
    const unsigned long long prev = atomicCAS(kp, ~0ull, want);
    if (prev == ~0ull || prev == want) return &tab[slot];
    slot = (slot + 1) & mask;
  }
}

struct TableSet {
  HashEntry* tab[kMaxLevels];
  uint32_t mask[kMaxLevels];
};

__global__ __launch_bounds__(256) void k_cells_fill(const uint64_t* __restrict__ keys, int64_t n,
                                                    int bits, TableSet ts) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint64_t k = keys[i];
  if (i == 0) {
    for (int l = 0; l <= bits; ++l) {
      const uint64_t c = k >> (3 * l);
      table_slot(ts.tab[l], ts.mask[l], compact3(c), compact3(c >> 1), compact3(c >> 2))->start = 0;
    }
  } else {
    const uint64_t kp = keys[i - 1];
    int lv = boundary_levels(k, kp);
    if (lv > bits) lv = bits;
    for (int l = 0; l <= lv; ++l) {
      const uint64_t c = k >> (3 * l), cp = kp >> (3 * l);
      table_slot(ts.tab[l], ts.mask[l], compact3(c), compact3(c >> 1), compact3(c >> 2))->start = (uint32_t)i;
      table_slot(ts.tab[l], ts.mask[l], compact3(cp), compact3(cp >> 1), compact3(cp >> 2))->end = (uint32_t)i;
    }
  }
  if (i == n - 1) {
    for (int l = 0; l <= bits; ++l) {
      const uint64_t c = k >> (3 * l);
      table_slot(ts.tab[l], ts.mask[l], compact3(c), compact3(c >> 1), compact3(c >> 2))->end = (uint32_t)n;
    }
  }
}

// ---------------------------------------------------------------- kNN: main pass
// One lane per query (queries are Morton ordered, so a wave works on one neighbourhood).  Search the
// 2x2x2 block of level-`ls` cells nearest the query; the block guarantees every point within
// r_safe = (distance to the nearest block face) has been seen.  best <= r_safe^2 => exact, done.
// Otherwise the query goes to the straggler list with its upper bound.
constexpr float kCellSlack = 2e-3f;  // cells; covers float rounding of the cell assignment

__global__ __launch_bounds__(256) void k_knn_main(const float4* __restrict__ rdq, int nq, Mat34 T,
                                                  GridDev g, int ls,
                                                  const float4* __restrict__ pts,
                                                  int* __restrict__ ids, float* __restrict__ d2out,
                                                  uint32_t* __restrict__ strag,
                                                  uint32_t* __restrict__ strag_count) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= nq) return;
  const float4 r = rdq[j];
  const float3 q = xform(T, r.x, r.y, r.z);
  const float sc = g.inv_h0 * (1.0f / (float)(1 << ls));
  const float gx = (q.x - g.ox) * sc, gy = (q.y - g.oy) * sc, gz = (q.z - g.oz) * sc;
  const float bxf = floorf(gx - 0.5f), byf = floorf(gy - 0.5f), bzf = floorf(gz - 0.5f);
  const float fx = gx - bxf, fy = gy - byf, fz = gz - bzf;  // in [0.5, 1.5)
  float rs = fminf(fminf(fminf(fx, 2.f - fx), fminf(fy, 2.f - fy)), fminf(fz, 2.f - fz));
  rs = (rs - kCellSlack) * (g.h0 * (float)(1 << ls));
  const int dim = 1 << (g.bits - ls);
  // clamp far-away queries so the int conversion is defined; such cells do not exist anyway
  const int bx = (int)fminf(fmaxf(bxf, -2.f), (float)dim);
  const int by = (int)fminf(fmaxf(byf, -2.f), (float)dim);
  const int bz = (int)fminf(fmaxf(bzf, -2.f), (float)dim);
  float best = INFINITY;
  int bi = -1;
#pragma unroll 1
  for (int c = 0; c < 8; ++c) {
    const int cx = bx + (c & 1), cy = by + ((c >> 1) & 1), cz = bz + (c >> 2);
    if ((unsigned)cx >= (unsigned)dim || (unsigned)cy >= (unsigned)dim || (unsigned)cz >= (unsigned)dim)
      continue;
    uint32_t s, e;
    if (!grid_lookup(g, ls, (uint32_t)cx, (uint32_t)cy, (uint32_t)cz, s, e)) continue;
    for (uint32_t i = s; i < e; ++i) {
      const float4 p = pts[i];
      const float d = dist2(q.x - p.x, q.y - p.y, q.z - p.z);
      if (d < best) { best = d; bi = (int)i; }
    }
  }
  ids[j] = bi;
  d2out[j] = best;
  if (!(rs > 0.f && best <= rs * rs)) {
    const uint32_t slot = atomicAdd(strag_count, 1u);
    strag[slot] = (uint32_t)j;
  }
}

// ---------------------------------------------------------------- kNN: exact fallback
// One WAVE per straggler.  (A) get any upper bound B on the NN distance by climbing the pyramid;
// (B) pick the level whose edge >= B, so the cube [q-B, q+B] overlaps at most 3x3x3 cells, and scan
// those cells cooperatively.  The true NN lies inside that cube, hence exact.  Ties -> lowest index.
__global__ __launch_bounds__(256) void k_knn_fallback(const float4* __restrict__ rdq, Mat34 T,
                                                      GridDev g, const float4* __restrict__ pts,
                                                      int* __restrict__ ids,
                                                      float* __restrict__ d2out,
                                                      const uint32_t* __restrict__ strag,
                                                      const uint32_t* __restrict__ strag_count) {
  const int lane = threadIdx.x & 63;
  const uint32_t nw = gridDim.x * 4u;
  const uint32_t count = *strag_count;
  for (uint32_t s = blockIdx.x * 4u + (threadIdx.x >> 6); s < count; s += nw) {
    const uint32_t j = strag[s];
    const float4 r = rdq[j];
    const float3 q = xform(T, r.x, r.y, r.z);
    const float gx = (q.x - g.ox) * g.inv_h0, gy = (q.y - g.oy) * g.inv_h0,
                gz = (q.z - g.oz) * g.inv_h0;  // level-0 cell units
    float best = d2out[j];
    // ---- (A) upper bound
    if (!(best < INFINITY)) {
      for (int l = 0; l <= g.bits; ++l) {
        const float sc = 1.0f / (float)(1 << l);
        const int dim = 1 << (g.bits - l);
        const int cx = (int)fminf(fmaxf(floorf(gx * sc), 0.f), (float)(dim - 1));
        const int cy = (int)fminf(fmaxf(floorf(gy * sc), 0.f), (float)(dim - 1));
        const int cz = (int)fminf(fmaxf(floorf(gz * sc), 0.f), (float)(dim - 1));
        uint32_t cs, ce;
        if (grid_lookup(g, l, (uint32_t)cx, (uint32_t)cy, (uint32_t)cz, cs, ce)) {
          float d = INFINITY;
          if (cs + lane < ce) {
            const float4 p = pts[cs + lane];
            d = dist2(q.x - p.x, q.y - p.y, q.z - p.z);
          }
          best = wave_min(d);
          break;
        }
      }
    }
    // ---- (B) exact search in the cube of half-width B
    const float B = sqrtf(best) * (1.0f + 1e-5f);
    const float Bc = B * g.inv_h0 + kCellSlack;  // level-0 cell units
    int l1 = 0;
    while (l1 < g.bits && (float)(1 << l1) < Bc) ++l1;
    const float sc = 1.0f / (float)(1 << l1);
    const int dim = 1 << (g.bits - l1);
    const float fdim = (float)(dim - 1);
    const int x0 = (int)fminf(fmaxf(floorf((gx - Bc) * sc), 0.f), fdim);
    const int x1 = (int)fminf(fmaxf(floorf((gx + Bc) * sc), 0.f), fdim);
    const int y0 = (int)fminf(fmaxf(floorf((gy - Bc) * sc), 0.f), fdim);
    const int y1 = (int)fminf(fmaxf(floorf((gy + Bc) * sc), 0.f), fdim);
    const int z0 = (int)fminf(fmaxf(floorf((gz - Bc) * sc), 0.f), fdim);
    const int z1 = (int)fminf(fmaxf(floorf((gz + Bc) * sc), 0.f), fdim);
    const int nx = x1 - x0 + 1, ny = y1 - y0 + 1, nz = z1 - z0 + 1;  // each <= 3
    const int ncell = nx * ny * nz;
    uint32_t cs = 0, ce = 0;
    if (lane < ncell) {
      const int cx = x0 + lane % nx, cy = y0 + (lane / nx) % ny, cz = z0 + lane / (nx * ny);
      if (!grid_lookup(g, l1, (uint32_t)cx, (uint32_t)cy, (uint32_t)cz, cs, ce)) { cs = 0; ce = 0; }
    }
    unsigned long long bestp = ~0ull;
    for (int c = 0; c < ncell; ++c) {
      const uint32_t s0 = __shfl(cs, c, 64), e0 = __shfl(ce, c, 64);
      for (uint32_t i = s0 + lane; i < e0; i += 64) {
        const float4 p = pts[i];
        const float d = dist2(q.x - p.x, q.y - p.y, q.z - p.z);
        const unsigned long long pk = ((unsigned long long)__float_as_uint(d) << 32) | i;
        bestp = pk < bestp ? pk : bestp;
      }
    }
    bestp = wave_min_u64(bestp);
    if (lane == 0) {
      ids[j] = (int)(uint32_t)(bestp & 0xFFFFFFFFull);
      d2out[j] = __uint_as_float((uint32_t)(bestp >> 32));
    }
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

// ---------------------------------------------------------------- trimmed-distance order statistic
// Exact radix select on the float bit pattern of d2 (non-negative floats order like their bits):
// pass 1 bits [31:20], pass 2 bits [19:9], pass 3 bits [8:0].
struct SelState {
  uint32_t prefix;  // selected high bits so far
  uint32_t k;       // rank still to find inside the selected bin
};

// Whole block (256 threads): find bin b with cum(b) <= k < cum(b)+hist[b].
__device__ void find_bin(const uint32_t* __restrict__ hist, int nbins, uint32_t k, uint32_t* bin,
                         uint32_t* krem, uint32_t* sh /* >= 260 words */) {
  const int t = threadIdx.x;
  const int per = nbins / 256;  // nbins is a multiple of 256
  uint32_t loc = 0;
  for (int i = 0; i < per; ++i) loc += hist[t * per + i];
  sh[t] = loc;
  __syncthreads();
  if (t == 0) {
    uint32_t cum = 0;
    int sel = 255;
    for (int i = 0; i < 256; ++i) {
      if (k < cum + sh[i]) { sel = i; break; }
      cum += sh[i];
    }
    uint32_t b = sel * per;
    for (int i = 0; i < per; ++i) {
      const uint32_t c = hist[sel * per + i];
      b = sel * per + i;
      if (k < cum + c) break;
      if (i + 1 < per) cum += c;
    }
    sh[256] = b;
    sh[257] = k - cum;
  }
  __syncthreads();
  *bin = sh[256];
  *krem = sh[257];
  __syncthreads();
}

__global__ __launch_bounds__(256) void k_hist1(const float* __restrict__ d2, int n,
                                               uint32_t* __restrict__ hist) {
  __shared__ uint32_t sh[kHistBins];
  for (int i = threadIdx.x; i < kHistBins; i += 256) sh[i] = 0;
  __syncthreads();
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
    atomicAdd(&sh[__float_as_uint(d2[i]) >> 20], 1u);
  __syncthreads();
  for (int i = threadIdx.x; i < kHistBins; i += 256)
    if (sh[i]) atomicAdd(&hist[i], sh[i]);
}

// PASS 2: shift 9, 11 bits, parent = hist1 ; PASS 3: shift 0, 9 bits, parent = hist2
template <int PASS>
__global__ __launch_bounds__(256) void k_hist_refine(const float* __restrict__ d2, int n,
                                                     const uint32_t* __restrict__ parent,
                                                     const SelState* __restrict__ st_in,
                                                     SelState* __restrict__ st_out,
                                                     uint32_t* __restrict__ hist) {
  __shared__ uint32_t sh[kHistBins];
  __shared__ uint32_t sc[260];
  const SelState in = *st_in;
  uint32_t bin, krem;
  find_bin(parent, kHistBins, in.k, &bin, &krem, sc);
  const uint32_t prefix = (PASS == 2) ? bin : ((in.prefix << 11) | bin);
  if (blockIdx.x == 0 && threadIdx.x == 0) { st_out->prefix = prefix; st_out->k = krem; }
  for (int i = threadIdx.x; i < kHistBins; i += 256) sh[i] = 0;
  __syncthreads();
  constexpr int SH_HI = (PASS == 2) ? 20 : 9;
  constexpr int SH_LO = (PASS == 2) ? 9 : 0;
  constexpr uint32_t MASK = (PASS == 2) ? 0x7FFu : 0x1FFu;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const uint32_t b = __float_as_uint(d2[i]);
    if ((b >> SH_HI) == prefix) atomicAdd(&sh[(b >> SH_LO) & MASK], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kHistBins; i += 256)
    if (sh[i]) atomicAdd(&hist[i], sh[i]);
}

// limit bits = prefix(23 high bits) << 9 | bin
__device__ __forceinline__ float select_limit(const uint32_t* __restrict__ hist3,
                                              const SelState* __restrict__ st, uint32_t* sc) {
  const SelState in = *st;
  uint32_t bin, krem;
  find_bin(hist3, kHistBins, in.k, &bin, &krem, sc);
  return __uint_as_float((in.prefix << 9) | bin);
}

__global__ __launch_bounds__(256) void k_limit_out(const uint32_t* __restrict__ hist3,
                                                   const SelState* __restrict__ st,
                                                   float* __restrict__ out) {
  __shared__ uint32_t sc[260];
  const float lim = select_limit(hist3, st, sc);
  if (threadIdx.x == 0) *out = lim;
}

// ---------------------------------------------------------------- point-to-plane normal equations
// Per pair with weight 1: J = [p x n ; n] (float, as libpointmatcher), r = (p - q).n ;
// accumulate 21 upper-tri J J^T, 6 of -J r, count, r^2 in double.  Per-block partials, then a
// single-block fixed-order reduction => bitwise reproducible.
constexpr int kNe = 29;

template <bool IDS_ORIG, bool LIMIT_DEV>
__global__ __launch_bounds__(256) void k_normal_eq(const float4* __restrict__ rdq, int nq, Mat34 T,
                                                   const int* __restrict__ ids,
                                                   const float* __restrict__ d2,
                                                   const float4* __restrict__ pts,
                                                   const float4* __restrict__ nrm,
                                                   const uint32_t* __restrict__ inv,
                                                   const uint32_t* __restrict__ hist3,
                                                   const SelState* __restrict__ st, float limit_val,
                                                   float* __restrict__ limit_out,
                                                   double* __restrict__ partials) {
  __shared__ uint32_t sc[260];
  __shared__ double red[4][kNe];
  float limit = limit_val;
  if (LIMIT_DEV) {
    limit = select_limit(hist3, st, sc);
    if (blockIdx.x == 0 && threadIdx.x == 0) *limit_out = limit;
  }
  double acc[kNe];
#pragma unroll
  for (int k = 0; k < kNe; ++k) acc[k] = 0.0;
  for (int j = blockIdx.x * 256 + threadIdx.x; j < nq; j += gridDim.x * 256) {
    const float d = d2[j];
    int id = ids[j];
    if (!(d <= limit) || id < 0) continue;
    if (IDS_ORIG) id = (int)inv[id];
    const float4 r = rdq[j];
    const float3 p = xform(T, r.x, r.y, r.z);
    const float4 q = pts[id];
    const float4 n = nrm[id];
    float J[6];
    J[0] = p.y * n.z - p.z * n.y;
    J[1] = p.z * n.x - p.x * n.z;
    J[2] = p.x * n.y - p.y * n.x;
    J[3] = n.x; J[4] = n.y; J[5] = n.z;
    const float res = (p.x - q.x) * n.x + (p.y - q.y) * n.y + (p.z - q.z) * n.z;
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
      for (int c = a; c < 6; ++c) acc[k++] += (double)J[a] * (double)J[c];
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) acc[21 + a] -= (double)J[a] * (double)res;
    acc[27] += 1.0;
    acc[28] += (double)res * (double)res;
  }
#pragma unroll
  for (int k = 0; k < kNe; ++k) acc[k] = wave_sum(acc[k]);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < kNe; ++k) red[w][k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x < kNe)
    partials[(size_t)blockIdx.x * 32 + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

__global__ __launch_bounds__(64) void k_ne_final(const double* __restrict__ partials, int nblocks,
                                                 double* __restrict__ out) {
  const int t = threadIdx.x;
  if (t >= kNe) return;
  double s = 0.0;
  for (int b = 0; b < nblocks; ++b) s += partials[(size_t)b * 32 + t];
  out[t] = s;
}

}  // namespace lsgpu
