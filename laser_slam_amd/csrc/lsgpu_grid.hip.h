// lsgpu_grid.hip.h -- reference preparation: the KDTreeMatcher::init stand-in
// (laser_slam/configurations/icp_default.yaml:9-12; executed once per ICP::compute call,
// laser_slam/src/laser_track.cpp:496).
//
// Pipeline: k_ref_stats (mean, bbox) -> k_ref_keys (fine Morton key) -> radix sort ->
// k_ref_gather (sorted, centred points + normals) -> k_chunk_flags + scan -> k_chunk_bounds /
// k_chunk_boxes (<=64-point chunks with AABBs) -> k_cells_count / k_cells_fill (per-level hash
// tables cell -> chunk range).  All writes are coalesced except the hash inserts.
#pragma once
#include "lsgpu_common.hip.h"

namespace lsgpu {

// ---------------------------------------------------------------- reference statistics
struct RefStats {
  double sum[3];
  float mn[3];
  float mx[3];
  float zmn, zmx;   // range of zeta = z / |p| seen from the frame's own origin (rows of the direction index, lsgpu_cone.hip.h)
};

__global__ __launch_bounds__(256) void k_ref_stats(const float4* __restrict__ in, int64_t n,
                                                   RefStats* __restrict__ partials) {
  double sx = 0, sy = 0, sz = 0;
  float mnx = INFINITY, mny = INFINITY, mnz = INFINITY, mxx = -INFINITY, mxy = -INFINITY,
        mxz = -INFINITY, zmn = INFINITY, zmx = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float4 p = in[i];
    sx += p.x; sy += p.y; sz += p.z;
    mnx = fminf(mnx, p.x); mny = fminf(mny, p.y); mnz = fminf(mnz, p.z);
    mxx = fmaxf(mxx, p.x); mxy = fmaxf(mxy, p.y); mxz = fmaxf(mxz, p.z);
    const float r2 = p.x * p.x + p.y * p.y + p.z * p.z;
    if (r2 > 0.f) {   // (fminf / fmaxf drop a NaN operand: non-finite points are reported through the box)
      const float zeta = p.z * __builtin_amdgcn_rsqf(r2);
      zmn = fminf(zmn, zeta); zmx = fmaxf(zmx, zeta);
    }
  }
  sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
  mnx = wave_min(mnx); mny = wave_min(mny); mnz = wave_min(mnz);
  mxx = wave_max(mxx); mxy = wave_max(mxy); mxz = wave_max(mxz);
  zmn = wave_min(zmn); zmx = wave_max(zmx);
  __shared__ RefStats sh[4];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sh[w].sum[0] = sx; sh[w].sum[1] = sy; sh[w].sum[2] = sz;
    sh[w].mn[0] = mnx; sh[w].mn[1] = mny; sh[w].mn[2] = mnz;
    sh[w].mx[0] = mxx; sh[w].mx[1] = mxy; sh[w].mx[2] = mxz;
    sh[w].zmn = zmn; sh[w].zmx = zmx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    RefStats r = sh[0];
    for (int k = 1; k < 4; ++k) {
      for (int d = 0; d < 3; ++d) {
        r.sum[d] += sh[k].sum[d];
        r.mn[d] = fminf(r.mn[d], sh[k].mn[d]);
        r.mx[d] = fmaxf(r.mx[d], sh[k].mx[d]);
      }
      r.zmn = fminf(r.zmn, sh[k].zmn); r.zmx = fmaxf(r.zmx, sh[k].zmx);
    }
    partials[blockIdx.x] = r;
  }
}

// Grid geometry, derived ON THE DEVICE from the reference statistics so that the grid build does not have to wait
// for a host round trip between the statistics and the keys (the host reads it back later, together with the cell
// counts it needs anyway).
struct GeomDev {
  float mean[3];      // the float mean ICP::compute subtracts (step 2)
  float ox, oy, oz;   // grid origin = bounding-box minimum in the mean frame
  float h0, hf, inv_hf;
  int fine, bits;
  int bad;            // non-finite coordinates
  // direction index (lsgpu_cone.hip.h): range of zeta = z / |p| about the frame's origin, and whether that origin lies
  // inside the cloud's bounding box (a cloud seen from far outside itself has no use for an index by direction)
  float zeta_lo, zeta_hi;
  int origin_inside;
};

// One wave; fixed summation tree => deterministic mean.  nblocks <= 512.
__global__ __launch_bounds__(64) void k_ref_stats_final(const RefStats* __restrict__ partials,
                                                        int nblocks, RefStats* __restrict__ out, int64_t n,
                                                        float cell_size_cfg, GeomDev* __restrict__ geom) {
  const int lane = threadIdx.x;
  double s[3] = {0, 0, 0};
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  float zmn = INFINITY, zmx = -INFINITY;
  for (int k = lane; k < nblocks; k += 64) {
    for (int d = 0; d < 3; ++d) {
      s[d] += partials[k].sum[d];
      mn[d] = fminf(mn[d], partials[k].mn[d]);
      mx[d] = fmaxf(mx[d], partials[k].mx[d]);
    }
    zmn = fminf(zmn, partials[k].zmn); zmx = fmaxf(zmx, partials[k].zmx);
  }
  RefStats r;
  for (int d = 0; d < 3; ++d) {
    r.sum[d] = wave_sum(s[d]);
    r.mn[d] = wave_min(mn[d]);
    r.mx[d] = wave_max(mx[d]);
  }
  r.zmn = wave_min(zmn); r.zmx = wave_max(zmx);
  if (lane == 0) {
    *out = r;
    // same arithmetic the host used to do: float mean from the double sums, box in the mean frame, then 16 key bits per
    // axis split between level-0 cells (`bits`) and the order inside a cell (`fine`)
    GeomDev gm;
    float mn[3], mx[3];
    gm.bad = 0;
    for (int d = 0; d < 3; ++d) {
      gm.mean[d] = (float)(r.sum[d] / (double)n);
      mn[d] = r.mn[d] - gm.mean[d];
      mx[d] = r.mx[d] - gm.mean[d];
      if (!isfinite(mn[d]) || !isfinite(mx[d])) gm.bad = 1;
    }
    const float ext = fmaxf(fmaxf(mx[0] - mn[0], mx[1] - mn[1]), mx[2] - mn[2]);
    int bits = 11, fine = 5;
    float h0 = cell_size_cfg > 0.f ? cell_size_cfg : 0.125f;
    while (bits < 13 && h0 * (float)((1 << bits) - 1) < ext * 1.0001f) { ++bits; --fine; }
    while (!gm.bad && h0 * (float)((1 << bits) - 1) < ext * 1.0001f) h0 *= 2.f;
    gm.ox = mn[0]; gm.oy = mn[1]; gm.oz = mn[2];
    gm.h0 = h0; gm.hf = h0 / (float)(1 << fine); gm.inv_hf = 1.0f / gm.hf; gm.fine = fine; gm.bits = bits;
    gm.zeta_lo = r.zmn; gm.zeta_hi = r.zmx;
    gm.origin_inside = 1;
    for (int d = 0; d < 3; ++d) {   // the input frame's origin against the box, widened by a tenth of its extent
      const float m = 0.1f * (r.mx[d] - r.mn[d]) + 1e-3f;
      if (!(r.mn[d] - m <= 0.f && 0.f <= r.mx[d] + m)) gm.origin_inside = 0;
    }
    if (!(r.zmn <= r.zmx)) gm.origin_inside = 0;
    *geom = gm;
  }
}

// ---------------------------------------------------------------- keys
// Reference: centre on the mean, quantise at hf, Morton key of (fine + bits) bits per axis.
__global__ __launch_bounds__(256) void k_ref_keys(const float4* __restrict__ in, int64_t n,
                                                  const GeomDev* __restrict__ geom,
                                                  uint64_t* __restrict__ keys,
                                                  uint32_t* __restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const GeomDev g = *geom;
  const float mx = g.mean[0], my = g.mean[1], mz = g.mean[2];
  const float4 p = in[i];
  const int lim = (1 << (g.bits + g.fine)) - 1;
  const int ix = fine_coord(p.x - mx, g.ox, g.inv_hf, lim);
  const int iy = fine_coord(p.y - my, g.oy, g.inv_hf, lim);
  const int iz = fine_coord(p.z - mz, g.oz, g.inv_hf, lim);
  keys[i] = morton3((uint32_t)ix, (uint32_t)iy, (uint32_t)iz);
  vals[i] = (uint32_t)i;
}

// Reading: fine Morton key in its own frame (2^-7 m steps, 21 bits per axis), only so that the 64
// queries of a wave are neighbours.  Rigid motion keeps them neighbours in every iteration.
// order == 1 (default): spherical cells seen from the cloud's own origin -- elevation bin, azimuth sector (sized
// by k_query_order: 0.57 x 0.25 deg on the benchmark scan), range bin (1 m) -- and azimuth inside a cell.  A spinning lidar's scan in its sensor frame is
// a set of rings of constant elevation, and range noise moves a point along its ray, so this reproduces
// (ring, azimuth) order whatever order the caller stored the points in: a wave's 64 queries are one short arc
// instead of pieces of several rings inside a Morton block, and the reference points near them are about half
// as many (measured on the benchmark scan: median 96 against 213 points in the dilated tile box; 214 against
// 312 for a reading made of three merged scans, where the cells act as a plain spherical grid).  The order is
// free: results are returned in caller order.  order == 0 (LSGPU_QUERY_ORDER=0): Morton order at 2^-7 m.
constexpr int kAngElev = 1024, kAngSect = 4096;  // key fields: elevation bins, azimuth sectors (cell sizes adapt)

__device__ __forceinline__ void angular_cell(const float4& p, float inv_elev, float inv_sect, uint32_t& eb,
                                             uint32_t& sec, float& azim, float& range) {
  const float rho = sqrtf(p.x * p.x + p.y * p.y);
  const float elev = atan2f(p.z, rho);                                       // [-pi/2, pi/2]
  azim = atan2f(p.y, p.x) + 3.1415927f;                                       // [0, 2 pi]
  range = sqrtf(rho * rho + p.z * p.z);
  eb = (uint32_t)fminf(fmaxf((elev + 1.5707964f) * inv_elev, 0.f), (float)(kAngElev - 1));
  sec = (uint32_t)fminf(fmaxf(azim * inv_sect, 0.f), (float)(kAngSect - 1));
}

// How densely are the rings sampled?  Points per occupied 1 deg x 1 deg angular cell: cells[kDecCells] counts;
// cells[kDecCells + 0/1] = occupied cells / points, cells[kDecCells + 2] = the chosen order (k_query_order).
constexpr int kDecElev = 181, kDecSect = 361, kDecCells = 65536;  // 181 * 361 = 65341 cells, table of 2^16
__global__ __launch_bounds__(256) void k_query_ang_hist(const float4* __restrict__ in, int64_t n,
                                                        uint32_t* __restrict__ cells) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t cell = 0xFFFFFFFFu;
  if (i < n) {
    const float4 p = in[i];
    const float elev = atan2f(p.z, sqrtf(p.x * p.x + p.y * p.y)), azim = atan2f(p.y, p.x) + 3.1415927f;
    const uint32_t eb = (uint32_t)fminf(fmaxf((elev + 1.5707964f) * 57.29578f, 0.f), (float)(kDecElev - 1));
    const uint32_t sec = (uint32_t)fminf(fmaxf(azim * 57.29578f, 0.f), (float)(kDecSect - 1));
    // (scattered over the table: neighbouring cells, hit by consecutive waves, would share cache lines)
    cell = ((eb * kDecSect + sec) * 40503u) & 0xFFFFu;
  }
  // scans usually arrive ring by ring: neighbouring lanes fall into the same cell, and 64 atomics on one address
  // serialise.  One atomic per run of equal cells inside the wave instead.
  const int lane = threadIdx.x & 63;
  const uint32_t prev = (uint32_t)__shfl_up((int)cell, 1, 64);
  const bool head = lane == 0 || cell != prev;
  const unsigned long long heads = __ballot(head);
  if (head && cell != 0xFFFFFFFFu) {
    const unsigned long long later = lane == 63 ? 0ull : (heads >> (lane + 1));
    const int run = later ? __ffsll((long long)later) : 64 - lane;
    atomicAdd(&cells[cell], (uint32_t)run);
  }
}

// Cell size of the spherical order from the angular sampling density m = points per occupied square degree
// (108 on the 1 M-point benchmark scan, 54 on its random half, 21 on a 200 k-point scan): about 15 points per
// cell, elevation : azimuth = 0.8 : 0.35 (measured best on all three: a band of one to three rings, a sector of
// a dozen points per ring).  cells[kDecCells + 2] = order, + 3 / + 4 = 1 / elevation bin, 1 / sector (radians).
__global__ __launch_bounds__(1024) void k_query_order(uint32_t* __restrict__ cells, int forced, float elev_deg,
                                                      float sect_deg) {
  __shared__ uint32_t occ[16], tot[16];
  uint32_t o = 0, t = 0;
  for (int i = threadIdx.x; i < kDecCells; i += 1024) { const uint32_t c = cells[i]; o += c ? 1u : 0u; t += c; }
  o = wave_sum_u32(o); t = wave_sum_u32(t);
  if ((threadIdx.x & 63) == 0) { occ[threadIdx.x >> 6] = o; tot[threadIdx.x >> 6] = t; }
  __syncthreads();
  if (threadIdx.x == 0) {
    o = 0; t = 0;
    for (int w = 0; w < 16; ++w) { o += occ[w]; t += tot[w]; }
    uint32_t* out = cells + kDecCells;
    out[0] = o; out[1] = t;
    const float m = o ? (float)t / (float)o : 54.f;
    const float f = sqrtf(54.f / fmaxf(m, 1.f));
    const float e = elev_deg > 0.f ? elev_deg : fminf(fmaxf(0.8f * f, 0.2f), 4.f);
    const float sc = sect_deg > 0.f ? sect_deg : fminf(fmaxf(0.35f * f, 0.1f), 2.f);
    // a cloud seen under a small solid angle (its origin is far outside it, e.g. world coordinates) has no
    // ring structure to exploit from here: Morton order
    const uint32_t automatic = (o < 64u && t > 4096u) ? 0u : 1u;
    out[2] = forced >= 0 ? (uint32_t)forced : automatic;
    out[3] = __float_as_uint(57.29578f / e);
    out[4] = __float_as_uint(57.29578f / sc);
  }
}

__global__ __launch_bounds__(256) void k_query_keys(const float4* __restrict__ in, int64_t n,
                                                    uint64_t* __restrict__ keys,
                                                    uint32_t* __restrict__ vals, const uint32_t* __restrict__ order_flag) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 p = in[i];
  if (*order_flag == 1u) {
    uint32_t e32, s32; float azim, range;
    angular_cell(p, __uint_as_float(order_flag[1]), __uint_as_float(order_flag[2]), e32, s32, azim, range);
    const uint64_t eb = e32, sec = s32;
    // 48 key bits (6 radix passes): 10 elevation + 12 sector + 10 range (1 m bins) + 16 azimuth (2 pi -> 2^16, four
    // steps per firing of a 16384-column scanner); the sort is stable, coarser fields only leave more input order
    const uint64_t rb = (uint64_t)fminf(fmaxf(range, 0.f), 1023.f);
    const uint64_t fine = (uint64_t)fminf(fmaxf(azim * 10430.378f, 0.f), 65535.f);
    keys[i] = (((eb * 4096ull + sec) * 1024ull + rb) << 16) | fine;
    vals[i] = (uint32_t)i;
    return;
  }
  const float lim = 65535.f, half = 32768.f;   // 48-bit Morton key: 7.8 mm cells, +-256 m (beyond: clamped)
  const uint32_t ix = (uint32_t)fminf(fmaxf(floorf(p.x * 128.f) + half, 0.f), lim);
  const uint32_t iy = (uint32_t)fminf(fmaxf(floorf(p.y * 128.f) + half, 0.f), lim);
  const uint32_t iz = (uint32_t)fminf(fmaxf(floorf(p.z * 128.f) + half, 0.f), lim);
  keys[i] = morton3(ix, iy, iz);
  vals[i] = (uint32_t)i;
}

// sorted reference: pts[j] = {centred xyz, original index}, nrm[j] = {normal, 0}, inv[orig] = j
__global__ __launch_bounds__(256) void k_ref_gather(const float4* __restrict__ in,
                                                    const float* __restrict__ nrm_in, int64_t n,
                                                    const uint32_t* __restrict__ perm,
                                                    const GeomDev* __restrict__ geom, float4* __restrict__ pts,
                                                    float4* __restrict__ nrm,
                                                    uint32_t* __restrict__ inv) {
  const float mx = geom->mean[0], my = geom->mean[1], mz = geom->mean[2];
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) {  // 8 far pad points behind the array (4-wide point loads may run past the end)
    if (j < n + 8) pts[j] = make_float4(3e18f, 3e18f, 3e18f, 0.f);
    return;
  }
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

// RigidTransformation::compute on a 3-row descriptor (`normals`, `observationDirections`): d' = R d, rows of T without the
// translation, same fma chain as the points (laser_track.cpp:265, 485, 630, 643 rotate the descriptors of stored scans)
__global__ __launch_bounds__(256) void k_rotate3(const float* __restrict__ in, int64_t n, Mat34 T, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
  out[3 * i + 0] = __fmaf_rn(T.m[2], z, __fmaf_rn(T.m[1], y, T.m[0] * x));
  out[3 * i + 1] = __fmaf_rn(T.m[6], z, __fmaf_rn(T.m[5], y, T.m[4] * x));
  out[3 * i + 2] = __fmaf_rn(T.m[10], z, __fmaf_rn(T.m[9], y, T.m[8] * x));
}

// ---------------------------------------------------------------- chunks
// A chunk starts at every level-0 cell boundary and at every 64th sorted point.
__global__ __launch_bounds__(256) void k_chunk_flags(const uint64_t* __restrict__ keys, int64_t n,
                                                     const GeomDev* __restrict__ geom, uint32_t* __restrict__ flags) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int fine = geom->fine;
  uint32_t f = ((i & (kChunkMax - 1)) == 0);
  if (i > 0 && ((keys[i] ^ keys[i - 1]) >> (3 * fine)) != 0) f = 1;
  flags[i] = f;
}

// cidx = inclusive scan of flags.  bounds[c] = first point of chunk c; bounds[nchunks] = n.
__global__ __launch_bounds__(256) void k_chunk_bounds(const uint32_t* __restrict__ flags,
                                                      const uint32_t* __restrict__ cidx, int64_t n,
                                                      uint32_t* __restrict__ bounds) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (flags[i]) bounds[cidx[i] - 1] = (uint32_t)i;
  if (i == n - 1) bounds[cidx[i]] = (uint32_t)n;
}

// one wave per chunk: bounding box + descriptor
__global__ __launch_bounds__(256) void k_chunk_boxes(const float4* __restrict__ pts,
                                                     const uint32_t* __restrict__ bounds,
                                                     uint32_t nchunks,
                                                     ChunkDesc* __restrict__ chunks) {
  const uint32_t c = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (c >= nchunks) return;
  const int lane = threadIdx.x & 63;
  const uint32_t s = bounds[c], e = bounds[c + 1];
  float lx = INFINITY, ly = INFINITY, lz = INFINITY, hx = -INFINITY, hy = -INFINITY, hz = -INFINITY;
  if (s + lane < e) {
    const float4 p = pts[s + lane];
    lx = hx = p.x; ly = hy = p.y; lz = hz = p.z;
  }
  lx = wave_min(lx); ly = wave_min(ly); lz = wave_min(lz);
  hx = wave_max(hx); hy = wave_max(hy); hz = wave_max(hz);
  if (lane == 0) {
    ChunkDesc d;
    d.lox = lx; d.loy = ly; d.loz = lz; d.start = s;
    d.hix = hx; d.hiy = hy; d.hiz = hz; d.count = e - s;
    chunks[c] = d;
  }
}

// Chunk-blocked SoA copy of the sorted reference for the broadcast evaluation (lsgpu_knn.hip.h, tile_eval_slot):
// chunk c owns 3 * cnt4 floats -- x[cnt4] y[cnt4] z[cnt4], cnt4 = count rounded up to 4, the rounding slots far
// away -- so that one ds_read_b128 per coordinate hands four candidates to the packed-pair arithmetic.
// k_chunk_cnt4: slots per chunk (scanned on the host side of the stream into first slots); k_soa_fill: one wave
// per chunk, soa_base[c] = float4 index of the block.
// (also: the bounding box of every GROUP of kChunkGroup consecutive chunks -- a per-lane search that has to walk a cell
// of hundreds of chunks, a wall a metre from the sensor, skips sixteen chunk boxes per test: lane_ball_search)
__global__ __launch_bounds__(256) void k_chunk_cnt4(const uint32_t* __restrict__ bounds, uint32_t nchunks,
                                                    uint32_t* __restrict__ cnt4, const ChunkDesc* __restrict__ chunks,
                                                    ChunkDesc* __restrict__ groups) {
  const uint32_t c = blockIdx.x * 256u + threadIdx.x;
  if (c < nchunks && (c & (uint32_t)(kChunkGroup - 1)) == 0u) {
    ChunkDesc g;
    g.lox = g.loy = g.loz = INFINITY; g.hix = g.hiy = g.hiz = -INFINITY;
    g.start = c; g.count = 0u;
    for (uint32_t k = c; k < min(c + (uint32_t)kChunkGroup, nchunks); ++k) {
      const ChunkDesc d = chunks[k];
      g.lox = fminf(g.lox, d.lox); g.loy = fminf(g.loy, d.loy); g.loz = fminf(g.loz, d.loz);
      g.hix = fmaxf(g.hix, d.hix); g.hiy = fmaxf(g.hiy, d.hiy); g.hiz = fmaxf(g.hiz, d.hiz);
      g.count += 1u;
    }
    groups[c / (uint32_t)kChunkGroup] = g;
  }
  if (c < nchunks) cnt4[c] = (bounds[c + 1] - bounds[c] + 3u) & ~3u;
}

__global__ __launch_bounds__(256) void k_soa_fill(const float4* __restrict__ pts, const uint32_t* __restrict__ bounds,
                                                  const uint32_t* __restrict__ first_slot, uint32_t nchunks,
                                                  float* __restrict__ soa, uint32_t* __restrict__ soa_base) {
  const uint32_t c = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (c >= nchunks) return;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t s = bounds[c], cnt = bounds[c + 1] - s, c4 = (cnt + 3u) & ~3u, f0 = first_slot[c];
  if (lane == 0) soa_base[c] = (3u * f0) >> 2;  // (f0 is a multiple of 4)
  if (lane < c4) {
    float4 p = make_float4(3e18f, 3e18f, 3e18f, 0.f);
    if (lane < cnt) p = pts[s + lane];
    float* o = soa + 3u * (size_t)f0;
    o[lane] = p.x; o[c4 + lane] = p.y; o[2u * c4 + lane] = p.z;
  }
}

// ---------------------------------------------------------------- cell tables
// A level-l cell boundary sits between sorted points i-1 and i when their keys differ at or above
// bit 3*(fine+l).  Returns the highest such level, or -1.
__device__ __forceinline__ int boundary_level(uint64_t k, uint64_t kp, int fine) {
  const uint64_t x = k ^ kp;
  if (x == 0) return -1;
  return (63 - __clzll((long long)x)) / 3 - fine;
}

__global__ __launch_bounds__(256) void k_cells_count(const uint64_t* __restrict__ keys, int64_t n,
                                                     const GeomDev* __restrict__ geom,
                                                     uint32_t* __restrict__ counts) {
  __shared__ uint32_t sh[kMaxLevels];
  const int fine = geom->fine, bits = geom->bits;
  if (threadIdx.x < kMaxLevels) sh[threadIdx.x] = 0;
  __syncthreads();
  // grid-stride: few blocks, so that the final global adds (same 12 addresses for every block) stay cheap
  for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < n; i0 += (int64_t)gridDim.x * 256) {
    const int64_t i = i0 + threadIdx.x;
    int lv = -1;
    if (i < n) {
      lv = (i == 0) ? bits : boundary_level(keys[i], keys[i - 1], fine);
      if (lv > bits) lv = bits;
    }
    // per level: one ballot + one LDS add per wave (boundaries are rare: ~6 % of the points at level 0)
    for (int l = 0; l <= bits; ++l) {
      const unsigned long long m = __ballot(lv >= l);
      if (!m) break;
      if ((threadIdx.x & 63) == 0) atomicAdd(&sh[l], (uint32_t)__popcll(m));
    }
  }
  __syncthreads();
  if ((int)threadIdx.x <= bits && sh[threadIdx.x]) atomicAdd(&counts[threadIdx.x], sh[threadIdx.x]);
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

__device__ __forceinline__ HashEntry* cell_of_key(const TableSet& ts, int l, uint64_t key,
                                                  int fine) {
  const uint64_t c = key >> (3 * (fine + l));
  return table_slot(ts.tab[l], ts.mask[l], compact3(c), compact3(c >> 1), compact3(c >> 2));
}

// Entries hold CHUNK ranges.  Every cell boundary (any level) is also a chunk boundary, so one thread
// per CHUNK looks at the key step in front of its first point: chunk c opens the cells of the levels
// that change there and closes the previous point's cells.
__global__ __launch_bounds__(256) void k_cells_fill(const uint64_t* __restrict__ keys,
                                                    const uint32_t* __restrict__ bounds,
                                                    uint32_t nchunks, int fine, int bits, TableSet ts, int level0 = 0) {
  // one thread per (chunk, level): the inserts are device-scope CAS round trips, so the levels of one
  // chunk must not queue up behind each other in a single thread
  const uint32_t c = blockIdx.x * 256u + threadIdx.x;
  const int l = (int)blockIdx.y + level0;
  if (c >= nchunks) return;
  const uint32_t i = bounds[c];
  const uint64_t k = keys[i];
  if (c == 0) {
    cell_of_key(ts, l, k, fine)->start = 0;
  } else {
    const uint64_t kp = keys[i - 1];
    int lv = boundary_level(k, kp, fine);
    if (lv > bits) lv = bits;
    if (l <= lv) {
      cell_of_key(ts, l, k, fine)->start = c;
      cell_of_key(ts, l, kp, fine)->end = c;
    }
  }
  if (c == nchunks - 1) {
    const uint64_t kl = keys[bounds[nchunks] - 1];
    cell_of_key(ts, l, kl, fine)->end = nchunks;
  }
}

}  // namespace lsgpu
