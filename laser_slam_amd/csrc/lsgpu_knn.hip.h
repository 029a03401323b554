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
// Ties in distance: any nearest point is returned (libnabo's order is implementation defined).
#pragma once
#include "lsgpu_common.hip.h"

namespace lsgpu {

constexpr float kPruneShrink = 1.0f - 2e-6f;  // a box is skipped only if mind2 * this > best
constexpr float kPadCoord = 3e18f;            // LDS pad point: squared distance ~2.7e37, never best

struct KnnArgs {
  const float4* rdq;        // sorted reading (already moved by T_refMean_dataIn), w = caller index
  int nq;
  Mat34 T;                  // T_iter, applied on load (RigidTransformation, yaml default)
  GridDev g;
  const float4* pts;        // Morton-sorted centred reference
  const ChunkDesc* chunks;
  int* ids;                 // out: sorted-reference index of the NN
  float* d2;                // out: squared distance
  int* prev;                // in/out: warm start (sorted-reference index)
  uint32_t* strag;          // out: straggler list
  uint32_t* strag_count;
  float r_cap;              // lanes with a larger ball go to the fallback
  float group_r;            // half extent of one search group inside a wave
};

__device__ __forceinline__ float box_dist2(float lx, float ly, float lz, float hx, float hy,
                                           float hz, float qx, float qy, float qz) {
  const float dx = fmaxf(fmaxf(lx - qx, qx - hx), 0.f);
  const float dy = fmaxf(fmaxf(ly - qy, qy - hy), 0.f);
  const float dz = fmaxf(fmaxf(lz - qz, qz - hz), 0.f);
  return dx * dx + dy * dy + dz * dz;
}

// ---------------------------------------------------------------- seed
// Any reference point near the query: climb the pyramid from level 0 until the cell holding the
// query (clamped into the grid) exists, take the best of the first 8 points of its first chunk.
__global__ __launch_bounds__(256) void k_knn_seed(KnnArgs a) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= a.nq) return;
  const float4 r = a.rdq[j];
  const float3 q = xform(a.T, r.x, r.y, r.z);
  const GridDev& g = a.g;
  const int lim = (1 << (g.bits + g.fine)) - 1;
  const int fx = fine_coord(q.x, g.ox, g.inv_hf, lim);
  const int fy = fine_coord(q.y, g.oy, g.inv_hf, lim);
  const int fz = fine_coord(q.z, g.oz, g.inv_hf, lim);
  int bi = 0;
  for (int l = 0; l <= g.bits; ++l) {
    const int sh = g.fine + l;
    uint32_t cs, ce;
    if (!grid_lookup(g, l, (uint32_t)(fx >> sh), (uint32_t)(fy >> sh), (uint32_t)(fz >> sh), cs, ce))
      continue;
    const ChunkDesc d = a.chunks[cs];
    float best = INFINITY;
    const uint32_t n = d.count < 8u ? d.count : 8u;
    for (uint32_t t = 0; t < n; ++t) {
      const float4 p = a.pts[d.start + t];
      const float dd = dist2(q.x - p.x, q.y - p.y, q.z - p.z);
      if (dd < best) { best = dd; bi = (int)(d.start + t); }
    }
    break;
  }
  a.prev[j] = bi;
}

// ---------------------------------------------------------------- tile search
__global__ __launch_bounds__(256) void k_knn_tile(KnnArgs a) {
  __shared__ float4 cand[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const bool act = j < a.nq;
  const GridDev& g = a.g;

  float qx = 0.f, qy = 0.f, qz = 0.f, best = 0.f;
  int bi = -1;
  if (act) {
    const float4 r = a.rdq[j];
    const float3 q = xform(a.T, r.x, r.y, r.z);
    qx = q.x; qy = q.y; qz = q.z;
    bi = a.prev[j];
    const float4 p = a.pts[bi];
    best = dist2(qx - p.x, qy - p.y, qz - p.z);
  }
  const float R = sqrtf(best) * (1.0f + 1e-5f) + 1e-7f;
  const bool straggler = act && !(R <= a.r_cap);
  unsigned long long todo = __ballot(act && !straggler);
  const int lim = (1 << (g.bits + g.fine)) - 1;

  while (todo) {
    // ---- one group: lanes within group_r (Chebyshev) of the first unresolved lane
    const int piv = __ffsll((long long)todo) - 1;
    const float px = __shfl(qx, piv, 64), py = __shfl(qy, piv, 64), pz = __shfl(qz, piv, 64);
    const bool ing = ((todo >> lane) & 1ull) &&
                     fmaxf(fmaxf(fabsf(qx - px), fabsf(qy - py)), fabsf(qz - pz)) <= a.group_r;
    todo &= ~__ballot(ing);
    // query bbox and ball bbox of the group
    const float tlx = wave_min(ing ? qx : INFINITY), thx = wave_max(ing ? qx : -INFINITY);
    const float tly = wave_min(ing ? qy : INFINITY), thy = wave_max(ing ? qy : -INFINITY);
    const float tlz = wave_min(ing ? qz : INFINITY), thz = wave_max(ing ? qz : -INFINITY);
    const float rlx = wave_min(ing ? qx - R : INFINITY), rhx = wave_max(ing ? qx + R : -INFINITY);
    const float rly = wave_min(ing ? qy - R : INFINITY), rhy = wave_max(ing ? qy + R : -INFINITY);
    const float rlz = wave_min(ing ? qz - R : INFINITY), rhz = wave_max(ing ? qz + R : -INFINITY);
    float maxbest = wave_max(ing ? best : 0.f);
    // fine-key box of the region (uniform), widened by the rounding slack
    const int flx = __builtin_amdgcn_readfirstlane(fine_coord(rlx - kFineSlack * g.hf, g.ox, g.inv_hf, lim));
    const int fly = __builtin_amdgcn_readfirstlane(fine_coord(rly - kFineSlack * g.hf, g.oy, g.inv_hf, lim));
    const int flz = __builtin_amdgcn_readfirstlane(fine_coord(rlz - kFineSlack * g.hf, g.oz, g.inv_hf, lim));
    const int fhx = __builtin_amdgcn_readfirstlane(fine_coord(rhx + kFineSlack * g.hf, g.ox, g.inv_hf, lim));
    const int fhy = __builtin_amdgcn_readfirstlane(fine_coord(rhy + kFineSlack * g.hf, g.oy, g.inv_hf, lim));
    const int fhz = __builtin_amdgcn_readfirstlane(fine_coord(rhz + kFineSlack * g.hf, g.oz, g.inv_hf, lim));
    int l = 0, sh = g.fine;
    for (; l < g.bits; ++l, ++sh)
      if ((fhx >> sh) - (flx >> sh) < 4 && (fhy >> sh) - (fly >> sh) < 4 && (fhz >> sh) - (flz >> sh) < 4)
        break;
    sh = g.fine + l;
    const int x0 = flx >> sh, y0 = fly >> sh, z0 = flz >> sh;
    const int nx = (fhx >> sh) - x0 + 1, ny = (fhy >> sh) - y0 + 1, nz = (fhz >> sh) - z0 + 1;
    // ---- one cell per lane
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
      const uint32_t ccs = __shfl(cs, c, 64), cce = __shfl(ce, c, 64);
      for (uint32_t base = ccs; base < cce; base += 64) {
        // ---- lane-parallel cull of 64 chunk boxes against the group's query bbox
        const uint32_t ch = base + lane;
        float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
        bool pass = false;
        if (ch < cce) {
          const float4* cd = reinterpret_cast<const float4*>(a.chunks + ch);
          b0 = cd[0]; b1 = cd[1];
          const float gx = fmaxf(fmaxf(b0.x - thx, tlx - b1.x), 0.f);
          const float gy = fmaxf(fmaxf(b0.y - thy, tly - b1.y), 0.f);
          const float gz = fmaxf(fmaxf(b0.z - thz, tlz - b1.z), 0.f);
          pass = (gx * gx + gy * gy + gz * gz) * kPruneShrink <= maxbest;
        }
        unsigned long long m = __ballot(pass);
        while (m) {
          const int k = __ffsll((long long)m) - 1;
          m &= m - 1;
          const float lx = __shfl(b0.x, k, 64), ly = __shfl(b0.y, k, 64), lz = __shfl(b0.z, k, 64);
          const float hx = __shfl(b1.x, k, 64), hy = __shfl(b1.y, k, 64), hz = __shfl(b1.z, k, 64);
          // ---- per-lane test against the lane's own best
          const bool need = ing && box_dist2(lx, ly, lz, hx, hy, hz, qx, qy, qz) * kPruneShrink <= best;
          if (!__ballot(need)) continue;
          const uint32_t st = __builtin_amdgcn_readfirstlane(__float_as_uint(__shfl(b0.w, k, 64)));
          const uint32_t cnt = __builtin_amdgcn_readfirstlane(__float_as_uint(__shfl(b1.w, k, 64)));
          // ---- stage the chunk's points in this wave's LDS slot, broadcast to every lane
          float4 p = make_float4(kPadCoord, kPadCoord, kPadCoord, 0.f);
          if ((uint32_t)lane < cnt) p = a.pts[st + lane];
          cand[w][lane] = p;
          const uint32_t cnt4 = (cnt + 3u) & ~3u;
          for (uint32_t t = 0; t < cnt4; t += 4) {
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
              const float4 cpt = cand[w][t + u];
              const float d = dist2(qx - cpt.x, qy - cpt.y, qz - cpt.z);
              if (d < best) { best = d; bi = (int)(st + t + u); }
            }
          }
        }
        maxbest = wave_max(ing ? best : 0.f);
      }
    }
  }
  if (act) {
    a.ids[j] = bi;
    a.d2[j] = best;
    a.prev[j] = bi;
    if (straggler) a.strag[atomicAdd(a.strag_count, 1u)] = (uint32_t)j;
  }
}

// ---------------------------------------------------------------- exact fallback, one wave per query
__global__ __launch_bounds__(256) void k_knn_fallback(KnnArgs a) {
  const int lane = threadIdx.x & 63;
  const uint32_t nw = gridDim.x * 4u;
  const uint32_t count = *a.strag_count;
  const GridDev& g = a.g;
  const int lim = (1 << (g.bits + g.fine)) - 1;
  for (uint32_t s = blockIdx.x * 4u + (threadIdx.x >> 6); s < count; s += nw) {
    const uint32_t j = a.strag[s];
    const float4 r = a.rdq[j];
    const float3 q = xform(a.T, r.x, r.y, r.z);
    float best = a.d2[j];  // finite: distance to the warm-start point (possibly improved)
    unsigned long long bestp =
        ((unsigned long long)__float_as_uint(best) << 32) | (uint32_t)a.ids[j];
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
      const uint32_t ccs = __shfl(cs, c, 64), cce = __shfl(ce, c, 64);
      for (uint32_t base = ccs; base < cce; base += 64) {
        const uint32_t ch = base + lane;
        float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
        bool pass = false;
        if (ch < cce) {
          const float4* cd = reinterpret_cast<const float4*>(a.chunks + ch);
          b0 = cd[0]; b1 = cd[1];
          pass = box_dist2(b0.x, b0.y, b0.z, b1.x, b1.y, b1.z, q.x, q.y, q.z) * kPruneShrink <= best;
        }
        unsigned long long m = __ballot(pass);
        while (m) {
          const int k = __ffsll((long long)m) - 1;
          m &= m - 1;
          // the bound may have shrunk since the cull: re-test (uniform)
          const float lx = __shfl(b0.x, k, 64), ly = __shfl(b0.y, k, 64), lz = __shfl(b0.z, k, 64);
          const float hx = __shfl(b1.x, k, 64), hy = __shfl(b1.y, k, 64), hz = __shfl(b1.z, k, 64);
          if (!(box_dist2(lx, ly, lz, hx, hy, hz, q.x, q.y, q.z) * kPruneShrink <= best)) continue;
          const uint32_t st = __float_as_uint(__shfl(b0.w, k, 64));
          const uint32_t cnt = __float_as_uint(__shfl(b1.w, k, 64));
          float d = INFINITY;
          if ((uint32_t)lane < cnt) {
            const float4 p = a.pts[st + lane];
            d = dist2(q.x - p.x, q.y - p.y, q.z - p.z);
            const unsigned long long pk = ((unsigned long long)__float_as_uint(d) << 32) | (st + lane);
            bestp = pk < bestp ? pk : bestp;
          }
          best = fminf(best, wave_min(d));
        }
      }
    }
    bestp = wave_min_u64(bestp);
    if (lane == 0) {
      const int id = (int)(uint32_t)(bestp & 0xFFFFFFFFull);
      a.ids[j] = id;
      a.d2[j] = __uint_as_float((uint32_t)(bestp >> 32));
      a.prev[j] = id;
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

}  // namespace lsgpu
