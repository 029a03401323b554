// lsgpu_ssn.hip.h -- SamplingSurfaceNormalDataPointsFilter on the device:
// the reference filter of laser_slam/configurations/icp_default.yaml:5-7 (knn 10, ratio 0.5), which
// PointMatcher::ICP::compute applies to the reference cloud on the CPU at every call
// (laser_slam/src/laser_track.cpp:496).
//
// The filter splits the cloud recursively at the median of the widest box axis until a box holds
// <= knn points, gives every point of a box the box's PCA normal and keeps a random `ratio` of them.
// Every split is an exact halving (left = count - count / 2), so the tree's SHAPE depends only on n
// and knn: level L has at most 2^L segments with static sizes, and all levels are processed
// breadth-first, one stable radix sort per level with key = (segment, cut coordinate).  Stable =
// equal coordinates keep their order, the same rule the oracle and the host filter follow, so the
// three build identical boxes in identical order.  The per-box arithmetic is lsgpu_box_normal.h.
#pragma once
#include "lsgpu_common.hip.h"
#include "lsgpu_box_normal.h"
#include "lsgpu_segsort.hip.h"

namespace lsgpu {

struct SsnSeg {
  uint32_t start, count;
  float lo[3], hi[3];
};
constexpr int kSsnMaxKnn = 32;  // box points are staged in registers / scratch

// float -> uint32 with the same order; -0 and +0 share a key (they compare equal on the host too)
__device__ __forceinline__ uint32_t float_order_key(float f) {
  uint32_t u = __float_as_uint(f);
  if (u == 0x80000000u) u = 0u;
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float float_from_order_key(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

__device__ __forceinline__ float coord_of(const float4& p, int d) { return d == 0 ? p.x : (d == 1 ? p.y : p.z); }

__device__ __forceinline__ int ssn_cut_axis(const SsnSeg& s) {
  int cut = 0;
  float ext = s.hi[0] - s.lo[0];
  if (s.hi[1] - s.lo[1] > ext) { ext = s.hi[1] - s.lo[1]; cut = 1; }
  if (s.hi[2] - s.lo[2] > ext) { cut = 2; }
  return cut;
}

// bounding box of the cloud -> ordered-uint min/max (bb[0..2] = lo, bb[3..5] = hi; preset to ~0 / 0)
__global__ __launch_bounds__(256) void k_ssn_bounds(const float4* __restrict__ p, int n, uint32_t* __restrict__ bb) {
  uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float4 v = p[i];
    const uint32_t k[3] = {float_order_key(v.x), float_order_key(v.y), float_order_key(v.z)};
#pragma unroll
    for (int d = 0; d < 3; ++d) { lo[d] = min(lo[d], k[d]); hi[d] = max(hi[d], k[d]); }
  }
  __shared__ uint32_t red[4][6];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    for (int o = 32; o; o >>= 1) {
      lo[d] = min(lo[d], (uint32_t)__shfl_xor((int)lo[d], o));
      hi[d] = max(hi[d], (uint32_t)__shfl_xor((int)hi[d], o));
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][d] = lo[d]; red[threadIdx.x >> 6][3 + d] = hi[d]; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {  // one atomic per block and bound (same-address device atomics cost ~12 ns each)
    const int d = threadIdx.x;
    atomicMin(&bb[d], min(min(red[0][d], red[1][d]), min(red[2][d], red[3][d])));
    atomicMax(&bb[3 + d], max(max(red[0][3 + d], red[1][3 + d]), max(red[2][3 + d], red[3][3 + d])));
  }
}

// The cloud's bounds AND the root segment in ONE launch, without fills in front of it (round 5: the reference filter's chain
// is short enough for four 3 - 5 us launches -- two fills, the bounds, a one-thread root kernel -- to show): every block stores its six partial bounds, the block that draws the last
// ticket reduces them, writes bb[0..6) and the root segment and puts the ticket back to zero for the next call.
// ws: [0] the ticket (zero when the buffer is made), [8 + 6 b + d] block b's partial (kSsnBoundsBlocks of them at most).
constexpr int kSsnBoundsBlocks = 256;
__global__ __launch_bounds__(256) void k_ssn_bounds_root(const float4* __restrict__ p, int n, uint32_t* __restrict__ ws,
                                                         uint32_t* __restrict__ bb, SsnSeg* __restrict__ seg) {
  uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
  const int stride = gridDim.x * 256;
  int i = blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {     // four loads in flight per thread (256 blocks: a ticket costs ~50 ns)
    const float4 v[4] = {p[i], p[i + stride], p[i + 2 * stride], p[i + 3 * stride]};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t k[3] = {float_order_key(v[u].x), float_order_key(v[u].y), float_order_key(v[u].z)};
#pragma unroll
      for (int d = 0; d < 3; ++d) { lo[d] = min(lo[d], k[d]); hi[d] = max(hi[d], k[d]); }
    }
  }
  for (; i < n; i += stride) {
    const float4 v = p[i];
    const uint32_t k[3] = {float_order_key(v.x), float_order_key(v.y), float_order_key(v.z)};
#pragma unroll
    for (int d = 0; d < 3; ++d) { lo[d] = min(lo[d], k[d]); hi[d] = max(hi[d], k[d]); }
  }
  __shared__ uint32_t red[4][6];
  __shared__ uint32_t last;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    for (int o = 32; o; o >>= 1) {
      lo[d] = min(lo[d], (uint32_t)__shfl_xor((int)lo[d], o));
      hi[d] = max(hi[d], (uint32_t)__shfl_xor((int)hi[d], o));
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][d] = lo[d]; red[threadIdx.x >> 6][3 + d] = hi[d]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int d = threadIdx.x;
    const uint32_t v = d < 3 ? min(min(red[0][d], red[1][d]), min(red[2][d], red[3][d]))
                             : max(max(red[0][d], red[1][d]), max(red[2][d], red[3][d]));
    // (an exchange, and its old value waited for: the partial has reached the memory side when the ticket is drawn.
    //  __threadfence() would do, at the price of writing this XCD's whole L2 back and invalidating it -- buffer_wbl2 sc1 +
    //  buffer_inv sc1 on gfx950, measured ~10 us in this kernel)
    const uint32_t old = __hip_atomic_exchange(&ws[8 + 6 * blockIdx.x + d], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::"v"(old));
  }
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(&ws[0], 1u) == gridDim.x - 1u ? 1u : 0u;
  __syncthreads();
  if (!last) return;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    lo[d] = 0xFFFFFFFFu; hi[d] = 0u;
    for (uint32_t b = threadIdx.x; b < gridDim.x; b += 256u) {
      lo[d] = min(lo[d], __hip_atomic_load(&ws[8 + 6 * b + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      hi[d] = max(hi[d], __hip_atomic_load(&ws[8 + 6 * b + 3 + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    for (int o = 32; o; o >>= 1) {
      lo[d] = min(lo[d], (uint32_t)__shfl_xor((int)lo[d], o));
      hi[d] = max(hi[d], (uint32_t)__shfl_xor((int)hi[d], o));
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][d] = lo[d]; red[threadIdx.x >> 6][3 + d] = hi[d]; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    SsnSeg s;
    s.start = 0; s.count = (uint32_t)n;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const uint32_t l = min(min(red[0][d], red[1][d]), min(red[2][d], red[3][d]));
      const uint32_t u = max(max(red[0][3 + d], red[1][3 + d]), max(red[2][3 + d], red[3][3 + d]));
      bb[d] = l; bb[3 + d] = u;
      s.lo[d] = float_from_order_key(l); s.hi[d] = float_from_order_key(u);
    }
    seg[0] = s;
    __hip_atomic_store(&ws[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// level keys: (segment << 32) | ordered cut coordinate for segments that still split, (segment << 32) |
// rank for finished ones (they keep their order).  idx_in == nullptr: identity (level 0).
__global__ __launch_bounds__(256) void k_ssn_keys(const float4* __restrict__ p, int n,
                                                  const uint32_t* __restrict__ idx_in,
                                                  const uint32_t* __restrict__ seg_of,
                                                  const SsnSeg* __restrict__ segs, int knn,
                                                  uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int pos = blockIdx.x * 256 + threadIdx.x;
  if (pos >= n) return;
  const uint32_t s = seg_of ? seg_of[pos] : 0u;
  const SsnSeg sg = segs[s];
  const uint32_t i = idx_in ? idx_in[pos] : (uint32_t)pos;
  uint32_t low;
  if (sg.count > (uint32_t)knn) low = float_order_key(coord_of(p[i], ssn_cut_axis(sg)));
  else low = (uint32_t)pos - sg.start;
  keys[pos] = ((uint64_t)s << 32) | low;
  vals[pos] = i;
}

// children of every segment of this level (2s, 2s+1); a finished segment carries over as child 2s.
// axis_in / axis_out (nullable): the axis whose stable order a segment's points are in -- a child is a contiguous half of
// its parent's sorted order, so it inherits the parent's cut axis (lsgpu_segsort.hip.h: a child that cuts along the same
// axis again needs no sort).
__global__ __launch_bounds__(256) void k_ssn_split(const float4* __restrict__ p, const uint32_t* __restrict__ idx,
                                                   const SsnSeg* __restrict__ segs, int nseg, int knn,
                                                   SsnSeg* __restrict__ out, const int* __restrict__ axis_in = nullptr,
                                                   int* __restrict__ axis_out = nullptr) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= nseg) return;
  const SsnSeg sg = segs[s];
  SsnSeg a = sg, b = sg;
  int ax = axis_in ? axis_in[s] : -1;
  if (sg.count > (uint32_t)knn) {
    const int cut = ssn_cut_axis(sg);
    const uint32_t right = sg.count / 2, left = sg.count - right;
    const float cutval = coord_of(p[idx[sg.start + left]], cut);
    a.count = left; a.hi[cut] = cutval;
    b.start = sg.start + left; b.count = right; b.lo[cut] = cutval;
    ax = cut;
  } else {
    b.start = sg.start + sg.count; b.count = 0;
  }
  out[2 * s] = a;
  out[2 * s + 1] = b;
  if (axis_out) { axis_out[2 * s] = ax; axis_out[2 * s + 1] = ax; }
}

// ---- segmented sorts of a level (lsgpu_segsort.hip.h)
constexpr int kSegItems = 8;
constexpr uint32_t kSegTile = 256u * kSegItems;   // elements per block
constexpr uint32_t kSegNone = 0xFFFFFFFFu;

// ONE block: which segments of the level have to be sorted (they still split, and along another axis than the one their
// order already follows), their blocks, the block table.  seg_fb[s] = first block of segment s or kSegNone.
__global__ __launch_bounds__(256) void k_ssn_plan(const SsnSeg* __restrict__ segs, int ns, int knn,
                                                  const int* __restrict__ axis, uint32_t* __restrict__ seg_fb,
                                                  SegBlock* __restrict__ tab, uint32_t* __restrict__ nblocks_dev) {
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t big_list[256];
  __shared__ uint32_t big_n;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t carry = 0u;
  for (int s0 = 0; s0 < ns; s0 += 256) {
    const int s = s0 + (int)threadIdx.x;
    SsnSeg sg;
    sg.start = 0; sg.count = 0;
    uint32_t nb = 0u;
    if (s < ns) {
      sg = segs[s];
      const bool need = sg.count > (uint32_t)knn && ssn_cut_axis(sg) != axis[s];
      nb = need ? (sg.count + kSegTile - 1u) / kSegTile : 0u;
    }
    const uint32_t incl = wave_scan_incl_u32(nb, lane);
    if (lane == 63) wsum[w] = incl;
    if (threadIdx.x == 0) big_n = 0u;
    __syncthreads();
    uint32_t before = carry;
    for (int ww = 0; ww < w; ++ww) before += wsum[ww];
    const uint32_t fb = before + incl - nb;
    if (s < ns) seg_fb[s] = nb ? fb : kSegNone;
    // the blocks of this segment: a few -> this thread; many (the top levels) -> the whole block, below
    if (nb > 8u) {
      big_list[atomicAdd(&big_n, 1u)] = (uint32_t)threadIdx.x;
    } else {
      for (uint32_t k = 0; k < nb; ++k) {
        SegBlock e;
        e.first = sg.start + k * kSegTile; e.count = min(kSegTile, sg.count - k * kSegTile);
        e.seg_start = sg.start; e.fb = fb; e.nb = nb; e.pad[0] = e.pad[1] = e.pad[2] = 0u;
        tab[fb + k] = e;
      }
    }
    carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    const uint32_t nbig = big_n;
    for (uint32_t q = 0; q < nbig; ++q) {
      const int sb = s0 + (int)big_list[q];
      const SsnSeg g = segs[sb];
      const uint32_t gnb = (g.count + kSegTile - 1u) / kSegTile, gfb = seg_fb[sb];
      for (uint32_t k = threadIdx.x; k < gnb; k += 256u) {
        SegBlock e;
        e.first = g.start + k * kSegTile; e.count = min(kSegTile, g.count - k * kSegTile);
        e.seg_start = g.start; e.fb = gfb; e.nb = gnb; e.pad[0] = e.pad[1] = e.pad[2] = 0u;
        tab[gfb + k] = e;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *nblocks_dev = carry;
}

// 32-bit level keys: the ordered cut coordinate of every point whose segment is sorted at this level; the other points
// are not touched (their segments keep their order).  idx == nullptr: level 0, identity order (written to vals).
__global__ __launch_bounds__(256) void k_ssn_keys32(const float4* __restrict__ p, int n, uint32_t* __restrict__ idx,
                                                    const uint32_t* __restrict__ seg_of, const SsnSeg* __restrict__ segs,
                                                    const uint32_t* __restrict__ seg_fb, int first_level,
                                                    uint32_t* __restrict__ keys) {
  const int pos = blockIdx.x * 256 + threadIdx.x;
  if (pos >= n) return;
  const uint32_t s = first_level ? 0u : seg_of[pos];
  uint32_t i = (uint32_t)pos;
  if (first_level) idx[pos] = i; else i = idx[pos];
  if (seg_fb[s] == kSegNone) return;
  keys[pos] = float_order_key(coord_of(p[i], ssn_cut_axis(segs[s])));
}

__global__ __launch_bounds__(256) void k_ssn_assign(int n, const SsnSeg* __restrict__ parents, int knn,
                                                    uint32_t* __restrict__ seg_of, int first_level) {
  const int pos = blockIdx.x * 256 + threadIdx.x;
  if (pos >= n) return;
  const uint32_t s = first_level ? 0u : seg_of[pos];
  const SsnSeg sg = parents[s];
  uint32_t c = 2u * s;
  if (sg.count > (uint32_t)knn) {
    const uint32_t left = sg.count - sg.count / 2;
    c += ((uint32_t)pos - sg.start) >= left ? 1u : 0u;
  }
  seg_of[pos] = c;
}

// ---- the last levels inside one workgroup.  Once a segment holds <= kSsnLdsMax points the remaining
// levels need no global sort: one block per segment keeps the coordinates in LDS and, per level, sorts
// 64-bit keys (local segment | ordered cut coordinate | current position) with a bitonic network -- the
// position in the lowest bits makes the order stable, i.e. identical to the global stable radix sort --
// then derives the children exactly like k_ssn_split / k_ssn_assign.
constexpr int kSsnLdsMax = 2048;
constexpr int kSsnLdsSegs = 256;  // local segments at the last in-block level (>= kSsnLdsMax / (knn / 2))

constexpr int kSsnLdsLevels = 8;  // log2(kSsnLdsSegs)
struct SsnLds {                   // 68 KB: two workgroups per CU
  float c[3][kSsnLdsMax];
  uint32_t key[2][kSsnLdsMax];    // ping-pong: ordered cut coordinate of the element ...
  uint16_t from[2][kSsnLdsMax];   // ... and the position it had when the level started
  uint32_t cnt[4][256];           // radix pass: per wave and digit
  uint32_t wtot[4];
  uint16_t perm[kSsnLdsMax];
  uint16_t sof[kSsnLdsMax];
  SsnSeg seg[kSsnLdsSegs];
};

// One stable radix pass (8 bits) over the block's `cnt` elements in LDS, src -> dst.  digit_of(i) is evaluated for the
// element at position i of `src`.  Ranks as in k_rs_scatter (lsgpu_sort.hip.h): wave by wave, 64 consecutive elements
// at a time, equal digits found with 8 ballots, the lowest lane advances the wave's counter.
template <class DigitOf>
__device__ __forceinline__ void ssn_lds_radix_pass(SsnLds& L, int src, int cnt, DigitOf digit_of) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int dst = src ^ 1;
#pragma unroll
  for (int i = 0; i < 4; ++i) L.cnt[i][tid] = 0u;
  __syncthreads();
  constexpr int kIt = kSsnLdsMax / 256;   // 8 groups of 64 per wave
  uint32_t rank[kIt], dig[kIt];
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int it = 0; it < kIt; ++it) {
    const int i = w * (kSsnLdsMax / 4) + it * 64 + lane;
    const bool valid = i < cnt;
    const uint32_t d = valid ? digit_of(i) : 0u;
    dig[it] = d;
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    const int leader = valid ? __ffsll((long long)peers) - 1 : lane;
    uint32_t old = 0u;
    if (valid && lane == leader) {
      old = L.cnt[w][d];
      L.cnt[w][d] = old + (uint32_t)__popcll(peers);
    }
    old = (uint32_t)__shfl((int)old, leader, 64);
    rank[it] = old + (uint32_t)__popcll(peers & lt);
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  }
  __syncthreads();
  {  // thread d: digit base (exclusive scan over the digits) + the waves before
    uint32_t c[4], tot = 0u;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) { c[ww] = L.cnt[ww][tid]; tot += c[ww]; }
    uint32_t incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (lane == 63) L.wtot[w] = incl;
    __syncthreads();
    uint32_t run = incl - tot;
    for (int ww = 0; ww < w; ++ww) run += L.wtot[ww];
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) { L.cnt[ww][tid] = run; run += c[ww]; }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kIt; ++it) {
    const int i = w * (kSsnLdsMax / 4) + it * 64 + lane;
    if (i < cnt) {
      const uint32_t pos = L.cnt[w][dig[it]] + rank[it];
      L.key[dst][pos] = L.key[src][i];
      L.from[dst][pos] = L.from[src][i];
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void k_ssn_finish(const float4* __restrict__ p, uint32_t* __restrict__ idx,
                                                    const SsnSeg* __restrict__ segs, int knn, int rem,
                                                    uint32_t* __restrict__ seg_of, SsnSeg* __restrict__ segs_out) {
  __shared__ SsnLds L;
  const int tid = threadIdx.x;
  const SsnSeg root = segs[blockIdx.x];
  const int cnt = (int)root.count;
  for (int i = tid; i < cnt; i += 256) {
    const uint32_t gi = idx[root.start + i];
    const float4 v = p[gi];
    L.c[0][i] = v.x; L.c[1][i] = v.y; L.c[2][i] = v.z;
    L.perm[i] = (uint16_t)i;
    L.sof[i] = 0;
  }
  if (tid == 0) { SsnSeg r = root; r.start = 0; L.seg[0] = r; }
  __syncthreads();
  for (int l = 0; l < rem; ++l) {
    const int ns = 1 << l;
    // Stable sort by (segment, ordered cut coordinate).  The positions are grouped by segment already, in segment
    // order, so: four stable radix passes over the coordinate's bytes, then one over the segment number brings the
    // groups back together with each group sorted -- the same order as one stable sort of the combined key (what the
    // global levels do), with 15 barriers instead of the 66 of a bitonic network over 2048 keys.
    for (int i = tid; i < cnt; i += 256) {
      const uint32_t sgi = L.sof[i];
      const SsnSeg& sg = L.seg[sgi];
      uint32_t low = 0;
      if (sg.count > (uint32_t)knn) low = float_order_key(L.c[ssn_cut_axis(sg)][L.perm[i]]);
      L.key[0][i] = low;
      L.from[0][i] = (uint16_t)i;
    }
    __syncthreads();
    int cur = 0;
    for (int shift = 0; shift < 32; shift += 8) {
      ssn_lds_radix_pass(L, cur, cnt, [&](int i) { return (L.key[cur][i] >> shift) & 255u; });
      cur ^= 1;
    }
    if (ns > 1) {
      ssn_lds_radix_pass(L, cur, cnt, [&](int i) { return (uint32_t)L.sof[L.from[cur][i]]; });
      cur ^= 1;
    }
    // apply the permutation (read everything, then write)
    uint16_t np[kSsnLdsMax / 256];
#pragma unroll
    for (int r = 0; r < kSsnLdsMax / 256; ++r) {
      const int i = tid + r * 256;
      np[r] = i < cnt ? L.perm[L.from[cur][i]] : (uint16_t)0;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kSsnLdsMax / 256; ++r) {
      const int i = tid + r * 256;
      if (i < cnt) L.perm[i] = np[r];
    }
    __syncthreads();
    // children (computed from the parents, written once nobody reads the parents any more)
    SsnSeg a, b;
    if (tid < ns) {
      const SsnSeg sg = L.seg[tid];
      a = sg; b = sg;
      if (sg.count > (uint32_t)knn) {
        const int cut = ssn_cut_axis(sg);
        const uint32_t right = sg.count / 2, left = sg.count - right;
        const float cutval = L.c[cut][L.perm[sg.start + left]];
        a.count = left; a.hi[cut] = cutval;
        b.start = sg.start + left; b.count = right; b.lo[cut] = cutval;
      } else {
        b.start = sg.start + sg.count; b.count = 0;
      }
    }
    for (int i = tid; i < cnt; i += 256) {
      const uint32_t s = L.sof[i];
      const SsnSeg& sg = L.seg[s];
      uint32_t c = 2u * s;
      if (sg.count > (uint32_t)knn) {
        const uint32_t left = sg.count - sg.count / 2;
        c += ((uint32_t)i - sg.start) >= left ? 1u : 0u;
      }
      L.sof[i] = (uint16_t)c;
    }
    __syncthreads();
    if (tid < ns) { L.seg[2 * tid] = a; L.seg[2 * tid + 1] = b; }
    __syncthreads();
  }
  const uint32_t base_seg = (uint32_t)blockIdx.x << rem;
  uint32_t gi[kSsnLdsMax / 256];  // in place: read the block's whole index range before writing it
#pragma unroll
  for (int r = 0; r < kSsnLdsMax / 256; ++r) {
    const int i = tid + r * 256;
    gi[r] = i < cnt ? idx[root.start + L.perm[i]] : 0u;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kSsnLdsMax / 256; ++r) {
    const int i = tid + r * 256;
    if (i < cnt) {
      idx[root.start + i] = gi[r];
      seg_of[root.start + i] = base_seg + L.sof[i];
    }
  }
  for (int s = tid; s < (1 << rem); s += 256) {
    SsnSeg sg = L.seg[s];
    sg.start += root.start;
    segs_out[base_seg + s] = sg;
  }
}

// one thread per box: normal (or "dropped"); box_pts = number of points that draw a random number
__global__ __launch_bounds__(128) void k_ssn_boxes(const float4* __restrict__ p, const uint32_t* __restrict__ idx,
                                                   const SsnSeg* __restrict__ segs, int nseg,
                                                   float* __restrict__ box_normal, uint32_t* __restrict__ box_pts) {
  const int s = blockIdx.x * 128 + threadIdx.x;
  if (s >= nseg) return;
  const SsnSeg sg = segs[s];
  uint32_t kept = 0;
  if (sg.count > 0) {
    float n[3];
    const uint32_t* bi = idx + sg.start;
    if (boxnormal::box_normal((int)sg.count, [&](int i, int d) { return coord_of(p[bi[i]], d); }, n)) {
      kept = sg.count;
      box_normal[3 * (size_t)s + 0] = n[0];
      box_normal[3 * (size_t)s + 1] = n[1];
      box_normal[3 * (size_t)s + 2] = n[2];
    }
  }
  box_pts[s] = kept;
}

// per position (box-traversal order): does the point survive?  r = the rand() draw of this point = draws[box_base + rank]
// (dropped boxes draw nothing, exactly like the sequential filter).  The verdict goes to the point's ORIGINAL index, as ONE
// word -- 0: dropped, box + 1: kept, with the normal of that box -- because upstream sorts indicesToKeep before it compacts the
// cloud in place: the filtered cloud is in original order (round 6; rounds 1-5 emitted in traversal order).  The scan
// behind it counts the non-zero words (scan_u32's `nonzero`).
__global__ __launch_bounds__(256) void k_ssn_select(int n, const uint32_t* __restrict__ idx,
                                                    const uint32_t* __restrict__ seg_of,
                                                    const SsnSeg* __restrict__ segs,
                                                    const uint32_t* __restrict__ box_pts,
                                                    const uint32_t* __restrict__ box_base,
                                                    const float* __restrict__ draws, float ratio,
                                                    uint32_t* __restrict__ keep) {
  const int pos = blockIdx.x * 256 + threadIdx.x;
  if (pos >= n) return;
  const uint32_t s = seg_of[pos];
  uint32_t k = 0;
  if (box_pts[s]) {
    const float r = draws[box_base[s] + ((uint32_t)pos - segs[s].start)];
    k = r < ratio ? 1u : 0u;
  }
  keep[idx[pos]] = k ? s + 1u : 0u;
}

// the compaction by original index: point i, if kept, with the normal of its box
__global__ __launch_bounds__(256) void k_ssn_emit(const float4* __restrict__ p, int n,
                                                  const float* __restrict__ box_normal,
                                                  const uint32_t* __restrict__ keep,
                                                  const uint32_t* __restrict__ out_pos,
                                                  float4* __restrict__ out_xyz1, float* __restrict__ out_nrm) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n || !keep[i]) return;
  const uint32_t o = out_pos[i], s = keep[i] - 1u;
  out_xyz1[o] = p[i];
  out_nrm[3 * (size_t)o + 0] = box_normal[3 * (size_t)s + 0];
  out_nrm[3 * (size_t)o + 1] = box_normal[3 * (size_t)s + 1];
  out_nrm[3 * (size_t)o + 2] = box_normal[3 * (size_t)s + 2];
}

// RandomSamplingDataPointsFilter (yaml:1-3): keep point i iff its draw < prob
__global__ __launch_bounds__(256) void k_draw_select(int n, const float* __restrict__ draws, float prob,
                                                     uint32_t* __restrict__ keep) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) keep[i] = draws[i] < prob ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_compact_points(const float4* __restrict__ p, int n,
                                                        const uint32_t* __restrict__ keep,
                                                        const uint32_t* __restrict__ out_pos,
                                                        float4* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n && keep[i]) out[out_pos[i]] = p[i];
}

// ---- local-map maintenance
// applyCylindricalFilter (laser_slam_ros/include/laser_slam_ros/common.hpp:194-223)
__global__ __launch_bounds__(256) void k_cylinder_select(const float4* __restrict__ p, int n, float cx, float cy,
                                                         float cz, double radius_squared, double height_halved,
                                                         int remove_point_inside, uint32_t* __restrict__ keep) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 v = p[i];
  const float dx = v.x - cx, dy = v.y - cy, dz = fabsf(v.z - cz);
  const double r2 = (double)dx * (double)dx + (double)dy * (double)dy;
  const bool inside = r2 <= radius_squared && (double)dz <= height_halved;
  const bool outside = r2 >= radius_squared || (double)dz >= height_halved;
  keep[i] = (remove_point_inside ? outside : inside) ? 1u : 0u;
}

// pcl::VoxelGrid: voxel index of every point (key), original index (value)
__global__ __launch_bounds__(256) void k_voxel_keys(const float4* __restrict__ p, int n, float ix, float iy, float iz,
                                                    int bx, int by, int bz, int mul1, int mul2,
                                                    uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 v = p[i];
  const int i0 = (int)(floorf(v.x * ix) - (float)bx);
  const int i1 = (int)(floorf(v.y * iy) - (float)by);
  const int i2 = (int)(floorf(v.z * iz) - (float)bz);
  keys[i] = (uint64_t)(uint32_t)(i0 + i1 * mul1 + i2 * mul2);
  vals[i] = (uint32_t)i;
}

// head flag of every voxel run in the sorted keys; flags[i] = 1 where a new voxel starts
__global__ __launch_bounds__(256) void k_voxel_heads(const uint64_t* __restrict__ keys, int n, uint32_t* __restrict__ flags) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

// one thread per sorted position that starts a voxel: centroid of the run (float sums in input order, as the
// stable sort left them), kept[i] = 1 if the voxel holds >= min_points points
__global__ __launch_bounds__(256) void k_voxel_centroids(const float4* __restrict__ p, const uint64_t* __restrict__ keys,
                                                         const uint32_t* __restrict__ idx, int n, int min_points,
                                                         const uint32_t* __restrict__ flags, float4* __restrict__ cent,
                                                         uint32_t* __restrict__ kept) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t k = 0;
  if (flags[i]) {
    const uint64_t key = keys[i];
    float sx = 0.f, sy = 0.f, sz = 0.f;
    int j = i;
    for (; j < n && keys[j] == key; ++j) {
      const float4 v = p[idx[j]];
      sx += v.x; sy += v.y; sz += v.z;
    }
    if (j - i >= min_points) {
      const float c = (float)(j - i);
      cent[i] = make_float4(sx / c, sy / c, sz / c, 1.0f);
      k = 1;
    }
  }
  kept[i] = k;
}

// ---------------------------------------------------------------- input filter chain (include/lsgpu_icp.h)
// One predicate per launch: keep[i] = 1 if point i survives filter f.  laser_slam/src/laser_track.cpp:146
// (input_filters_.apply(scan.scan)); semantics restated in the header.
struct PointFilterDev {
  int type, dim, flag;
  float v[6];
  uint32_t step, phase;  // FixStepSampling
};

__global__ __launch_bounds__(256) void k_point_filter_select(const float4* __restrict__ src, int n, PointFilterDev f,
                                                             const float* __restrict__ draws,
                                                             uint32_t* __restrict__ keep) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 p = src[i];
  bool k = true;
  if (f.type == 1 || f.type == 2) {  // Max / MinDist
    // upstream asymmetry (libpointmatcher MaxDist.cpp / MinDist.cpp, from knowledge): the radial branch of both compares
    // the norm with |limit|; on ONE axis MaxDist compares the SIGNED coordinate (x < maxDist keeps every negative x),
    // MinDist the absolute one
    const float c = f.dim == 0 ? p.x : f.dim == 1 ? p.y : p.z;
    if (f.dim < 0) {
      const float val = sqrtf(__fmaf_rn(p.z, p.z, __fmaf_rn(p.y, p.y, p.x * p.x)));
      k = f.type == 1 ? val < fabsf(f.v[0]) : val > fabsf(f.v[0]);
    } else {
      k = f.type == 1 ? c < f.v[0] : fabsf(c) > f.v[0];
    }
  } else if (f.type == 3) {          // BoundingBox
    const bool in = p.x > f.v[0] && p.x < f.v[1] && p.y > f.v[2] && p.y < f.v[3] && p.z > f.v[4] && p.z < f.v[5];
    k = f.flag ? !in : in;
  } else if (f.type == 4) {          // FixStepSampling
    k = (uint32_t)i >= f.phase && ((uint32_t)i - f.phase) % f.step == 0u;
  } else if (f.type == 5) {          // RandomSampling
    k = draws[i] < f.v[0];
  } else if (f.type == 6) {          // RemoveNaN: a column with a NaN in any feature row goes (colArray == colArray).all()
    k = p.x == p.x && p.y == p.y && p.z == p.z && p.w == p.w;
  }
  keep[i] = k ? 1u : 0u;
}

// ---------------------------------------------------------------- PointCloud2 <-> x,y,z,1 (include/lsgpu_icp.h)
__device__ __forceinline__ float pc2_field(const unsigned char* __restrict__ rec, int off, int big) {
  const uint32_t b0 = rec[off], b1 = rec[off + 1], b2 = rec[off + 2], b3 = rec[off + 3];  // any alignment
  const uint32_t u = big ? (b0 << 24) | (b1 << 16) | (b2 << 8) | b3 : b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
  return __uint_as_float(u);
}
// rosMsgToPointMatcherCloud<float> (laser_slam_worker.cpp:125): record i -> out[i] = {x, y, z, 1}; keep[i] = finite
__global__ __launch_bounds__(256) void k_pc2_unpack(const unsigned char* __restrict__ data, int n, int step, int ox,
                                                    int oy, int oz, int big, int drop, float4* __restrict__ out,
                                                    uint32_t* __restrict__ keep) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const unsigned char* rec = data + (size_t)i * (size_t)step;
  const float x = pc2_field(rec, ox, big), y = pc2_field(rec, oy, big), z = pc2_field(rec, oz, big);
  out[i] = make_float4(x, y, z, 1.f);
  const bool fin = isfinite(x) && isfinite(y) && isfinite(z);
  keep[i] = (!drop || fin) ? 1u : 0u;
}
// lpmToPcl / pcl::toROSMsg<PointXYZ> (common.hpp:159-191): the same 16-byte records, padding written as 1
__global__ __launch_bounds__(256) void k_pc2_pack(const float4* __restrict__ in, int64_t n, float4* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 p = in[i];
  out[i] = make_float4(p.x, p.y, p.z, 1.f);
}

}  // namespace lsgpu
