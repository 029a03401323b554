// lsgpu_ssn_levels.hip.h -- the UPPER levels of SamplingSurfaceNormalDataPointsFilter's box tree (segments too large for
// one workgroup) from three presorted axes, in global memory (icp_default.yaml:5-7; the reference filter of
// PointMatcher::ICP::compute, laser_slam/src/laser_track.cpp:496).  Round 5.
//
// Rounds 1-4 sorted at every level: a segmented stable radix sort of the cut coordinate (lsgpu_segsort.hip.h), four
// passes of three launches plus key / plan / split / assign kernels = 16 dependent launches per level, ~100 us per level
// at 1 M points whatever the amount of data (profiles/r04_bench.stats.txt: 252 k_seg_* launches per compute).  A chain
// of short dependent launches is bound by launch latency, so the cure is a shorter chain:
//   * the three axes are sorted ONCE, up front -- three independent sorts, enqueued on three streams, so the chain is as
//     long as ONE sort -- into lists of (point, ordered key) pairs: list d = the cloud in the stable order of coordinate d;
//     every segment owns the same index range in all three lists;
//   * a level is then k_gt_plan, k_gt_fix (the cut axis' list into the OUT buffers, tie runs in the segment's current
//     order; one byte per point: which child), k_gt_count / k_gt_scan / k_gt_part (one stable partition of the other two
//     lists by child) and k_gt_split (children's boxes): six launches, no sort, every stream access by position coalesced.
// The scheme is k_ssn_tree's (lsgpu_ssn_tree.hip.h) with one difference: there is no cur_pos array -- keeping it would
// cost a random 4-byte gather and scatter per point and level (the first version did: 80 us per k_gt_fix).  The members
// of a tie run (equal cut coordinates, found by comparing NEIGHBOURING keys of the list) are ordered by what the
// segment's current order IS: the keys on the axes the segment was cut along before, most recent first, then the
// original index (the segment's "signature", three bytes per segment).  Both variants are modelled step for step in
// tests/ssn_tree_model.py and checked there against the chain of stable sorts the restatement defines.
// Tie runs are walked element by element: runs longer than kGtRunCap (a cloud with thousands of EQUAL coordinates along
// its widest axis) raise a flag and the host repeats the filter with the segmented sorts, which do not care
// (LSGPU_SSN_SORT_LEVELS selects them always).  Bit-identical to the segmented sorts and to the oracle.
#pragma once
#include "lsgpu_ssn.hip.h"
#include "lsgpu_ssn_tree.hip.h"

namespace lsgpu {

constexpr int kGtRunCap = 256;
constexpr uint32_t kGtNoAxis = 0xFFu;

struct GtLists {            // one buffer set: per axis the points and their ordered keys, in list order
  uint32_t* e[3];
  uint32_t* k[3];
};
// (selects, not g.e[d]: a dynamically indexed kernel argument would be copied to scratch memory)
__device__ __forceinline__ uint32_t* gt_e(const GtLists& g, int d) { return d == 0 ? g.e[0] : d == 1 ? g.e[1] : g.e[2]; }
__device__ __forceinline__ uint32_t* gt_k(const GtLists& g, int d) { return d == 0 ? g.k[0] : d == 1 ? g.k[1] : g.k[2]; }

// a segment's signature: the axes it was cut along, most recent first, one byte each (kGtNoAxis: none); 0xFFFFFFFF at the root
__device__ __forceinline__ uint32_t gt_sig_push(uint32_t sig, uint32_t a) {
  const uint32_t o1 = sig & 0xFFu, o2 = (sig >> 8) & 0xFFu, o3 = (sig >> 16) & 0xFFu;
  if (o1 == a) return sig;
  const uint32_t r2 = o2 == a ? o3 : o2;               // the others, a removed (three axes: at most two remain)
  return a | (o1 << 8) | (r2 << 16) | 0xFF000000u;
}

// presort: ordered key of coordinate d, identity values
__global__ __launch_bounds__(256) void k_gt_keys(const float4* __restrict__ p, int n, int d, uint32_t* __restrict__ keys,
                                                 uint32_t* __restrict__ vals) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  keys[i] = float_order_key(coord_of(p[i], d));
  vals[i] = (uint32_t)i;
}

// block table of ONE segment [0, n): the presort is a segmented sort with a single segment
__global__ __launch_bounds__(256) void k_gt_fulltab(int n, SegBlock* __restrict__ tab, uint32_t* __restrict__ nblocks_dev) {
  const uint32_t nb = ((uint32_t)n + kSegTile - 1u) / kSegTile;
  for (uint32_t b = blockIdx.x * 256 + threadIdx.x; b < nb; b += gridDim.x * 256) {
    SegBlock e;
    e.first = b * kSegTile; e.count = min(kSegTile, (uint32_t)n - b * kSegTile);
    e.seg_start = 0u; e.fb = 0u; e.nb = nb; e.pad[0] = e.pad[1] = e.pad[2] = 0u;
    tab[b] = e;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *nblocks_dev = nb;
}

// ONE block: every segment of the level that still splits gets its blocks of kSegTile positions (all of them do at the
// global levels: a segment there holds more points than a workgroup of k_ssn_tree).  pad[0] of a block = its segment.
__global__ __launch_bounds__(256) void k_gt_plan(const SsnSeg* __restrict__ segs, int ns, int knn, SegBlock* __restrict__ tab,
                                                 uint32_t* __restrict__ nblocks_dev) {
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t big_list[256], big_fb[256];
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
      nb = sg.count > (uint32_t)knn ? (sg.count + kSegTile - 1u) / kSegTile : 0u;
    }
    const uint32_t incl = wave_scan_incl_u32(nb, lane);
    if (lane == 63) wsum[w] = incl;
    if (threadIdx.x == 0) big_n = 0u;
    __syncthreads();
    uint32_t before = carry;
    for (int ww = 0; ww < w; ++ww) before += wsum[ww];
    const uint32_t fb = before + incl - nb;
    // the blocks of this segment: a few -> this thread; many (the top levels) -> the whole block, below
    if (nb > 8u) {
      const uint32_t q = atomicAdd(&big_n, 1u);
      big_list[q] = (uint32_t)s; big_fb[q] = fb;
    } else {
      for (uint32_t k = 0; k < nb; ++k) {
        SegBlock e;
        e.first = sg.start + k * kSegTile; e.count = min(kSegTile, sg.count - k * kSegTile);
        e.seg_start = sg.start; e.fb = fb; e.nb = nb; e.pad[0] = (uint32_t)s; e.pad[1] = sg.count; e.pad[2] = 0u;
        tab[fb + k] = e;
      }
    }
    carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    const uint32_t nbig = big_n;
    for (uint32_t q = 0; q < nbig; ++q) {
      const uint32_t sb = big_list[q], gfb = big_fb[q];
      const SsnSeg g = segs[sb];
      const uint32_t gnb = (g.count + kSegTile - 1u) / kSegTile;
      for (uint32_t k = threadIdx.x; k < gnb; k += 256u) {
        SegBlock e;
        e.first = g.start + k * kSegTile; e.count = min(kSegTile, g.count - k * kSegTile);
        e.seg_start = g.start; e.fb = gfb; e.nb = gnb; e.pad[0] = sb; e.pad[1] = g.count; e.pad[2] = 0u;
        tab[gfb + k] = e;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *nblocks_dev = carry;
}

// step 1 of a level: the cut axis' list of every segment into the OUT buffers with its tie runs in the segment's current
// order, and one byte per point: does it go to the right child?
__global__ __launch_bounds__(256) void k_gt_fix(const SegBlock* __restrict__ tab, const uint32_t* __restrict__ nblocks_dev,
                                                const SsnSeg* __restrict__ segs, const uint32_t* __restrict__ sig, GtLists in, GtLists out,
                                                const float4* __restrict__ p, unsigned char* __restrict__ side, uint32_t* __restrict__ err) {
  if (blockIdx.x >= *nblocks_dev) return;
  const SegBlock sb = tab[blockIdx.x];
  const SsnSeg sg = segs[sb.pad[0]];
  const int a = ssn_cut_axis(sg);
  const uint32_t sv = sig[sb.pad[0]];
  const uint32_t o1 = sv & 0xFFu, o2 = (sv >> 8) & 0xFFu, o3 = (sv >> 16) & 0xFFu;
  const bool fix = o1 != kGtNoAxis && o1 != (uint32_t)a;
  // the axes that define the current order, the cut axis left out (its keys are equal inside a tie run)
  const uint32_t x1 = o1, x2 = o2 == (uint32_t)a ? o3 : o2;
  const uint32_t* __restrict__ ea = gt_e(in, a);
  const uint32_t* __restrict__ ka = gt_k(in, a);
  uint32_t* __restrict__ eo = gt_e(out, a);
  uint32_t* __restrict__ ko = gt_k(out, a);
  const uint32_t s0 = sg.start, s1 = sg.start + sg.count;
  const uint32_t left = sg.count - sg.count / 2u;
  for (uint32_t j = threadIdx.x; j < sb.count; j += 256u) {
    const uint32_t i = sb.first + j;
    const uint32_t e = ea[i], k = ka[i];
    uint32_t np = i;
    if (fix && ((i > s0 && ka[i - 1] == k) || (i + 1 < s1 && ka[i + 1] == k))) {
      uint32_t lo = i, hi = i + 1;
      while (lo > s0 && i - lo <= (uint32_t)kGtRunCap && ka[lo - 1] == k) --lo;
      while (hi < s1 && hi - i <= (uint32_t)kGtRunCap && ka[hi] == k) ++hi;
      if (hi - lo > (uint32_t)kGtRunCap) {
        *err = 1u;                 // the host repeats the filter with the segmented sorts
      } else {
        const float4 pe = p[e];
        const uint32_t ke1 = float_order_key(coord_of(pe, (int)x1));
        const uint32_t ke2 = x2 != kGtNoAxis ? float_order_key(coord_of(pe, (int)x2)) : 0u;
        uint32_t c = 0u;
        for (uint32_t q = lo; q < hi; ++q) {
          const uint32_t f = ea[q];
          if (f == e) continue;
          const float4 pf = p[f];
          const uint32_t kf1 = float_order_key(coord_of(pf, (int)x1));
          const uint32_t kf2 = x2 != kGtNoAxis ? float_order_key(coord_of(pf, (int)x2)) : 0u;
          const bool less = kf1 != ke1 ? kf1 < ke1 : kf2 != ke2 ? kf2 < ke2 : f < e;
          c += less ? 1u : 0u;
        }
        np = lo + c;
      }
    }
    eo[np] = e; ko[np] = k;
    side[e] = np - s0 >= left ? 1 : 0;
  }
}

// this thread's position of the block in wave-contiguous order (stable scans need positions in order): wave w of the
// block owns 512 consecutive positions, 64 at a time
__device__ __forceinline__ uint32_t gt_pos(int w, int k, int lane) { return (uint32_t)(w * 512 + k * 64 + lane); }

// step 2: per block, how many of its positions go to the right child in each of the other two lists
__global__ __launch_bounds__(256) void k_gt_count(const SegBlock* __restrict__ tab, const uint32_t* __restrict__ nblocks_dev,
                                                  const SsnSeg* __restrict__ segs, GtLists in, const unsigned char* __restrict__ side,
                                                  uint32_t* __restrict__ cnt /* 2 x cap: low list, high list */, int cap) {
  if (blockIdx.x >= *nblocks_dev) return;
  __shared__ uint32_t ws[4];
  const SegBlock sb = tab[blockIdx.x];
  const SsnSeg sg = segs[sb.pad[0]];
  const int a = ssn_cut_axis(sg);
  const int d1 = a == 2 ? 0 : a + 1, d2 = a == 0 ? 2 : a - 1;
  const uint32_t* __restrict__ l1 = gt_e(in, d1);
  const uint32_t* __restrict__ l2 = gt_e(in, d2);
  uint32_t v = 0u;
  for (uint32_t j = threadIdx.x; j < sb.count; j += 256u) {
    const uint32_t i = sb.first + j;
    v += (uint32_t)side[l1[i]] | ((uint32_t)side[l2[i]] << 16);
  }
  v = wave_sum_u32(v);     // (halves cannot overflow: a block holds 2048 positions)
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t t = ws[0] + ws[1] + ws[2] + ws[3];
    cnt[blockIdx.x] = t & 0xFFFFu;
    cnt[cap + blockIdx.x] = t >> 16;
  }
}

// step 3: ONE block: exclusive prefix of both count arrays over the level's blocks, in place
__global__ __launch_bounds__(1024) void k_gt_scan(uint32_t* __restrict__ cnt, int cap, const uint32_t* __restrict__ nblocks_dev) {
  __shared__ uint32_t ws[2][16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nb = (int)*nblocks_dev;
  uint32_t carry0 = 0u, carry1 = 0u;
  for (int b0 = 0; b0 < nb; b0 += 1024) {
    const int b = b0 + (int)threadIdx.x;
    const uint32_t v0 = b < nb ? cnt[b] : 0u, v1 = b < nb ? cnt[cap + b] : 0u;
    const uint32_t i0 = wave_scan_incl_u32(v0, lane), i1 = wave_scan_incl_u32(v1, lane);
    if (lane == 63) { ws[0][w] = i0; ws[1][w] = i1; }
    __syncthreads();
    uint32_t b0s = carry0, b1s = carry1, a0 = 0u, a1 = 0u;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const uint32_t t0 = ws[0][k], t1 = ws[1][k];
      b0s += k < w ? t0 : 0u; b1s += k < w ? t1 : 0u;
      a0 += t0; a1 += t1;
    }
    if (b < nb) { cnt[b] = b0s + i0 - v0; cnt[cap + b] = b1s + i1 - v1; }
    carry0 += a0; carry1 += a1;
    __syncthreads();
  }
}

// step 4: the stable partition itself: both other lists into the OUT buffers, every point to its child's range; the
// positions' segment numbers for the next level
__global__ __launch_bounds__(256) void k_gt_part(const SegBlock* __restrict__ tab, const uint32_t* __restrict__ nblocks_dev,
                                                 const SsnSeg* __restrict__ segs, GtLists in, GtLists out,
                                                 const unsigned char* __restrict__ side, const uint32_t* __restrict__ pref /* 2 x cap */,
                                                 int cap, uint32_t* __restrict__ seg_of) {
  if (blockIdx.x >= *nblocks_dev) return;
  __shared__ uint32_t ws[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const SegBlock sb = tab[blockIdx.x];
  const uint32_t s = sb.pad[0];
  const SsnSeg sg = segs[s];
  const int a = ssn_cut_axis(sg);
  const int d1 = a == 2 ? 0 : a + 1, d2 = a == 0 ? 2 : a - 1;
  const uint32_t left = sg.count - sg.count / 2u;
  const uint32_t* __restrict__ l1 = gt_e(in, d1);
  const uint32_t* __restrict__ l2 = gt_e(in, d2);
  const uint32_t* __restrict__ k1 = gt_k(in, d1);
  const uint32_t* __restrict__ k2 = gt_k(in, d2);
  uint32_t* __restrict__ o1 = gt_e(out, d1);
  uint32_t* __restrict__ o2 = gt_e(out, d2);
  uint32_t* __restrict__ q1 = gt_k(out, d1);
  uint32_t* __restrict__ q2 = gt_k(out, d2);
  uint32_t e1[8], e2[8], xk[8], vk[8];
  uint32_t carry = 0u;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const uint32_t j = gt_pos(w, k, lane);
    uint32_t v = 0u;
    e1[k] = e2[k] = 0u;
    if (j < sb.count) {
      const uint32_t i = sb.first + j;
      e1[k] = l1[i]; e2[k] = l2[i];
      v = (uint32_t)side[e1[k]] | ((uint32_t)side[e2[k]] << 16);
    }
    vk[k] = v;
    const uint32_t incl = tree_wave_scan(v, lane);
    xk[k] = carry + incl - v;
    carry += rl_u(incl, 63);
  }
  if (lane == 0) ws[w] = carry;
  __syncthreads();
  uint32_t before = 0u;
  for (int ww = 0; ww < w; ++ww) before += ws[ww];
  // the block's own offset inside its segment: the prefix over the level's blocks minus the entry of the segment's first
  const uint32_t seg1 = pref[blockIdx.x] - pref[sb.fb], seg2 = pref[cap + blockIdx.x] - pref[cap + sb.fb];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const uint32_t j = gt_pos(w, k, lane);
    if (j < sb.count) {
      const uint32_t i = sb.first + j;
      const uint32_t x = xk[k] + before;
      const uint32_t r1 = seg1 + (x & 0xFFFFu), r2 = seg2 + (x >> 16);
      const uint32_t off = i - sg.start;
      const uint32_t f1 = vk[k] & 1u, f2 = vk[k] >> 16;
      const uint32_t p1 = f1 ? sg.start + left + r1 : sg.start + (off - r1);
      const uint32_t p2 = f2 ? sg.start + left + r2 : sg.start + (off - r2);
      o1[p1] = e1[k]; q1[p1] = k1[i];
      o2[p2] = e2[k]; q2[p2] = k2[i];
      seg_of[i] = 2u * s + (off >= left ? 1u : 0u);
    }
  }
}

// step 5: the children (2s, 2s + 1) of every segment: boxes from the cut value = coordinate of the first point of the
// right half in the cut axis' list (OUT buffers: after the fix), their signature = the parent's with the cut axis in front
__global__ __launch_bounds__(256) void k_gt_split(const float4* __restrict__ p, GtLists lists, const SsnSeg* __restrict__ segs, int nseg,
                                                  int knn, SsnSeg* __restrict__ out, const uint32_t* __restrict__ sig_in,
                                                  uint32_t* __restrict__ sig_out) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= nseg) return;
  const SsnSeg sg = segs[s];
  SsnSeg a = sg, b = sg;
  uint32_t sv = sig_in[s];
  if (sg.count > (uint32_t)knn) {
    const int cut = ssn_cut_axis(sg);
    const uint32_t right = sg.count / 2, left = sg.count - right;
    const float cutval = coord_of(p[gt_e(lists, cut)[sg.start + left]], cut);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      a.hi[d] = d == cut ? cutval : a.hi[d];
      b.lo[d] = d == cut ? cutval : b.lo[d];
    }
    a.count = left;
    b.start = sg.start + left; b.count = right;
    sv = gt_sig_push(sv, (uint32_t)cut);
  } else {
    b.start = sg.start + sg.count; b.count = 0;
  }
  out[2 * s] = a;
  out[2 * s + 1] = b;
  sig_out[2 * s] = sv; sig_out[2 * s + 1] = sv;
}

// hand-over to k_ssn_tree: every root's points in its current order (the list of the axis it was cut along last)
__global__ __launch_bounds__(256) void k_gt_idx(int n, const uint32_t* __restrict__ seg_of, const uint32_t* __restrict__ sig, GtLists lists,
                                                uint32_t* __restrict__ idx) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t o1 = sig[seg_of[i]] & 0xFFu;
  idx[i] = o1 == kGtNoAxis ? (uint32_t)i : gt_e(lists, (int)o1)[i];
}

}  // namespace lsgpu
