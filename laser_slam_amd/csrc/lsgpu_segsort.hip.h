// lsgpu_segsort.hip.h -- the sorts behind the levels of SamplingSurfaceNormalDataPointsFilter
// (laser_slam/configurations/icp_default.yaml:5-7; lsgpu_ssn.hip.h): every segment of a level that still splits has to be
// put into the stable order of its cut coordinate.  Round 1-3 sorted the WHOLE cloud per level by (segment, coordinate),
// 32 + L key bits = five 8-bit passes.  Two facts make most of that unnecessary:
//   * segments are contiguous ranges of the current order, so the segment number need not be part of the key: a
//     SEGMENTED stable LSD sort (digit counts and output positions per segment) needs the four passes of the 32-bit
//     ordered coordinate only, on (uint32 key, uint32 value) pairs instead of (uint64, uint32);
//   * a child is a contiguous half of its parent's sorted order, i.e. it is already stably sorted by the parent's cut
//     axis: if it cuts along the SAME axis again (long clouds: a street scanned from its middle cuts x for the first
//     levels) its sort is the identity and is skipped.  On the benchmark scan 3.5 of the first 7 levels' sorts remain,
//     on a three-scan sub-map 4.3 of 9 (devtools study in DESIGN.md).
// Blocks: a segment that sorts gets ceil(count / tile) blocks of its own (k_ssn_plan writes the block table on the
// device; the host launches the worst-case grid, surplus blocks exit).  Per pass, three launches as in lsgpu_sort.hip.h:
//   k_seg_hist     block b: histogram of the pass's digit over its elements -> blockhist[digit][b]
//   k_seg_scan     block d: exclusive prefix of blockhist[d][0 .. nblocks) over ALL blocks + the column total; the
//                  prefix inside a segment is the difference to the entry of the segment's first block
//   k_seg_scatter  ranks like k_rs_scatter (wave-level digit matching, LDS-staged stores); positions are relative to the
//                  segment's start, so nothing ever leaves its segment
// An even number of passes brings the sorted segments back into the buffers they came from; skipped segments are never
// touched.  Same stable order as any other stable sort of the same keys: bit-identical filter output.
#pragma once
#include "lsgpu_sort.hip.h"

namespace lsgpu {

struct SegBlock {            // 32 bytes
  uint32_t first, count;     // elements [first, first + count) of the arrays
  uint32_t seg_start;        // first element of the block's segment
  uint32_t fb, nb;           // the segment's blocks: [fb, fb + nb)
  uint32_t pad[3];
};

template <int ITEMS>
__global__ __launch_bounds__(256) void k_seg_hist(const uint32_t* __restrict__ keys, const SegBlock* __restrict__ tab,
                                                  const uint32_t* __restrict__ nblocks_dev, int shift,
                                                  uint32_t* __restrict__ blockhist, int cap) {
  if (blockIdx.x >= *nblocks_dev) return;
  __shared__ uint32_t hist[256];
  hist[threadIdx.x] = 0u;
  __syncthreads();
  const SegBlock sb = tab[blockIdx.x];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t base = (uint32_t)(w * (64 * ITEMS) + lane);
  uint32_t k[ITEMS];
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const uint32_t li = base + (uint32_t)i * 64u;
    k[i] = li < sb.count ? keys[sb.first + li] : 0u;
  }
#pragma unroll
  for (int i = 0; i < ITEMS; ++i)
    if (base + (uint32_t)i * 64u < sb.count) atomicAdd(&hist[(k[i] >> shift) & 255u], 1u);
  __syncthreads();
  blockhist[(size_t)threadIdx.x * cap + blockIdx.x] = hist[threadIdx.x];
}

// block d: blockhist[d][0 .. nblocks) -> exclusive prefix over the blocks (in place); dtot[d] = the column's total
__global__ __launch_bounds__(256) void k_seg_scan(uint32_t* __restrict__ blockhist, int cap, const uint32_t* __restrict__ nblocks_dev,
                                                  uint32_t* __restrict__ dtot) {
  __shared__ uint32_t wsum[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nblocks = (int)*nblocks_dev;
  uint32_t* row = blockhist + (size_t)blockIdx.x * cap;
  uint32_t carry = 0u;
  for (int b0 = 0; b0 < nblocks; b0 += 256) {
    const int b = b0 + (int)threadIdx.x;
    const uint32_t v = b < nblocks ? row[b] : 0u;
    const uint32_t incl = wave_scan_incl_u32(v, lane);
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t before = carry;
    for (int ww = 0; ww < w; ++ww) before += wsum[ww];
    if (b < nblocks) row[b] = before + incl - v;
    carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) dtot[blockIdx.x] = carry;
}

template <int ITEMS>
__global__ __launch_bounds__(256) void k_seg_scatter(const uint32_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                     uint32_t* __restrict__ kout, uint32_t* __restrict__ vout,
                                                     const SegBlock* __restrict__ tab, const uint32_t* __restrict__ nblocks_dev,
                                                     int shift, const uint32_t* __restrict__ blockpref,
                                                     const uint32_t* __restrict__ dtot, int cap) {
  const uint32_t nblocks = *nblocks_dev;
  if (blockIdx.x >= nblocks) return;
  __shared__ uint32_t cnt[4][256];   // per wave and digit: keys seen so far, then: keys of the waves before
  __shared__ uint32_t base_sh[256];  // per digit: first output position of this block's keys, minus their first local slot
  __shared__ uint32_t dstart[256];   // per digit: first local slot of this block's keys
  __shared__ uint32_t wtot[4], wtot2[4];
  __shared__ uint32_t skey[256 * ITEMS];
  __shared__ uint32_t sval[256 * ITEMS];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 4; ++i) cnt[i][threadIdx.x] = 0u;
  const SegBlock sb = tab[blockIdx.x];
  const uint32_t base = (uint32_t)(w * (64 * ITEMS) + lane);
  uint32_t key[ITEMS], val[ITEMS], rank[ITEMS];
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const uint32_t li = base + (uint32_t)i * 64u;
    key[i] = li < sb.count ? kin[sb.first + li] : 0u;
    val[i] = li < sb.count ? vin[sb.first + li] : 0u;
  }
  __syncthreads();
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const bool valid = base + (uint32_t)i * 64u < sb.count;
    const uint32_t dig = (key[i] >> shift) & 255u;
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (dig >> b) & 1u;
      const unsigned long long m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    const int leader = valid ? __ffsll((long long)peers) - 1 : lane;
    uint32_t old = 0u;
    if (valid && lane == leader) {
      old = cnt[w][dig];
      cnt[w][dig] = old + (uint32_t)__popcll(peers);
    }
    old = (uint32_t)__shfl((int)old, leader, 64);
    rank[i] = old + (uint32_t)__popcll(peers & lt);
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  }
  __syncthreads();
  {
    const int d = (int)threadIdx.x;
    uint32_t run = 0u;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) {
      const uint32_t t = cnt[ww][d];
      cnt[ww][d] = run;
      run += t;
    }
    // the digit's total over the SEGMENT and the keys of the segment's earlier blocks: differences of the column prefix
    const uint32_t* col = blockpref + (size_t)d * cap;
    const uint32_t p_fb = col[sb.fb];
    const uint32_t end = sb.fb + sb.nb;
    const uint32_t p_end = end < nblocks ? col[end] : dtot[d];
    const uint32_t tot = p_end - p_fb;
    const uint32_t incl = wave_scan_incl_u32(tot, lane);
    const uint32_t incl2 = wave_scan_incl_u32(run, lane);
    if (lane == 63) { wtot[w] = incl; wtot2[w] = incl2; }
    __syncthreads();
    uint32_t before = 0u, before2 = 0u;
    for (int ww = 0; ww < w; ++ww) { before += wtot[ww]; before2 += wtot2[ww]; }
    const uint32_t ds = before2 + incl2 - run;
    dstart[d] = ds;
    base_sh[d] = sb.seg_start + (before + incl - tot) + (col[blockIdx.x] - p_fb) - ds;   // (+ local slot = output position; wraps harmlessly)
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    if (base + (uint32_t)i * 64u < sb.count) {
      const uint32_t dig = (key[i] >> shift) & 255u;
      const uint32_t slot = dstart[dig] + cnt[w][dig] + rank[i];
      skey[slot] = key[i];
      sval[slot] = val[i];
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const uint32_t slot = (uint32_t)i * 256u + threadIdx.x;
    if (slot < sb.count) {
      const uint32_t k = skey[slot];
      const uint32_t pos = base_sh[(k >> shift) & 255u] + slot;
      kout[pos] = k;
      vout[pos] = sval[slot];
    }
  }
}

}  // namespace lsgpu
