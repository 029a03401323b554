// Stable LSD radix sort of (64-bit key, 32-bit value) pairs, 8 key bits per pass -- the sort behind the grid build
// (Morton keys of the reference, KDTreeMatcher::init's role), the query ordering, the levels of the
// surface-normal filter and the voxel grid.
//
// Why not the library sort: for 1 M pairs rocPRIM picks its merge sort (one block sort + 10 merge passes of 3 launches,
// 260 us for 48 key bits); its onesweep radix sort was measured 10 % slower still at this size.  The sort is plain
// HBM-bound integer work: per pass the keys are read twice and the pairs written once (32 B per pair).
//
// Per pass, three launches on the handle's stream:
//   k_rs_hist     per block (256 threads x ITEMS keys) the histogram of the pass's digit -> blockhist[digit][block]
//   k_rs_scan     one block per digit: exclusive prefix over the blocks (in place) + the digit's total
//                 (Leaving this launch out was tried twice and measured slower both times: as the tail of k_rs_hist -- last
//                 block, ticket -- 2 %; and with two-level counts -- k_rs_hist adds to per-group-of-32-blocks totals,
//                 the scatter's thread d sums <= 32 groups + <= 31 blocks with 16 16-byte loads -- filters + grid
//                 1.94 -> 2.13 ms per 1 M-point compute, profiles/r03_knn_variants.txt r03r.)
//   k_rs_scatter  every block ranks its keys again -- wave by wave, 64 consecutive keys at a time: the lanes holding the
//                 same digit find each other with 8 ballots, the lowest of them advances the wave's counter of that digit
//                 in LDS -- which gives each pair its slot  digit base + blocks before + waves before + rank  and keeps
//                 equal digits in input order (stable), hence the whole sort stable and identical to any other stable
//                 sort of the same keys.  The block's pairs go through LDS in digit order first, so that consecutive
//                 threads store to consecutive addresses of each digit's run (round 3; straight from the registers every
//                 lane of a store hit its own cache line).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lsgpu {

__device__ __forceinline__ uint32_t wave_scan_incl_u32(uint32_t v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

template <int ITEMS>
__global__ __launch_bounds__(256) void k_rs_hist(const uint64_t* __restrict__ keys, int64_t n, int shift,
                                                 uint32_t mask, uint32_t* __restrict__ blockhist, int nblocks) {
  __shared__ uint32_t hist[256];
  hist[threadIdx.x] = 0u;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t base = (int64_t)blockIdx.x * (256 * ITEMS) + (int64_t)w * (64 * ITEMS) + lane;
  uint64_t k[ITEMS];
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int64_t idx = base + i * 64;
    k[i] = idx < n ? keys[idx] : 0ull;
  }
#pragma unroll
  for (int i = 0; i < ITEMS; ++i)
    if (base + i * 64 < n) atomicAdd(&hist[(uint32_t)(k[i] >> shift) & mask], 1u);
  __syncthreads();
  blockhist[(size_t)threadIdx.x * nblocks + blockIdx.x] = hist[threadIdx.x];
}

// block d: blockhist[d][0..nblocks) -> exclusive prefix over the blocks; dtot[d] = the digit's total
__global__ __launch_bounds__(256) void k_rs_scan(uint32_t* __restrict__ blockhist, int nblocks,
                                                 uint32_t* __restrict__ dtot) {
  __shared__ uint32_t wsum[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t* row = blockhist + (size_t)blockIdx.x * nblocks;
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
__global__ __launch_bounds__(256) void k_rs_scatter(const uint64_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                    uint64_t* __restrict__ kout, uint32_t* __restrict__ vout,
                                                    int64_t n, int shift, uint32_t mask,
                                                    const uint32_t* __restrict__ blockpref,
                                                    const uint32_t* __restrict__ dtot, int nblocks) {
  __shared__ uint32_t cnt[4][256];   // per wave and digit: keys seen so far, then: keys of the waves before
  __shared__ uint32_t base_sh[256];  // per digit: first output position of this block's keys, minus their first local slot
  __shared__ uint32_t dstart[256];   // per digit: first local slot of this block's keys (exclusive prefix of the block's digit counts)
  __shared__ uint32_t wtot[4], wtot2[4];
  // the block's pairs in digit order: stores straight from the registers hit a different cache line per lane (the digits of
  // 64 consecutive keys are unrelated); staged through LDS, consecutive threads write consecutive addresses of each digit's run
  __shared__ uint64_t skey[256 * ITEMS];
  __shared__ uint32_t sval[256 * ITEMS];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 4; ++i) cnt[i][threadIdx.x] = 0u;
  const int64_t blk0 = (int64_t)blockIdx.x * (256 * ITEMS);
  const int64_t base = blk0 + (int64_t)w * (64 * ITEMS) + lane;
  uint64_t key[ITEMS];
  uint32_t val[ITEMS], rank[ITEMS];
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int64_t idx = base + i * 64;
    key[i] = idx < n ? kin[idx] : 0ull;
    val[i] = idx < n ? vin[idx] : 0u;
  }
  __syncthreads();
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const bool valid = base + i * 64 < n;
    const uint32_t dig = (uint32_t)(key[i] >> shift) & mask;
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (dig >> b) & 1u;
      const unsigned long long m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    const int leader = valid ? __ffsll((long long)peers) - 1 : lane;  // (a valid lane is among its own peers)
    uint32_t old = 0u;
    if (valid && lane == leader) {
      old = cnt[w][dig];
      cnt[w][dig] = old + (uint32_t)__popcll(peers);
    }
    old = (uint32_t)__shfl((int)old, leader, 64);
    rank[i] = old + (uint32_t)__popcll(peers & lt);
    // the next group's leader may be another lane reading the counter this one just wrote: LDS accesses of a wave
    // complete in order, the compiler only has to keep them in program order
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
    // two block-wide exclusive scans over the digits: the digit totals of the whole input (global base) and of this block
    // (local slot of the digit's run)
    const uint32_t tot = dtot[d];
    const uint32_t incl = wave_scan_incl_u32(tot, lane);
    const uint32_t incl2 = wave_scan_incl_u32(run, lane);
    if (lane == 63) { wtot[w] = incl; wtot2[w] = incl2; }
    __syncthreads();
    uint32_t before = 0u, before2 = 0u;
    for (int ww = 0; ww < w; ++ww) { before += wtot[ww]; before2 += wtot2[ww]; }
    const uint32_t ds = before2 + incl2 - run;
    dstart[d] = ds;
    base_sh[d] = before + incl - tot + blockpref[(size_t)d * nblocks + blockIdx.x] - ds;   // (+ local slot = output position; wraps harmlessly)
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    if (base + i * 64 < n) {
      const uint32_t dig = (uint32_t)(key[i] >> shift) & mask;
      const uint32_t slot = dstart[dig] + cnt[w][dig] + rank[i];
      skey[slot] = key[i];
      sval[slot] = val[i];
    }
  }
  __syncthreads();
  const int64_t left = n - blk0;
  const uint32_t nloc = left >= (int64_t)(256 * ITEMS) ? (uint32_t)(256 * ITEMS) : (uint32_t)left;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const uint32_t slot = (uint32_t)i * 256u + threadIdx.x;
    if (slot < nloc) {
      const uint64_t k = skey[slot];
      const uint32_t pos = base_sh[(uint32_t)(k >> shift) & mask] + slot;
      kout[pos] = k;
      vout[pos] = sval[slot];
    }
  }
}

}  // namespace lsgpu
