// lsgpu_scan.hip.h -- prefix sums of uint32 arrays on the handle's stream: the compaction step behind the chunk
// boundaries of KDTreeMatcher::init's stand-in (lsgpu_grid.hip.h), the sampling filters' survivors (lsgpu_ssn.hip.h,
// icp_default.yaml:1-7), the input filter chain and the voxel grid.  Plain HBM-bound integer work: the input is read
// twice, the output written once.
//
// Three launches for n > kScanTile (one for smaller inputs):
//   k_scan_sums   block b adds up its tile of kScanTile elements
//   k_scan_top    ONE block turns the tile sums into exclusive tile offsets, in place
//   k_scan_write  block b scans its tile again, starting from its offset, and writes the inclusive / exclusive sums
// A tile is walked in four coalesced sub-tiles of 1024 elements (one dwordx4 per thread), the running total carried
// from one to the next.  Sums wrap modulo 2^32 like any unsigned prefix sum.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lsgpu {

constexpr int kScanSub = 1024;              // elements per sub-tile: 256 threads x 4
constexpr int kScanTile = 4 * kScanSub;     // elements per block

__device__ __forceinline__ uint32_t scan_wave_incl(uint32_t v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

__device__ __forceinline__ uint4 scan_load4(const uint32_t* __restrict__ in, size_t i, size_t n) {
  if (i + 4 <= n && ((reinterpret_cast<uintptr_t>(in + i) & 15u) == 0)) return *reinterpret_cast<const uint4*>(in + i);
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (i < n) v.x = in[i];
  if (i + 1 < n) v.y = in[i + 1];
  if (i + 2 < n) v.z = in[i + 2];
  if (i + 3 < n) v.w = in[i + 3];
  return v;
}

// NZ: the input is read as flags -- any non-zero word counts 1 (a compaction whose flag word carries something else as well,
// e.g. the surface-normal filter's "kept, and this is your box": one scattered store per point instead of two)
__device__ __forceinline__ uint4 scan_flags(uint4 v) {
  return make_uint4(v.x ? 1u : 0u, v.y ? 1u : 0u, v.z ? 1u : 0u, v.w ? 1u : 0u);
}

template <bool NZ = false>
__global__ __launch_bounds__(256) void k_scan_sums(const uint32_t* __restrict__ in, size_t n, uint32_t* __restrict__ sums) {
  __shared__ uint32_t ws[4];
  const size_t base = (size_t)blockIdx.x * kScanTile;
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    uint4 v = scan_load4(in, base + (size_t)k * kScanSub + 4u * threadIdx.x, n);
    if (NZ) v = scan_flags(v);
    s += v.x + v.y + v.z + v.w;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += (uint32_t)__shfl_xor((int)s, o, 64);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// one block: sums[0..nb) -> exclusive prefix, in place
__global__ __launch_bounds__(1024) void k_scan_top(uint32_t* __restrict__ sums, int nb) {
  __shared__ uint32_t ws[16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t carry = 0;
  for (int b0 = 0; b0 < nb; b0 += 1024) {
    const int b = b0 + (int)threadIdx.x;
    const uint32_t v = b < nb ? sums[b] : 0u;
    const uint32_t incl = scan_wave_incl(v, lane);
    if (lane == 63) ws[w] = incl;
    __syncthreads();
    uint32_t before = carry, all = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const uint32_t t = ws[k]; before += k < w ? t : 0u; all += t; }
    if (b < nb) sums[b] = before + incl - v;
    carry += all;
    __syncthreads();
  }
}

template <bool INCLUSIVE, bool NZ = false>
__global__ __launch_bounds__(256) void k_scan_write(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n,
                                                    const uint32_t* __restrict__ offsets /* nullable: single tile */) {
  __shared__ uint32_t ws[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t base = (size_t)blockIdx.x * kScanTile;
  uint32_t carry = offsets ? offsets[blockIdx.x] : 0u;
#pragma unroll 1
  for (int k = 0; k < 4; ++k) {
    const size_t i = base + (size_t)k * kScanSub + 4u * threadIdx.x;
    uint4 v = scan_load4(in, i, n);
    if (NZ) v = scan_flags(v);
    const uint32_t t = v.x + v.y + v.z + v.w;
    const uint32_t incl = scan_wave_incl(t, lane);
    if (lane == 63) ws[w] = incl;
    __syncthreads();
    uint32_t before = carry;
    const uint32_t s0 = ws[0], s1 = ws[1], s2 = ws[2], s3 = ws[3];
    before += (w > 0 ? s0 : 0u) + (w > 1 ? s1 : 0u) + (w > 2 ? s2 : 0u);
    const uint32_t e0 = before + incl - t;          // exclusive sum in front of this thread's four
    uint4 o;
    if (INCLUSIVE) { o.x = e0 + v.x; o.y = o.x + v.y; o.z = o.y + v.z; o.w = o.z + v.w; }
    else { o.x = e0; o.y = e0 + v.x; o.z = o.y + v.y; o.w = o.z + v.z; }
    if (i + 4 <= n && ((reinterpret_cast<uintptr_t>(out + i) & 15u) == 0)) {
      *reinterpret_cast<uint4*>(out + i) = o;
    } else {
      if (i < n) out[i] = o.x;
      if (i + 1 < n) out[i + 1] = o.y;
      if (i + 2 < n) out[i + 2] = o.z;
      if (i + 3 < n) out[i + 3] = o.w;
    }
    carry += s0 + s1 + s2 + s3;
    __syncthreads();
  }
}

}  // namespace lsgpu
