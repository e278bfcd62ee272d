// scan_util.h -- wave64 / workgroup prefix-sum and reduction primitives (device only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bce {

// inclusive sum scan across the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t v) {
  const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(v, o);
    if (lane >= (uint32_t)o) v += t;
  }
  return v;
}
__device__ __forceinline__ uint64_t wave_incl_sum64(uint64_t v) {
  const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint64_t t = __shfl_up(v, o);
    if (lane >= (uint32_t)o) v += t;
  }
  return v;
}
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t v) {
  const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(v, o);
    if (lane >= (uint32_t)o) v = v > t ? v : t;
  }
  return v;
}

// Exclusive sum scan over the NT threads of a block; *total = block sum.  Two barriers.
template <int NT>
__device__ __forceinline__ uint32_t block_excl_scan_sum(uint32_t v, uint32_t *total) {
  __shared__ uint32_t ws[NT / 64];
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  const uint32_t inc = wave_incl_sum(v);
  if (lane == 63) ws[wid] = inc;
  __syncthreads();
  uint32_t wbase = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) {
    const uint32_t t = ws[i];
    if ((uint32_t)i < wid) wbase += t;
    tot += t;
  }
  __syncthreads();
  *total = tot;
  return wbase + inc - v;
}
template <int NT>
__device__ __forceinline__ uint64_t block_excl_scan_sum64(uint64_t v, uint64_t *total) {
  __shared__ uint64_t ws[NT / 64];
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  const uint64_t inc = wave_incl_sum64(v);
  if (lane == 63) ws[wid] = inc;
  __syncthreads();
  uint64_t wbase = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) {
    const uint64_t t = ws[i];
    if ((uint32_t)i < wid) wbase += t;
    tot += t;
  }
  __syncthreads();
  *total = tot;
  return wbase + inc - v;
}

// Inclusive max scan over the NT threads of a block; *total = block max.
template <int NT>
__device__ __forceinline__ uint32_t block_incl_scan_max(uint32_t v, uint32_t *total) {
  __shared__ uint32_t ws[NT / 64];
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  const uint32_t inc = wave_incl_max(v);
  if (lane == 63) ws[wid] = inc;
  __syncthreads();
  uint32_t wbase = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) {
    const uint32_t t = ws[i];
    if ((uint32_t)i < wid) wbase = wbase > t ? wbase : t;
    tot = tot > t ? tot : t;
  }
  __syncthreads();
  *total = tot;
  return wbase > inc ? wbase : inc;
}

template <int NT>
__device__ __forceinline__ uint32_t block_reduce_sum(uint32_t v) {
  uint32_t tot;
  (void)block_excl_scan_sum<NT>(v, &tot);
  return tot;
}
template <int NT>
__device__ __forceinline__ uint32_t block_reduce_max(uint32_t v) {
  uint32_t tot;
  (void)block_incl_scan_max<NT>(v, &tot);
  return tot;
}

}  // namespace bce
