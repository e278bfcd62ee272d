// radix_sort.hip -- stable LSD radix sort of (u32 key, u32 value) pairs for gfx950.
//
// Used by K1 (prefix-doubling suffix sort) and K4 (grouping symbol records by model slot).
// wave64 design: the stable rank of an element among equal digits is computed with __ballot
// match-any masks (one ballot per digit bit) + popcount of the lower lanes; per-wave digit counters
// live in LDS and are bumped once per peer group (one ds_add_rtn per distinct digit per 64 keys).
// Each block owns ONE contiguous range of the input, so the digit-major histogram matrix is at most
// 256 x 1024 and a single-block scan suffices; per pass: histogram, scan, scatter.
#include "common.h"
#include "scan_util.h"

namespace bce {

constexpr int RS_THREADS = 256;
constexpr int RS_ITEMS = 8;
constexpr int RS_CHUNK = RS_THREADS * RS_ITEMS;  // 2048 keys per block iteration
constexpr int RS_MAXB = 1024;

struct RsPlan { uint32_t nb, per_block; };
static RsPlan rs_plan(uint32_t n) {
  uint32_t chunks = (uint32_t)(((uint64_t)n + RS_CHUNK - 1) / RS_CHUNK);
  if (chunks == 0) chunks = 1;
  uint32_t nb = chunks < (uint32_t)RS_MAXB ? chunks : (uint32_t)RS_MAXB;
  uint32_t cpb = (chunks + nb - 1) / nb;
  nb = (chunks + cpb - 1) / cpb;
  return {nb, cpb * (uint32_t)RS_CHUNK};
}

// lanes holding the same digit as this lane (valid lanes only)
__device__ __forceinline__ uint64_t match_digit(uint32_t d, int nbits, bool valid) {
  uint64_t peers = __ballot(valid);
  for (int b = 0; b < nbits; ++b) {
    const bool bit = (d >> b) & 1u;
    const uint64_t m = __ballot(bit);
    peers &= bit ? m : ~m;
  }
  return valid ? peers : 0ull;
}

__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(const uint32_t *__restrict__ keys, uint32_t n,
                                                             uint32_t per_block, uint32_t nb, int shift, int nbits,
                                                             uint32_t *__restrict__ hist) {
  __shared__ uint32_t lh[256];
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t mask = (1u << nbits) - 1u;
  lh[tid] = 0;
  __syncthreads();
  const uint64_t beg = (uint64_t)blockIdx.x * per_block;
  uint64_t end = beg + per_block;
  if (end > n) end = n;
  for (uint64_t base = beg; base < end; base += RS_CHUNK) {
    uint32_t k[RS_ITEMS];
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
      const uint64_t i = base + (uint64_t)it * RS_THREADS + tid;
      k[it] = i < end ? keys[i] : 0u;
    }
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
      const uint64_t i = base + (uint64_t)it * RS_THREADS + tid;
      const bool valid = i < end;
      const uint32_t d = (k[it] >> shift) & mask;
      const uint64_t peers = match_digit(d, nbits, valid);
      if (valid && lane == (uint32_t)(__ffsll((long long)peers) - 1)) atomicAdd(&lh[d], (uint32_t)__popcll(peers));
    }
  }
  __syncthreads();
  if (tid <= mask) hist[(size_t)tid * nb + blockIdx.x] = lh[tid];
}

// exclusive scan of hist[0..total) in place (digit-major, block-minor order); one block of 1024
__global__ __launch_bounds__(1024) void rs_scan_kernel(uint32_t *__restrict__ hist, uint32_t total) {
  const uint32_t tid = threadIdx.x;
  const uint32_t per = (total + 1023u) / 1024u;
  const uint32_t b = tid * per;
  uint32_t e = b + per;
  if (e > total) e = total;
  uint32_t s = 0;
  for (uint32_t i = b; i < e; ++i) s += hist[i];
  uint32_t tot;
  uint32_t run = block_excl_scan_sum<1024>(s, &tot);
  for (uint32_t i = b; i < e; ++i) {
    const uint32_t v = hist[i];
    hist[i] = run;
    run += v;
  }
}

__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(const uint32_t *__restrict__ keys_in,
                                                                const uint32_t *__restrict__ vals_in,
                                                                uint32_t *__restrict__ keys_out,
                                                                uint32_t *__restrict__ vals_out, uint32_t n,
                                                                uint32_t per_block, uint32_t nb, int shift, int nbits,
                                                                const uint32_t *__restrict__ hist) {
  __shared__ uint32_t wcnt[4][256];
  __shared__ uint32_t goff[256];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
  const uint32_t mask = (1u << nbits) - 1u;
  if (tid <= mask) goff[tid] = hist[(size_t)tid * nb + blockIdx.x];
  const uint64_t beg = (uint64_t)blockIdx.x * per_block;
  uint64_t end = beg + per_block;
  if (end > n) end = n;
  const uint64_t lt = (1ull << lane) - 1ull;
  for (uint64_t base = beg; base < end; base += RS_CHUNK) {
    wcnt[0][tid] = 0; wcnt[1][tid] = 0; wcnt[2][tid] = 0; wcnt[3][tid] = 0;
    __syncthreads();
    uint32_t key[RS_ITEMS], val[RS_ITEMS], lr[RS_ITEMS];
    // wave w owns the contiguous quarter [base + w*512, base + (w+1)*512): order = (wave, step, lane)
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
      const uint64_t i = base + (uint64_t)w * (RS_CHUNK / 4) + (uint64_t)it * 64 + lane;
      const bool valid = i < end;
      key[it] = valid ? keys_in[i] : 0u;
      val[it] = valid ? vals_in[i] : 0u;
    }
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
      const uint64_t i = base + (uint64_t)w * (RS_CHUNK / 4) + (uint64_t)it * 64 + lane;
      const bool valid = i < end;
      const uint32_t d = (key[it] >> shift) & mask;
      const uint64_t peers = match_digit(d, nbits, valid);
      const uint32_t leader = valid ? (uint32_t)(__ffsll((long long)peers) - 1) : lane;
      uint32_t pre = 0;
      if (valid && lane == leader) pre = atomicAdd(&wcnt[w][d], (uint32_t)__popcll(peers));
      pre = __shfl(pre, (int)leader);
      lr[it] = pre + (uint32_t)__popcll(peers & lt);
    }
    __syncthreads();
    if (tid <= mask) {
      const uint32_t c0 = wcnt[0][tid], c1 = wcnt[1][tid], c2 = wcnt[2][tid], c3 = wcnt[3][tid];
      const uint32_t b = goff[tid];
      wcnt[0][tid] = b; wcnt[1][tid] = b + c0; wcnt[2][tid] = b + c0 + c1; wcnt[3][tid] = b + c0 + c1 + c2;
      goff[tid] = b + c0 + c1 + c2 + c3;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
      const uint64_t i = base + (uint64_t)w * (RS_CHUNK / 4) + (uint64_t)it * 64 + lane;
      if (i < end) {
        const uint32_t d = (key[it] >> shift) & mask;
        const uint32_t pos = wcnt[w][d] + lr[it];
        keys_out[pos] = key[it];
        vals_out[pos] = val[it];
      }
    }
    __syncthreads();
  }
}

int radix_sort_pairs(bce_hip_ctx *c, uint32_t *key[2], uint32_t *val[2], uint32_t n, uint32_t first_bit, uint32_t bits, int *res) {
  *res = 0;
  if (n <= 1 || bits == 0) return BCE_HIP_OK;
  const RsPlan pl = rs_plan(n);
  BCE_TRY(ensure(c, c->rs_hist, (size_t)256 * pl.nb * sizeof(uint32_t)));
  uint32_t *hist = c->rs_hist.as<uint32_t>();
  int cur = 0;
  for (uint32_t done = 0; done < bits; done += 8) {
    const uint32_t shift = first_bit + done;
    const int nbits = (int)((bits - done) < 8 ? (bits - done) : 8);
    const uint32_t nbins = 1u << nbits;
    hipLaunchKernelGGL(rs_hist_kernel, dim3(pl.nb), dim3(RS_THREADS), 0, c->stream, key[cur], n, pl.per_block, pl.nb,
                       (int)shift, nbits, hist);
    hipLaunchKernelGGL(rs_scan_kernel, dim3(1), dim3(1024), 0, c->stream, hist, nbins * pl.nb);
    hipLaunchKernelGGL(rs_scatter_kernel, dim3(pl.nb), dim3(RS_THREADS), 0, c->stream, key[cur], val[cur],
                       key[cur ^ 1], val[cur ^ 1], n, pl.per_block, pl.nb, (int)shift, nbits, hist);
    cur ^= 1;
  }
  BCE_HIP_TRY(c, hipGetLastError());
  *res = cur;
  return BCE_HIP_OK;
}

}  // namespace bce
