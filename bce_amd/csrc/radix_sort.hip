// radix_sort.hip -- stable LSD radix sort of (u32 key, u32 value) pairs for gfx950.
//
// Used by K1 (prefix-doubling suffix sort) and K4 (grouping symbol records by model slot).
// wave64 design: the stable rank of an element among equal digits is computed with __ballot
// match-any masks (one ballot per digit bit) + popcount of the lower lanes; per-wave digit counters
// live in LDS and are bumped once per peer group (one ds_add_rtn per distinct digit per 64 keys).
// Each block owns ONE contiguous range of the input, so the digit-major histogram matrix is at most
// 512 x 768 and one block per digit row scans it; per pass: histogram, scan, scatter.
// The scatter reorders each 4096-pair chunk by digit in LDS first, so that consecutive lanes write consecutive
// addresses of a digit's output segment (64 B+ runs instead of 4 B scatters).
#include "common.h"
#include "scan_util.h"

namespace bce {

constexpr int RS_THREADS = 256;
constexpr int RS_ITEMS = 16;
constexpr int RS_CHUNK = RS_THREADS * RS_ITEMS;  // 4096 keys per block iteration
constexpr int RS_MAXB = 768;                     // 3 blocks per CU (44 KB of LDS each) x 256 CUs
constexpr int RS_MAXBITS = 10;                   // digit width per pass (<= 1024 bins; K1 uses 9, K4's 19-bit keys 10 + 9)
constexpr int RS_MAXBINS = 1 << RS_MAXBITS;

struct RsPlan { uint32_t nb, per_block; };
static RsPlan rs_plan(uint32_t n) {
  uint32_t chunks = (uint32_t)(((uint64_t)n + RS_CHUNK - 1) / RS_CHUNK);
  if (chunks == 0) chunks = 1;
  uint32_t nb = chunks < (uint32_t)RS_MAXB ? chunks : (uint32_t)RS_MAXB;
  uint32_t cpb = (chunks + nb - 1) / nb;
  nb = (chunks + cpb - 1) / cpb;
  return {nb, cpb * (uint32_t)RS_CHUNK};
}

// lanes holding the same digit as this lane (valid lanes only).  NBITS > 0: digit width known at compile time (the
// scatter kernel is bound by exactly these instructions -- rocprofv3: 108 VALU per key with the runtime loop, SIMDs
// ~60 % busy issuing -- so the loop is unrolled and written as "lanes that DIFFER in some bit": per bit one bitfield
// extract that gives 0 / ~0, one compare for the ballot, two xors, and one three-way or per two bits and half).
template <int NBITS>
__device__ __forceinline__ uint64_t match_digit(uint32_t d, int nbits, bool valid) {
  const uint64_t vm = __ballot(valid);
  if (NBITS > 0) {
    uint32_t dl = 0, dh = 0;                     // lanes whose digit differs from mine in some bit
#pragma unroll
    for (int b = 0; b + 1 < NBITS; b += 2) {
      const uint32_t n0 = (uint32_t)__builtin_amdgcn_sbfe((int)d, b, 1), n1 = (uint32_t)__builtin_amdgcn_sbfe((int)d, b + 1, 1);   // 0 or ~0
      const uint64_t m0 = __ballot(n0 != 0u), m1 = __ballot(n1 != 0u);
      dl |= ((uint32_t)m0 ^ n0) | ((uint32_t)m1 ^ n1);          // bit set: my bit is 1 and theirs 0, or the other way round
      dh |= ((uint32_t)(m0 >> 32) ^ n0) | ((uint32_t)(m1 >> 32) ^ n1);
    }
    if (NBITS & 1) {
      const uint32_t n0 = (uint32_t)__builtin_amdgcn_sbfe((int)d, NBITS - 1, 1);
      const uint64_t m0 = __ballot(n0 != 0u);
      dl |= (uint32_t)m0 ^ n0;
      dh |= (uint32_t)(m0 >> 32) ^ n0;
    }
    // (my bit 1: m ^ ~0 = lanes with 0 = the lanes that differ; my bit 0: m ^ 0 = lanes with 1 = the lanes that differ)
    const uint64_t differ = ((uint64_t)dh << 32) | dl;
    return valid ? (vm & ~differ) : 0ull;
  }
  uint64_t peers = vm;
  for (int b = 0; b < nbits; ++b) {
    const bool bit = (d >> b) & 1u;
    const uint64_t m = __ballot(bit);
    peers &= bit ? m : ~m;
  }
  return valid ? peers : 0ull;
}

// Per-block digit histogram.  Counting needs no ranks: one LDS add per key into the wave's private row (a wave
// whose 64 keys share the digit -- the usual case in the top passes -- adds 64 at once instead of colliding).
__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(const uint32_t *__restrict__ keys, uint32_t n,
                                                             uint32_t per_block, uint32_t nb, int shift, int nbits,
                                                             uint32_t *__restrict__ hist) {
  __shared__ uint32_t lh[4][RS_MAXBINS];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
  const uint32_t nbins = 1u << nbits, mask = nbins - 1u;
  for (uint32_t d = tid; d < nbins; d += RS_THREADS) { lh[0][d] = 0; lh[1][d] = 0; lh[2][d] = 0; lh[3][d] = 0; }
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
      const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
      if (__all(valid && d == d0)) {
        if (lane == 0) atomicAdd(&lh[w][d0], 64u);
      } else if (valid) {
        atomicAdd(&lh[w][d], 1u);
      }
    }
  }
  __syncthreads();
  for (uint32_t d = tid; d < nbins; d += RS_THREADS) hist[(size_t)d * nb + blockIdx.x] = lh[0][d] + lh[1][d] + lh[2][d] + lh[3][d];
}

// One block per digit: exclusive scan of the digit's row hist[d][0..nb) in place (coalesced), row total aside.
// The digit bases (exclusive scan of the row totals) are folded in by the scatter kernel's prologue.
__global__ __launch_bounds__(RS_THREADS) void rs_scan_kernel(uint32_t *__restrict__ hist, uint32_t nb,
                                                             uint32_t *__restrict__ rowtotal) {
  uint32_t *row = hist + (size_t)blockIdx.x * nb;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < nb; base += RS_THREADS) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < nb ? row[i] : 0u;
    uint32_t tot;
    const uint32_t ex = block_excl_scan_sum<RS_THREADS>(v, &tot);
    if (i < nb) row[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) rowtotal[blockIdx.x] = carry;
}

template <int NBITS>
__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(const uint32_t *__restrict__ keys_in,
                                                                const uint32_t *__restrict__ vals_in,
                                                                uint32_t *__restrict__ keys_out,
                                                                uint32_t *__restrict__ vals_out, uint32_t n,
                                                                uint32_t per_block, uint32_t nb, int shift, int nbits,
                                                                const uint32_t *__restrict__ hist,
                                                                const uint32_t *__restrict__ rowtotal) {
  constexpr int BINS = NBITS > 0 ? (1 << NBITS) : RS_MAXBINS;   // (LDS for the digits this instantiation can see: 3 blocks per CU up to 9 bits)
  __shared__ uint32_t wcnt[4][BINS];           // per-wave digit counts of the chunk, then per-wave local bases
  __shared__ uint32_t goff[BINS];              // this block's next output position per digit
  __shared__ uint32_t gdelta[BINS];            // output position - position in the chunk's LDS order (mod 2^32)
  __shared__ uint32_t skey[RS_CHUNK], sval[RS_CHUNK];
  constexpr int BPT = (BINS + RS_THREADS - 1) / RS_THREADS;  // digits per thread in the per-digit steps
  const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
  const uint32_t nbins = 1u << nbits, mask = nbins - 1u;
  // digit bases: exclusive scan of the row totals, plus this block's row offset
  {
    uint32_t v[BPT], sum = 0;
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
      const uint32_t d = tid * BPT + q;
      v[q] = d < nbins ? rowtotal[d] : 0u;
      sum += v[q];
    }
    uint32_t tot;
    uint32_t run = block_excl_scan_sum<RS_THREADS>(sum, &tot);
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
      const uint32_t d = tid * BPT + q;
      if (d < nbins) goff[d] = run + hist[(size_t)d * nb + blockIdx.x];
      run += v[q];
    }
  }
  const uint64_t beg = (uint64_t)blockIdx.x * per_block;
  uint64_t end = beg + per_block;
  if (end > n) end = n;
  const uint64_t lt = (1ull << lane) - 1ull;
  for (uint64_t base = beg; base < end; base += RS_CHUNK) {
    const uint32_t cnt = (uint32_t)(end - base < (uint64_t)RS_CHUNK ? end - base : (uint64_t)RS_CHUNK);
    for (uint32_t d = tid; d < nbins; d += RS_THREADS) { wcnt[0][d] = 0; wcnt[1][d] = 0; wcnt[2][d] = 0; wcnt[3][d] = 0; }
    __syncthreads();
    uint32_t key[RS_ITEMS], val[RS_ITEMS], lr[RS_ITEMS];
    // wave w owns the contiguous quarter [w*1024, (w+1)*1024) of the chunk: order = (wave, step, lane)
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
      const uint32_t q = w * (RS_CHUNK / 4) + (uint32_t)it * 64u + lane;
      const bool valid = q < cnt;
      key[it] = valid ? keys_in[base + q] : 0u;
      val[it] = valid ? vals_in[base + q] : 0u;
    }
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
      const uint32_t q = w * (RS_CHUNK / 4) + (uint32_t)it * 64u + lane;
      const bool valid = q < cnt;
      const uint32_t d = (key[it] >> shift) & mask;
      const uint64_t peers = match_digit<NBITS>(d, nbits, valid);
      const uint32_t leader = valid ? (uint32_t)(__ffsll((long long)peers) - 1) : lane;
      uint32_t pre = 0;
      if (valid && lane == leader) pre = atomicAdd(&wcnt[w][d], (uint32_t)__popcll(peers));
      pre = __shfl(pre, (int)leader);
      lr[it] = pre + (uint32_t)__popcll(peers & lt);
    }
    __syncthreads();
    {
      // chunk-local digit bases (exclusive scan over the digits), per-wave bases, output deltas
      uint32_t c[BPT][4], sum = 0;
#pragma unroll
      for (int q = 0; q < BPT; ++q) {
        const uint32_t d = tid * BPT + q;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) { c[q][ww] = d < nbins ? wcnt[ww][d] : 0u; sum += c[q][ww]; }
      }
      uint32_t tot;
      uint32_t run = block_excl_scan_sum<RS_THREADS>(sum, &tot);
#pragma unroll
      for (int q = 0; q < BPT; ++q) {
        const uint32_t d = tid * BPT + q;
        if (d < nbins) {
          const uint32_t t4 = c[q][0] + c[q][1] + c[q][2] + c[q][3];
          wcnt[0][d] = run; wcnt[1][d] = run + c[q][0]; wcnt[2][d] = run + c[q][0] + c[q][1];
          wcnt[3][d] = run + c[q][0] + c[q][1] + c[q][2];
          const uint32_t g = goff[d];
          gdelta[d] = g - run;
          goff[d] = g + t4;
          run += t4;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
      const uint32_t q = w * (RS_CHUNK / 4) + (uint32_t)it * 64u + lane;
      if (q < cnt) {
        const uint32_t d = (key[it] >> shift) & mask;
        const uint32_t pos = wcnt[w][d] + lr[it];
        skey[pos] = key[it];
        sval[pos] = val[it];
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
      const uint32_t q = (uint32_t)it * RS_THREADS + tid;
      if (q < cnt) {
        const uint32_t k = skey[q];
        const uint32_t pos = gdelta[(k >> shift) & mask] + q;
        keys_out[pos] = k;
        vals_out[pos] = sval[q];
      }
    }
    __syncthreads();
  }
}

// ---- 64-bit keys (K1's first sort: up to 12 symbols per key) ---------------------------------------------------------
// The key is two u32 arrays (lo, hi), the value a third; a digit is (hi:lo >> shift) & mask and may straddle the words.
// Same three kernels per pass; the scatter moves three words per element through LDS.  ITEMS is smaller than the pair
// sort's (chunk of 2048) so that four blocks still fit a CU.
#ifndef RW_ITEMS_VALUE
#define RW_ITEMS_VALUE 8
#endif
constexpr int RW_ITEMS = RW_ITEMS_VALUE;
constexpr int RW_CHUNK = RS_THREADS * RW_ITEMS;
static RsPlan rw_plan(uint32_t n) {
  uint32_t chunks = (uint32_t)(((uint64_t)n + RW_CHUNK - 1) / RW_CHUNK);
  if (chunks == 0) chunks = 1;
  const uint32_t maxb = RW_ITEMS <= 8 ? 1024 : 768;         // 4 (3) per CU
  uint32_t nb = chunks < maxb ? chunks : maxb;
  uint32_t cpb = (chunks + nb - 1) / nb;
  nb = (chunks + cpb - 1) / cpb;
  return {nb, cpb * (uint32_t)RW_CHUNK};
}
__device__ __forceinline__ uint32_t digit64(uint32_t lo, uint32_t hi, int shift, uint32_t mask) {
  return (uint32_t)((((uint64_t)hi << 32) | lo) >> shift) & mask;
}

__global__ __launch_bounds__(RS_THREADS) void rw_hist_kernel(const uint32_t *__restrict__ lo, const uint32_t *__restrict__ hi,
                                                             uint32_t n, uint32_t per_block, uint32_t nb, int shift, int nbits,
                                                             uint32_t *__restrict__ hist) {
  __shared__ uint32_t lh[4][RS_MAXBINS];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
  const uint32_t nbins = 1u << nbits, mask = nbins - 1u;
  for (uint32_t d = tid; d < nbins; d += RS_THREADS) { lh[0][d] = 0; lh[1][d] = 0; lh[2][d] = 0; lh[3][d] = 0; }
  __syncthreads();
  const uint64_t beg = (uint64_t)blockIdx.x * per_block;
  uint64_t end = beg + per_block;
  if (end > n) end = n;
  const bool need_lo = shift < 32, need_hi = shift + nbits > 32;       // (uniform: a pass reads only the words its digit touches)
  for (uint64_t base = beg; base < end; base += (uint64_t)RS_THREADS * 16) {
    uint32_t kl[16], kh[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const uint64_t i = base + (uint64_t)it * RS_THREADS + tid;
      kl[it] = (need_lo && i < end) ? lo[i] : 0u;
      kh[it] = (need_hi && i < end) ? hi[i] : 0u;
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const uint64_t i = base + (uint64_t)it * RS_THREADS + tid;
      const bool valid = i < end;
      const uint32_t d = digit64(kl[it], kh[it], shift, mask);
      const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
      if (__all(valid && d == d0)) {
        if (lane == 0) atomicAdd(&lh[w][d0], 64u);
      } else if (valid) {
        atomicAdd(&lh[w][d], 1u);
      }
    }
  }
  __syncthreads();
  for (uint32_t d = tid; d < nbins; d += RS_THREADS) hist[(size_t)d * nb + blockIdx.x] = lh[0][d] + lh[1][d] + lh[2][d] + lh[3][d];
}

template <int NBITS>
__global__ __launch_bounds__(RS_THREADS) void rw_scatter_kernel(const uint32_t *__restrict__ lo_in, const uint32_t *__restrict__ hi_in,
                                                                const uint32_t *__restrict__ vals_in,
                                                                uint32_t *__restrict__ lo_out, uint32_t *__restrict__ hi_out,
                                                                uint32_t *__restrict__ vals_out, uint32_t n,
                                                                uint32_t per_block, uint32_t nb, int shift, int nbits,
                                                                const uint32_t *__restrict__ hist,
                                                                const uint32_t *__restrict__ rowtotal) {
  constexpr int BINS = NBITS > 0 ? (1 << NBITS) : RS_MAXBINS;
  __shared__ uint32_t wcnt[4][BINS];
  __shared__ uint32_t goff[BINS];
  __shared__ uint32_t gdelta[BINS];
  __shared__ uint32_t slo[RW_CHUNK], shi[RW_CHUNK], sval[RW_CHUNK];
  constexpr int BPT = (BINS + RS_THREADS - 1) / RS_THREADS;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
  const uint32_t nbins = 1u << nbits, mask = nbins - 1u;
  {
    uint32_t v[BPT], sum = 0;
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
      const uint32_t d = tid * BPT + q;
      v[q] = d < nbins ? rowtotal[d] : 0u;
      sum += v[q];
    }
    uint32_t tot;
    uint32_t run = block_excl_scan_sum<RS_THREADS>(sum, &tot);
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
      const uint32_t d = tid * BPT + q;
      if (d < nbins) goff[d] = run + hist[(size_t)d * nb + blockIdx.x];
      run += v[q];
    }
  }
  const uint64_t beg = (uint64_t)blockIdx.x * per_block;
  uint64_t end = beg + per_block;
  if (end > n) end = n;
  const uint64_t lt = (1ull << lane) - 1ull;
  for (uint64_t base = beg; base < end; base += RW_CHUNK) {
    const uint32_t cnt = (uint32_t)(end - base < (uint64_t)RW_CHUNK ? end - base : (uint64_t)RW_CHUNK);
    for (uint32_t d = tid; d < nbins; d += RS_THREADS) { wcnt[0][d] = 0; wcnt[1][d] = 0; wcnt[2][d] = 0; wcnt[3][d] = 0; }
    __syncthreads();
    uint32_t kl[RW_ITEMS], kh[RW_ITEMS], val[RW_ITEMS], lr[RW_ITEMS];
#pragma unroll
    for (int it = 0; it < RW_ITEMS; ++it) {
      const uint32_t q = w * (RW_CHUNK / 4) + (uint32_t)it * 64u + lane;
      const bool valid = q < cnt;
      kl[it] = valid ? lo_in[base + q] : 0u;
      kh[it] = valid ? hi_in[base + q] : 0u;
      val[it] = valid ? vals_in[base + q] : 0u;
    }
#pragma unroll
    for (int it = 0; it < RW_ITEMS; ++it) {
      const uint32_t q = w * (RW_CHUNK / 4) + (uint32_t)it * 64u + lane;
      const bool valid = q < cnt;
      const uint32_t d = digit64(kl[it], kh[it], shift, mask);
      const uint64_t peers = match_digit<NBITS>(d, nbits, valid);
      const uint32_t leader = valid ? (uint32_t)(__ffsll((long long)peers) - 1) : lane;
      uint32_t pre = 0;
      if (valid && lane == leader) pre = atomicAdd(&wcnt[w][d], (uint32_t)__popcll(peers));
      pre = __shfl(pre, (int)leader);
      lr[it] = pre + (uint32_t)__popcll(peers & lt);
    }
    __syncthreads();
    {
      uint32_t c[BPT][4], sum = 0;
#pragma unroll
      for (int q = 0; q < BPT; ++q) {
        const uint32_t d = tid * BPT + q;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) { c[q][ww] = d < nbins ? wcnt[ww][d] : 0u; sum += c[q][ww]; }
      }
      uint32_t tot;
      uint32_t run = block_excl_scan_sum<RS_THREADS>(sum, &tot);
#pragma unroll
      for (int q = 0; q < BPT; ++q) {
        const uint32_t d = tid * BPT + q;
        if (d < nbins) {
          const uint32_t t4 = c[q][0] + c[q][1] + c[q][2] + c[q][3];
          wcnt[0][d] = run; wcnt[1][d] = run + c[q][0]; wcnt[2][d] = run + c[q][0] + c[q][1];
          wcnt[3][d] = run + c[q][0] + c[q][1] + c[q][2];
          const uint32_t g = goff[d];
          gdelta[d] = g - run;
          goff[d] = g + t4;
          run += t4;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < RW_ITEMS; ++it) {
      const uint32_t q = w * (RW_CHUNK / 4) + (uint32_t)it * 64u + lane;
      if (q < cnt) {
        const uint32_t d = digit64(kl[it], kh[it], shift, mask);
        const uint32_t pos = wcnt[w][d] + lr[it];
        slo[pos] = kl[it];
        shi[pos] = kh[it];
        sval[pos] = val[it];
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < RW_ITEMS; ++it) {
      const uint32_t q = (uint32_t)it * RS_THREADS + tid;
      if (q < cnt) {
        const uint32_t l = slo[q], hh = shi[q];
        const uint32_t pos = gdelta[digit64(l, hh, shift, mask)] + q;
        lo_out[pos] = l;
        hi_out[pos] = hh;
        vals_out[pos] = sval[q];
      }
    }
    __syncthreads();
  }
}

// Stable LSD sort of (64-bit key = hi:lo, u32 value) triples on key bits [0, bits), bits <= 64.  Result in lo[res] / hi[res] / val[res].
int radix_sort_wide(bce_hip_ctx *c, uint32_t *lo[2], uint32_t *hi[2], uint32_t *val[2], uint32_t n, uint32_t bits, int *res,
                    uint32_t max_digit_bits) {
  *res = 0;
  if (n <= 1 || bits == 0) return BCE_HIP_OK;
  if (bits > 64) return BCE_HIP_E_ARG;
  if (max_digit_bits < 1 || max_digit_bits > (uint32_t)RS_MAXBITS) max_digit_bits = 8;
  const RsPlan pl = rw_plan(n);
  BCE_TRY(ensure(c, c->rs_hist, ((size_t)RS_MAXBINS * pl.nb + RS_MAXBINS) * sizeof(uint32_t)));
  uint32_t *hist = c->rs_hist.as<uint32_t>();
  uint32_t *rowtotal = hist + (size_t)RS_MAXBINS * pl.nb;
  const uint32_t npass = (bits + max_digit_bits - 1) / max_digit_bits;
  int cur = 0;
  uint32_t done = 0;
  for (uint32_t pass = 0; pass < npass; ++pass) {
    const uint32_t left = bits - done, pleft = npass - pass;
    const int nbits = (int)((left + pleft - 1) / pleft);
    const uint32_t nbins = 1u << nbits;
    hipLaunchKernelGGL(rw_hist_kernel, dim3(pl.nb), dim3(RS_THREADS), 0, c->stream, lo[cur], hi[cur], n, pl.per_block, pl.nb,
                       (int)done, nbits, hist);
    hipLaunchKernelGGL(rs_scan_kernel, dim3(nbins), dim3(RS_THREADS), 0, c->stream, hist, pl.nb, rowtotal);
#define RW_SCATTER(NB)                                                                                                      \
  hipLaunchKernelGGL(rw_scatter_kernel<NB>, dim3(pl.nb), dim3(RS_THREADS), 0, c->stream, lo[cur], hi[cur], val[cur], lo[cur ^ 1], \
                     hi[cur ^ 1], val[cur ^ 1], n, pl.per_block, pl.nb, (int)done, nbits, hist, rowtotal)
    switch (nbits) {
      case 10: RW_SCATTER(10); break;
      case 9: RW_SCATTER(9); break;
      case 8: RW_SCATTER(8); break;
      case 7: RW_SCATTER(7); break;
      default: RW_SCATTER(0); break;
    }
#undef RW_SCATTER
    cur ^= 1;
    done += (uint32_t)nbits;
  }
  BCE_HIP_TRY(c, hipGetLastError());
  *res = cur;
  return BCE_HIP_OK;
}

int radix_sort_pairs(bce_hip_ctx *c, uint32_t *key[2], uint32_t *val[2], uint32_t n, uint32_t first_bit, uint32_t bits,
                     int *res, uint32_t max_digit_bits) {
  return radix_sort_pairs_on(c, c->stream, c->rs_hist, key, val, n, first_bit, bits, res, max_digit_bits);
}

// The same on a stream and with a histogram buffer of the caller's (K4's flushes sort on their own stream while K3's
// tail sorts its tagged symbols on the main one).
int radix_sort_pairs_on(bce_hip_ctx *c, hipStream_t stream, DevBuf &hbuf, uint32_t *key[2], uint32_t *val[2], uint32_t n,
                        uint32_t first_bit, uint32_t bits, int *res, uint32_t max_digit_bits) {
  *res = 0;
  if (n <= 1 || bits == 0) return BCE_HIP_OK;
  if (max_digit_bits < 1 || max_digit_bits > (uint32_t)RS_MAXBITS) max_digit_bits = 8;
  const RsPlan pl = rs_plan(n);
  BCE_TRY(ensure(c, hbuf, ((size_t)RS_MAXBINS * pl.nb + RS_MAXBINS) * sizeof(uint32_t)));
  uint32_t *hist = hbuf.as<uint32_t>();
  uint32_t *rowtotal = hist + (size_t)RS_MAXBINS * pl.nb;
  // balanced digits: as few passes as the digit width allows, all of (almost) equal width
  const uint32_t npass = (bits + max_digit_bits - 1) / max_digit_bits;
  int cur = 0;
  uint32_t done = 0;
  for (uint32_t pass = 0; pass < npass; ++pass) {
    const uint32_t left = bits - done, pleft = npass - pass;
    const int nbits = (int)((left + pleft - 1) / pleft);
    const uint32_t shift = first_bit + done;
    const uint32_t nbins = 1u << nbits;
    hipLaunchKernelGGL(rs_hist_kernel, dim3(pl.nb), dim3(RS_THREADS), 0, stream, key[cur], n, pl.per_block, pl.nb,
                       (int)shift, nbits, hist);
    hipLaunchKernelGGL(rs_scan_kernel, dim3(nbins), dim3(RS_THREADS), 0, stream, hist, pl.nb, rowtotal);
#define RS_SCATTER(NB)                                                                                              \
  hipLaunchKernelGGL(rs_scatter_kernel<NB>, dim3(pl.nb), dim3(RS_THREADS), 0, stream, key[cur], val[cur], key[cur ^ 1], \
                     val[cur ^ 1], n, pl.per_block, pl.nb, (int)shift, nbits, hist, rowtotal)
    switch (nbits) {
      case 10: RS_SCATTER(10); break;
      case 9: RS_SCATTER(9); break;
      case 8: RS_SCATTER(8); break;
      case 7: RS_SCATTER(7); break;
      case 6: RS_SCATTER(6); break;
      default: RS_SCATTER(0); break;
    }
#undef RS_SCATTER
    cur ^= 1;
    done += (uint32_t)nbits;
  }
  BCE_HIP_TRY(c, hipGetLastError());
  *res = cur;
  return BCE_HIP_OK;
}

}  // namespace bce
