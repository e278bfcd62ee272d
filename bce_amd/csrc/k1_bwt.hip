// k1_bwt.hip -- K1: rotation sort + BWT on the GPU.  Replaces File::rotate + File::bwt
// (bce.cpp:858-910), i.e. the host libdivsufsort call divbwt(T,T,0,n-1) (bce.cpp:901) and the two
// std::rotate fix-ups around it.
//
// What the reference produces (verified against the oracle): the BWT of ALL CYCLIC
// ROTATIONS of the input (rotations in sorted order, each contributing the byte that precedes it) and
// offset_ = the first index at which a minimal rotation starts.  Equal rotations (periodic inputs)
// have equal preceding bytes, so their relative order does not matter.
//
// Algorithm (MI355X-first, no sentinel, no recursion): prefix doubling on cyclic ranks.
//   round 0: sort positions by their first 4 bytes (32-bit keys, 4 radix passes).
//   round h: the sequence SA[j]-h (j ascending) is already ordered by rank[p+h]; a STABLE sort of it by
//            rank[p] alone therefore orders by (rank[p], rank[p+h]) -- half the radix passes of a
//            pair sort (the Manber-Myers induction written as a stable LSD radix sort).
//   ranks are "index of the group head", so they only ever refine; stop when every group is a
//   singleton or h >= n.
// All passes are coalesced streams over u32 arrays plus two random gathers and one scatter per round.
#include "common.h"
#include "scan_util.h"

namespace bce {

constexpr int K1_T = 256;

struct K1Plan { uint32_t nb, per_block; };
static K1Plan k1_plan(uint32_t n) {
  const uint32_t chunk = 2048;
  uint32_t chunks = (uint32_t)(((uint64_t)n + chunk - 1) / chunk);
  if (!chunks) chunks = 1;
  uint32_t nb = chunks < 1024u ? chunks : 1024u;
  uint32_t cpb = (chunks + nb - 1) / nb;
  nb = (chunks + cpb - 1) / cpb;
  return {nb, cpb * chunk};
}

__global__ __launch_bounds__(K1_T) void k1_init_kernel(const uint8_t *__restrict__ T, uint32_t n,
                                                       uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
  for (uint64_t i = (uint64_t)blockIdx.x * K1_T + threadIdx.x; i < n; i += (uint64_t)gridDim.x * K1_T) {
    uint32_t k = 0;
#pragma unroll
    for (uint32_t b = 0; b < 4; ++b) {
      uint64_t q = i + b;
      if (q >= n) q %= n;
      k = (k << 8) | T[q];
    }
    keys[i] = k;
    vals[i] = (uint32_t)i;
  }
}

// keys[j] = rank[SA[j] - h], vals[j] = SA[j] - h   (indices mod n)
__global__ __launch_bounds__(K1_T) void k1_gather_prev_kernel(const uint32_t *__restrict__ sa,
                                                              const uint32_t *__restrict__ rank, uint32_t n,
                                                              uint32_t h, uint32_t *__restrict__ keys,
                                                              uint32_t *__restrict__ vals) {
  for (uint64_t j = (uint64_t)blockIdx.x * K1_T + threadIdx.x; j < n; j += (uint64_t)gridDim.x * K1_T) {
    const uint32_t q = sa[j];
    const uint32_t p = q >= h ? q - h : q + (n - h);
    vals[j] = p;
    keys[j] = rank[p];
  }
}

// k2[j] = rank[SA[j] + h]
__global__ __launch_bounds__(K1_T) void k1_gather_next_kernel(const uint32_t *__restrict__ sa,
                                                              const uint32_t *__restrict__ rank, uint32_t n,
                                                              uint32_t h, uint32_t *__restrict__ k2) {
  for (uint64_t j = (uint64_t)blockIdx.x * K1_T + threadIdx.x; j < n; j += (uint64_t)gridDim.x * K1_T) {
    const uint64_t q = (uint64_t)sa[j] + h;
    k2[j] = rank[q >= n ? q - n : q];
  }
}

// nrk[j] = j where a new group starts, else 0; per-block max and the global group count
__global__ __launch_bounds__(K1_T) void k1_heads_kernel(const uint32_t *__restrict__ k1,
                                                        const uint32_t *__restrict__ k2, uint32_t n,
                                                        uint32_t per_block, uint32_t *__restrict__ nrk,
                                                        uint32_t *__restrict__ blockmax, uint32_t *__restrict__ groups) {
  const uint64_t beg = (uint64_t)blockIdx.x * per_block;
  uint64_t end = beg + per_block;
  if (end > n) end = n;
  uint32_t mx = 0, cnt = 0;
  for (uint64_t j = beg + threadIdx.x; j < end; j += K1_T) {
    bool head = (j == 0);
    if (!head) {
      head = k1[j] != k1[j - 1];
      if (!head && k2) head = k2[j] != k2[j - 1];
    }
    nrk[j] = head ? (uint32_t)j : 0u;
    if (head) { mx = (uint32_t)j; ++cnt; }
  }
  mx = block_reduce_max<K1_T>(mx);
  cnt = block_reduce_sum<K1_T>(cnt);
  if (threadIdx.x == 0) {
    blockmax[blockIdx.x] = mx;
    if (cnt) atomicAdd(groups, cnt);
  }
}

// rank[SA[j]] = max(nrk[0..j])  (running max == index of the group head)
__global__ __launch_bounds__(K1_T) void k1_apply_kernel(const uint32_t *__restrict__ nrk,
                                                        const uint32_t *__restrict__ blockmax, uint32_t n,
                                                        uint32_t per_block, const uint32_t *__restrict__ sa,
                                                        uint32_t *__restrict__ rank) {
  uint32_t c = 0;
  for (uint32_t b = threadIdx.x; b < blockIdx.x; b += K1_T) { const uint32_t v = blockmax[b]; c = c > v ? c : v; }
  uint32_t carry = block_reduce_max<K1_T>(c);
  const uint64_t beg = (uint64_t)blockIdx.x * per_block;
  uint64_t end = beg + per_block;
  if (end > n) end = n;
  for (uint64_t base = beg; base < end; base += K1_T) {
    const uint64_t j = base + threadIdx.x;
    const bool valid = j < end;
    const uint32_t v = valid ? nrk[j] : 0u;
    uint32_t tot;
    uint32_t inc = block_incl_scan_max<K1_T>(v, &tot);
    inc = inc > carry ? inc : carry;
    if (valid) rank[sa[j]] = inc;
    carry = carry > tot ? carry : tot;
  }
}

__global__ __launch_bounds__(K1_T) void k1_bwt_kernel(const uint8_t *__restrict__ T, const uint32_t *__restrict__ sa,
                                                      uint32_t n, uint8_t *__restrict__ bwt) {
  for (uint64_t j = (uint64_t)blockIdx.x * K1_T + threadIdx.x; j < n; j += (uint64_t)gridDim.x * K1_T) {
    const uint32_t q = sa[j];
    bwt[j] = T[q ? q - 1 : n - 1];
  }
}

// offset_ = smallest position whose rotation is minimal (rank 0): rotate() keeps the FIRST minimal
// index (bce.cpp:873-874)
__global__ __launch_bounds__(K1_T) void k1_offset_kernel(const uint32_t *__restrict__ rank, uint32_t n,
                                                         uint32_t *__restrict__ out) {
  uint32_t best = 0xFFFFFFFFu;
  for (uint64_t p = (uint64_t)blockIdx.x * K1_T + threadIdx.x; p < n; p += (uint64_t)gridDim.x * K1_T)
    if (rank[p] == 0 && (uint32_t)p < best) best = (uint32_t)p;
  // wave min, then one atomic per wave
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const uint32_t t = __shfl_xor(best, o); best = best < t ? best : t; }
  if ((threadIdx.x & 63u) == 0 && best != 0xFFFFFFFFu) atomicMin(out, best);
}

static uint32_t grid_for(uint32_t n) {
  uint64_t b = ((uint64_t)n + K1_T - 1) / K1_T;
  return (uint32_t)(b < 4096 ? (b ? b : 1) : 4096);
}

int k1_bwt(bce_hip_ctx *c) {
  const uint32_t n = c->n;
  uint8_t *T = c->text.as<uint8_t>();
  BCE_TRY(ensure(c, c->bwt, n));
  uint8_t *bwt = c->bwt.as<uint8_t>();
  c->stats.sort_rounds = 0;
  if (n == 1) {
    BCE_HIP_TRY(c, hipMemcpyAsync(bwt, T, 1, hipMemcpyDeviceToDevice, c->stream));
    BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->offset = 0;
    return BCE_HIP_OK;
  }
  const size_t b4 = (size_t)n * 4;
  for (int i = 0; i < 2; ++i) { BCE_TRY(ensure(c, c->sa[i], b4)); BCE_TRY(ensure(c, c->key[i], b4)); }
  BCE_TRY(ensure(c, c->rank, b4));
  BCE_TRY(ensure(c, c->k2, b4));
  BCE_TRY(ensure(c, c->nrk, b4));
  const K1Plan pl = k1_plan(n);
  BCE_TRY(ensure(c, c->blk, (size_t)(pl.nb + 16) * 4));
  uint32_t *blockmax = c->blk.as<uint32_t>();
  uint32_t *scalars = blockmax + pl.nb;  // [0] groups, [1] offset
  uint32_t *rank = c->rank.as<uint32_t>(), *k2 = c->k2.as<uint32_t>(), *nrk = c->nrk.as<uint32_t>();
  const uint32_t g = grid_for(n);

  uint32_t *key[2] = {c->key[0].as<uint32_t>(), c->key[1].as<uint32_t>()};
  uint32_t *val[2] = {c->sa[0].as<uint32_t>(), c->sa[1].as<uint32_t>()};
  hipLaunchKernelGGL(k1_init_kernel, dim3(g), dim3(K1_T), 0, c->stream, T, n, key[0], val[0]);
  int res = 0;
  BCE_TRY(radix_sort_pairs(c, key, val, n, 0, 32, &res));
  uint32_t groups = 0;
  auto rerank = [&](const uint32_t *k1s, const uint32_t *k2s, const uint32_t *sa) -> int {
    BCE_HIP_TRY(c, hipMemsetAsync(scalars, 0, 4, c->stream));
    hipLaunchKernelGGL(k1_heads_kernel, dim3(pl.nb), dim3(K1_T), 0, c->stream, k1s, k2s, n, pl.per_block, nrk,
                       blockmax, scalars);
    hipLaunchKernelGGL(k1_apply_kernel, dim3(pl.nb), dim3(K1_T), 0, c->stream, nrk, blockmax, n, pl.per_block, sa,
                       rank);
    BCE_HIP_TRY(c, hipMemcpyAsync(&groups, scalars, 4, hipMemcpyDeviceToHost, c->stream));
    BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return BCE_HIP_OK;
  };
  BCE_TRY(rerank(key[res], nullptr, val[res]));
  const uint32_t bits = ceil_log2(n);
  uint64_t h = 4;
  while (groups < n && h < n) {
    // input of the sort goes to slot 0 of a local ping-pong so the result index is well defined
    uint32_t *ki[2] = {key[res ^ 1], key[res]};
    uint32_t *vi[2] = {val[res ^ 1], val[res]};
    hipLaunchKernelGGL(k1_gather_prev_kernel, dim3(g), dim3(K1_T), 0, c->stream, val[res], rank, n, (uint32_t)h,
                       ki[0], vi[0]);
    int r2 = 0;
    BCE_TRY(radix_sort_pairs(c, ki, vi, n, 0, bits, &r2, 9));
    uint32_t *sk = ki[r2], *ssa = vi[r2];
    hipLaunchKernelGGL(k1_gather_next_kernel, dim3(g), dim3(K1_T), 0, c->stream, ssa, rank, n, (uint32_t)h, k2);
    BCE_TRY(rerank(sk, k2, ssa));
    // make (key[res], val[res]) name the sorted arrays again
    res = (ssa == val[0]) ? 0 : 1;
    h <<= 1;
    c->stats.sort_rounds++;
  }
  BCE_HIP_TRY(c, hipMemsetAsync(scalars + 1, 0xFF, 4, c->stream));
  hipLaunchKernelGGL(k1_bwt_kernel, dim3(g), dim3(K1_T), 0, c->stream, T, val[res], n, bwt);
  hipLaunchKernelGGL(k1_offset_kernel, dim3(g), dim3(K1_T), 0, c->stream, rank, n, scalars + 1);
  uint32_t off = 0;
  BCE_HIP_TRY(c, hipMemcpyAsync(&off, scalars + 1, 4, hipMemcpyDeviceToHost, c->stream));
  BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
  BCE_HIP_TRY(c, hipGetLastError());
  if (off >= n) { snprintf(c->err, sizeof c->err, "k1: no rank-0 rotation found"); return BCE_HIP_E_INTERNAL; }
  c->offset = off;
  return BCE_HIP_OK;
}

}  // namespace bce
