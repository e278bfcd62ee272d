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
#include <stdlib.h>

#include <utility>

#include "common.h"
#include "scan_util.h"

namespace bce {

#ifndef K1_ACT_TENTHS
#define K1_ACT_TENTHS 6   // switch to the active-set rounds below this many tenths of n (4, 6, 8 measured: 6 is best on repetitive text)
#endif

constexpr int K1_T = 256;

// Blocks of the per-block passes (heads / apply / active list): at most K1_MAXB, each a whole number of 2048-element chunks.
// (nb is NOT monotonic in n -- 10^8 elements give 1018 blocks, 4*10^7 give 1022 -- so everything laid out behind the
//  per-block array sits at K1_MAXB, not at the first plan's nb: with the scalars at nb(n) the active rounds' block 1020
//  shared its count with the list length that block 0 writes at the end of the same kernel, which showed as soon as
//  other contexts delayed the last blocks of a launch.)
constexpr uint32_t K1_MAXB = 1024;
struct K1Plan { uint32_t nb, per_block; };
static K1Plan k1_plan(uint32_t n) {
  const uint32_t chunk = 2048;
  uint32_t chunks = (uint32_t)(((uint64_t)n + chunk - 1) / chunk);
  if (!chunks) chunks = 1;
  uint32_t nb = chunks < K1_MAXB ? chunks : K1_MAXB;
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

// The same for T$ with a unique smallest sentinel (the libdivsufsort seam, k1_divbwt): m = n + 1 rotations, 9-bit symbols
// (0 = $, byte + 1 otherwise), three of them per key.  Rotations of T$ compare like the suffixes of T$.
__global__ __launch_bounds__(K1_T) void k1_init_sentinel_kernel(const uint8_t *__restrict__ T, uint32_t m,
                                                                uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
  for (uint64_t i = (uint64_t)blockIdx.x * K1_T + threadIdx.x; i < m; i += (uint64_t)gridDim.x * K1_T) {
    uint32_t k = 0;
#pragma unroll
    for (uint32_t b = 0; b < 3; ++b) {
      uint64_t q = i + b;
      if (q >= m) q -= m;
      k = (k << 9) | (q == m - 1u ? 0u : (uint32_t)T[q] + 1u);
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

// ---- active-set rounds -----------------------------------------------------------------------------
// Once most suffixes sit in singleton groups, only the rest (the ACTIVE list A of SA indices, ascending) is
// worked on: sort the active suffixes by rank[p+h], then stably by their group (rank[p]), write them back into
// their SA slots (groups are contiguous in A as in SA), re-rank inside the groups, drop new singletons.

// count (pass 0) / write (pass 1) the elements that stay active.  head(i) = i == 0 || nrk[i] != 0;
// an element is a singleton iff it is a head and its successor is a head (or it is the last one).
template <int PASS>
__global__ __launch_bounds__(K1_T) void k1_active_kernel(const uint32_t *__restrict__ nrk,
                                                         const uint32_t *__restrict__ src, uint32_t m,
                                                         uint32_t per_block, uint32_t nb, uint32_t *__restrict__ blockcnt,
                                                         uint32_t *__restrict__ out, uint32_t *__restrict__ total) {
  const uint64_t beg = (uint64_t)blockIdx.x * per_block;
  uint64_t end = beg + per_block;
  if (end > m) end = m;
  auto keep = [&](uint64_t i) -> bool {
    const bool h0 = i == 0 || nrk[i] != 0;
    const bool h1 = i + 1 >= m || nrk[i + 1] != 0;
    return !(h0 && h1);
  };
  if (PASS == 0) {
    uint32_t c = 0;
    for (uint64_t i = beg + threadIdx.x; i < end; i += K1_T) c += keep(i) ? 1u : 0u;
    c = block_reduce_sum<K1_T>(c);
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = c;
  } else {
    uint32_t b = 0, a = 0;
    for (uint32_t k = threadIdx.x; k < nb; k += K1_T) { const uint32_t v = blockcnt[k]; a += v; if (k < blockIdx.x) b += v; }
    uint32_t base = block_reduce_sum<K1_T>(b);
    const uint32_t all = block_reduce_sum<K1_T>(a);
    if (blockIdx.x == 0 && threadIdx.x == 0) *total = all;
    for (uint64_t c0 = beg; c0 < end; c0 += K1_T) {
      const uint64_t i = c0 + threadIdx.x;
      const bool k = i < end && keep(i);
      uint32_t tot;
      const uint32_t ex = block_excl_scan_sum<K1_T>(k ? 1u : 0u, &tot);
      if (k) out[base + ex] = src ? src[i] : (uint32_t)i;
      base += tot;
    }
  }
}

// keys[i] = rank[SA[A[i]] + h], vals[i] = SA[A[i]]
__global__ __launch_bounds__(K1_T) void k1_act_gather_kernel(const uint32_t *__restrict__ A, const uint32_t *__restrict__ sa,
                                                             const uint32_t *__restrict__ rank, uint32_t n, uint32_t m,
                                                             uint32_t h, uint32_t *__restrict__ keys,
                                                             uint32_t *__restrict__ vals) {
  for (uint64_t i = (uint64_t)blockIdx.x * K1_T + threadIdx.x; i < m; i += (uint64_t)gridDim.x * K1_T) {
    const uint32_t p = sa[A[i]];
    const uint64_t q = (uint64_t)p + h;
    vals[i] = p;
    keys[i] = rank[q >= n ? q - n : q];
  }
}
// keys[i] = rank[vals[i]]  (the group of each active suffix)
__global__ __launch_bounds__(K1_T) void k1_act_group_kernel(const uint32_t *__restrict__ vals,
                                                            const uint32_t *__restrict__ rank, uint32_t m,
                                                            uint32_t *__restrict__ keys) {
  for (uint64_t i = (uint64_t)blockIdx.x * K1_T + threadIdx.x; i < m; i += (uint64_t)gridDim.x * K1_T)
    keys[i] = rank[vals[i]];
}
// write the sorted suffixes back to their SA slots; nrk[i] = A[i] where a new group starts (else 0);
// k2 = rank[p+h] is gathered again here (the second sort carried only the group key)
__global__ __launch_bounds__(K1_T) void k1_act_heads_kernel(const uint32_t *__restrict__ A, const uint32_t *__restrict__ vals,
                                                            const uint32_t *__restrict__ k1s,
                                                            const uint32_t *__restrict__ rank, uint32_t n, uint32_t m,
                                                            uint32_t h, uint32_t per_block, uint32_t *__restrict__ sa,
                                                            uint32_t *__restrict__ nrk, uint32_t *__restrict__ blockmax) {
  const uint64_t beg = (uint64_t)blockIdx.x * per_block;
  uint64_t end = beg + per_block;
  if (end > m) end = m;
  uint32_t mx = 0;
  auto key2 = [&](uint64_t i) -> uint32_t { const uint64_t q = (uint64_t)vals[i] + h; return rank[q >= n ? q - n : q]; };
  for (uint64_t i = beg + threadIdx.x; i < end; i += K1_T) {
    const uint32_t slot = A[i];
    sa[slot] = vals[i];
    bool head = i == 0 || k1s[i] != k1s[i - 1];
    if (!head) head = key2(i) != key2(i - 1);
    nrk[i] = head ? slot : 0u;
    if (head) mx = slot;
  }
  mx = block_reduce_max<K1_T>(mx);
  if (threadIdx.x == 0) blockmax[blockIdx.x] = mx;
}
// rank[vals[i]] = running max of nrk (= SA index of the group head)
__global__ __launch_bounds__(K1_T) void k1_act_apply_kernel(const uint32_t *__restrict__ nrk,
                                                            const uint32_t *__restrict__ blockmax, uint32_t m,
                                                            uint32_t per_block, const uint32_t *__restrict__ vals,
                                                            uint32_t *__restrict__ rank) {
  uint32_t c = 0;
  for (uint32_t b = threadIdx.x; b < blockIdx.x; b += K1_T) { const uint32_t v = blockmax[b]; c = c > v ? c : v; }
  uint32_t carry = block_reduce_max<K1_T>(c);
  const uint64_t beg = (uint64_t)blockIdx.x * per_block;
  uint64_t end = beg + per_block;
  if (end > m) end = m;
  for (uint64_t base = beg; base < end; base += K1_T) {
    const uint64_t i = base + threadIdx.x;
    const bool valid = i < end;
    const uint32_t v = valid ? nrk[i] : 0u;
    uint32_t tot;
    uint32_t inc = block_incl_scan_max<K1_T>(v, &tot);
    inc = inc > carry ? inc : carry;
    if (valid) rank[vals[i]] = inc;
    carry = carry > tot ? carry : tot;
  }
}

static uint32_t grid_for(uint32_t n) {
  uint64_t b = ((uint64_t)n + K1_T - 1) / K1_T;
  return (uint32_t)(b < 4096 ? (b ? b : 1) : 4096);
}

// Sort the n cyclic rotations of T (sentinel = false), or the n = |T| + 1 rotations of T$ (sentinel = true: T holds n - 1
// bytes).  Leaves the order in c->sa[c->sa_res] and its inverse (the rank of the group head for tied rotations) in c->rank.
static int k1_sort_rotations(bce_hip_ctx *c, const uint8_t *T, uint32_t n, bool sentinel) {
  const size_t b4 = (size_t)n * 4;
  for (int i = 0; i < 2; ++i) { BCE_TRY(ensure(c, c->sa[i], b4)); BCE_TRY(ensure(c, c->key[i], b4)); }
  BCE_TRY(ensure(c, c->rank, b4));
  BCE_TRY(ensure(c, c->k2, b4));
  BCE_TRY(ensure(c, c->nrk, b4));
  for (int i = 0; i < 2; ++i) BCE_TRY(ensure(c, c->act[i], b4));
  const K1Plan pl = k1_plan(n);
  BCE_TRY(ensure(c, c->blk, (size_t)(K1_MAXB + 16) * 4));
  uint32_t *blockmax = c->blk.as<uint32_t>();
  uint32_t *scalars = blockmax + K1_MAXB;  // [0] groups, [1] offset, [2] active elements
  uint32_t *rank = c->rank.as<uint32_t>(), *k2 = c->k2.as<uint32_t>(), *nrk = c->nrk.as<uint32_t>();
  const uint32_t g = grid_for(n);

  uint32_t *key[2] = {c->key[0].as<uint32_t>(), c->key[1].as<uint32_t>()};
  uint32_t *val[2] = {c->sa[0].as<uint32_t>(), c->sa[1].as<uint32_t>()};
  int res = 0;
  if (sentinel) {
    hipLaunchKernelGGL(k1_init_sentinel_kernel, dim3(g), dim3(K1_T), 0, c->stream, T, n, key[0], val[0]);
    BCE_TRY(radix_sort_pairs(c, key, val, n, 0, 27, &res, 9));
  } else {
    hipLaunchKernelGGL(k1_init_kernel, dim3(g), dim3(K1_T), 0, c->stream, T, n, key[0], val[0]);
    BCE_TRY(radix_sort_pairs(c, key, val, n, 0, 32, &res));
  }
  uint32_t groups = 0;
  auto rerank = [&](const uint32_t *k1s, const uint32_t *k2s, const uint32_t *sa) -> int {
    BCE_HIP_TRY(c, hipMemsetAsync(scalars, 0, 4, c->stream));
    hipLaunchKernelGGL(k1_heads_kernel, dim3(pl.nb), dim3(K1_T), 0, c->stream, k1s, k2s, n, pl.per_block, nrk,
                       blockmax, scalars);
    hipLaunchKernelGGL(k1_apply_kernel, dim3(pl.nb), dim3(K1_T), 0, c->stream, nrk, blockmax, n, pl.per_block, sa,
                       rank);
    BCE_TRY(read_back(c, &groups, scalars, 4));
    return BCE_HIP_OK;
  };
  BCE_TRY(rerank(key[res], nullptr, val[res]));
  const uint32_t bits = ceil_log2(n);
  const uint32_t dg = 9;   // digit width of the rank sorts (7..9 measured equal)
  uint64_t h = sentinel ? 3 : 4;
  uint32_t *act[2] = {c->act[0].as<uint32_t>(), c->act[1].as<uint32_t>()};
  uint32_t m = n;                 // active elements; the list is implicit (identity) while every element is active
  bool have_list = false;
  // active elements after a re-rank: not (head && next is head)
  auto build_active = [&](const uint32_t *nrk_in, const uint32_t *src, uint32_t cnt, uint32_t *out) -> int {
    const K1Plan ap = k1_plan(cnt);
    hipLaunchKernelGGL(k1_active_kernel<0>, dim3(ap.nb), dim3(K1_T), 0, c->stream, nrk_in, src, cnt, ap.per_block, ap.nb,
                       blockmax, out, scalars + 2);
    hipLaunchKernelGGL(k1_active_kernel<1>, dim3(ap.nb), dim3(K1_T), 0, c->stream, nrk_in, src, cnt, ap.per_block, ap.nb,
                       blockmax, out, scalars + 2);
    BCE_TRY(read_back(c, &m, scalars + 2, 4));
    return BCE_HIP_OK;
  };
  const bool trace = getenv("BCE_K1_TRACE") != nullptr;
  double t_prev = now_s();
  if (trace) fprintf(stderr, "k1 %p: start n %u groups %u\n", (void *)c, n, groups);
  while (groups < n && h < n) {
    if (trace) { const double t = now_s(); fprintf(stderr, "k1 %p: h %llu groups %u m %u list %d (+%.2f ms)\n", (void *)c, (unsigned long long)h, groups, m, (int)have_list, (t - t_prev) * 1e3); t_prev = t; }
    if (!have_list && (uint64_t)(n - groups) * 10 < (uint64_t)n * K1_ACT_TENTHS + 10) {
      // few elements can still be in non-singleton groups (at most 2 per missing group... bound: n - groups < 0.4 n
      // means at most 0.8 n active): build the explicit list and check its real size
      BCE_TRY(build_active(nrk, nullptr, n, act[0]));
      have_list = true;
      if (m == 0) break;
    }
    if (have_list && (uint64_t)m * 10 < (uint64_t)n * K1_ACT_TENTHS) {
      // ---- active round on m elements ----
      const uint32_t *A = act[0];
      uint32_t *ki[2] = {key[0], key[1]};
      uint32_t *vi[2] = {val[res ^ 1], k2};
      const uint32_t ga = grid_for(m);
      hipLaunchKernelGGL(k1_act_gather_kernel, dim3(ga), dim3(K1_T), 0, c->stream, A, val[res], rank, n, m, (uint32_t)h,
                         ki[0], vi[0]);
      int r1 = 0;
      BCE_TRY(radix_sort_pairs(c, ki, vi, m, 0, bits, &r1, dg));
      // second sort: by group, input = output of the first
      uint32_t *kj[2] = {ki[r1 ^ 1], ki[r1]};
      uint32_t *vj[2] = {vi[r1], vi[r1 ^ 1]};
      hipLaunchKernelGGL(k1_act_group_kernel, dim3(ga), dim3(K1_T), 0, c->stream, vj[0], rank, m, kj[0]);
      int r2 = 0;
      BCE_TRY(radix_sort_pairs(c, kj, vj, m, 0, bits, &r2, dg));
      const K1Plan ap = k1_plan(m);
      hipLaunchKernelGGL(k1_act_heads_kernel, dim3(ap.nb), dim3(K1_T), 0, c->stream, A, vj[r2], kj[r2], rank, n, m,
                         (uint32_t)h, ap.per_block, val[res], nrk, blockmax);
      hipLaunchKernelGGL(k1_act_apply_kernel, dim3(ap.nb), dim3(K1_T), 0, c->stream, nrk, blockmax, m, ap.per_block,
                         vj[r2], rank);
      BCE_TRY(build_active(nrk, A, m, act[1]));
      std::swap(act[0], act[1]);
      h <<= 1;
      c->stats.sort_rounds++;
      if (m == 0) break;
      continue;
    }
    // ---- full round on all n elements ----
    // input of the sort goes to slot 0 of a local ping-pong so the result index is well defined
    uint32_t *ki[2] = {key[res ^ 1], key[res]};
    uint32_t *vi[2] = {val[res ^ 1], val[res]};
    hipLaunchKernelGGL(k1_gather_prev_kernel, dim3(g), dim3(K1_T), 0, c->stream, val[res], rank, n, (uint32_t)h,
                       ki[0], vi[0]);
    int r2 = 0;
    BCE_TRY(radix_sort_pairs(c, ki, vi, n, 0, bits, &r2, dg));
    uint32_t *sk = ki[r2], *ssa = vi[r2];
    hipLaunchKernelGGL(k1_gather_next_kernel, dim3(g), dim3(K1_T), 0, c->stream, ssa, rank, n, (uint32_t)h, k2);
    BCE_TRY(rerank(sk, k2, ssa));
    // make (key[res], val[res]) name the sorted arrays again
    res = (ssa == val[0]) ? 0 : 1;
    h <<= 1;
    c->stats.sort_rounds++;
    have_list = false;
  }
  // all rotations distinct <=> rank[] is the inverse of the suffix array (used by the depth-first tail of K3)
  c->k1_unique = (groups >= n) || (have_list && m == 0);
  c->sa_res = res;
  return BCE_HIP_OK;
}

int k1_bwt(bce_hip_ctx *c) {
  const uint32_t n = c->n;
  uint8_t *T = c->text.as<uint8_t>();
  BCE_TRY(ensure(c, c->bwt, n));
  uint8_t *bwt = c->bwt.as<uint8_t>();
  c->stats.sort_rounds = 0;
  c->k1_unique = false;
  c->k1_valid = false;
  if (n == 1) {
    BCE_HIP_TRY(c, hipMemcpyAsync(bwt, T, 1, hipMemcpyDeviceToDevice, c->stream));
    BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->offset = 0;
    return BCE_HIP_OK;
  }
  BCE_TRY(k1_sort_rotations(c, T, n, false));
  const uint32_t g = grid_for(n);
  uint32_t *scalars = c->blk.as<uint32_t>() + K1_MAXB;   // [0] groups, [1] offset
  const uint32_t *sa = c->sa[c->sa_res].as<uint32_t>();
  BCE_HIP_TRY(c, hipMemsetAsync(scalars + 1, 0xFF, 4, c->stream));
  hipLaunchKernelGGL(k1_bwt_kernel, dim3(g), dim3(K1_T), 0, c->stream, T, sa, n, bwt);
  hipLaunchKernelGGL(k1_offset_kernel, dim3(g), dim3(K1_T), 0, c->stream, c->rank.as<uint32_t>(), n, scalars + 1);
  uint32_t off = 0;
  BCE_TRY(read_back(c, &off, scalars + 1, 4));
  BCE_HIP_TRY(c, hipGetLastError());
  if (off >= n) { snprintf(c->err, sizeof c->err, "k1: no rank-0 rotation found"); return BCE_HIP_E_INTERNAL; }
  c->offset = off;
  c->k1_valid = true;
  return BCE_HIP_OK;
}

// ---- the libdivsufsort seam (include/divsufsort_hip.h; bce.cpp:901) -------------------------------------------------------
// divbwt(T, U, A, n): the BWT of T[0, n) with an implicit smallest sentinel -- U[0] = T[n-1], then T[SA[i] - 1] for the
// suffixes in sorted order, the one with SA[i] = 0 left out; returns its 1-based position (the primary index).  Here: the
// rotation sort above on T$ (unique sentinel: rotations compare like suffixes), then one gather.
__global__ __launch_bounds__(K1_T) void k1_divbwt_gather_kernel(const uint8_t *__restrict__ T, const uint32_t *__restrict__ sa,
                                                                uint32_t m, uint32_t pidx, uint8_t *__restrict__ U) {
  for (uint64_t j = (uint64_t)blockIdx.x * K1_T + threadIdx.x; j < m; j += (uint64_t)gridDim.x * K1_T) {
    const uint32_t p = sa[j];                    // sa[0] = m - 1: the rotation that starts with $
    if (p != 0u) U[j < pidx ? j : j - 1u] = T[p - 1u];
  }
}

int k1_divbwt(bce_hip_ctx *c, const uint8_t *T_host, uint8_t *U_host, uint32_t n, uint32_t *pidx_out) {
  if (n == 0 || n >= 0x7FFFFFFFu) return BCE_HIP_E_ARG;
  const uint32_t m = n + 1u;
  BCE_TRY(ensure(c, c->text, m));
  BCE_TRY(ensure(c, c->bwt, m));
  uint8_t *T = c->text.as<uint8_t>();
  BCE_HIP_TRY(c, hipMemcpyAsync(T, T_host, n, hipMemcpyHostToDevice, c->stream));
  c->stage = 0; c->k1_valid = false; c->enum_active = false;          // the context's compression state is gone
  c->stats.sort_rounds = 0;
  BCE_TRY(k1_sort_rotations(c, T, m, true));
  uint32_t pidx = 0;
  BCE_TRY(read_back(c, &pidx, c->rank.as<uint32_t>(), 4));   // row of suffix 0
  if (pidx == 0 || pidx > n) { snprintf(c->err, sizeof c->err, "divbwt: primary index %u out of range", pidx); return BCE_HIP_E_INTERNAL; }
  hipLaunchKernelGGL(k1_divbwt_gather_kernel, dim3(grid_for(m)), dim3(K1_T), 0, c->stream, T, c->sa[c->sa_res].as<uint32_t>(), m,
                     pidx, c->bwt.as<uint8_t>());
  BCE_HIP_TRY(c, hipMemcpyAsync(U_host, c->bwt.p, n, hipMemcpyDeviceToHost, c->stream));
  BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
  BCE_HIP_TRY(c, hipGetLastError());
  *pidx_out = pidx;
  return BCE_HIP_OK;
}

}  // namespace bce
