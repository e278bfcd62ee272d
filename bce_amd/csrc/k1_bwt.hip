// k1_bwt.hip -- K1: rotation sort + BWT on the GPU.  Replaces File::rotate + File::bwt
// (bce.cpp:858-910), i.e. the host libdivsufsort call divbwt(T,T,0,n-1) (bce.cpp:901) and the two
// std::rotate fix-ups around it.
//
// What the reference produces (verified against the oracle): the BWT of ALL CYCLIC
// ROTATIONS of the input (rotations in sorted order, each contributing the byte that precedes it) and
// offset_ = the first index at which a minimal rotation starts.  Equal rotations (periodic inputs)
// have equal preceding bytes, so their relative order does not matter.
//
// Algorithm (MI355X-first, no sentinel, no recursion): prefix doubling on cyclic ranks; ranks are "index of the group head",
// so they only ever refine; stop when every group is a singleton or h >= n.  The sorter is k1_sort_rotations ("v2"):
//   * first sort on a 64-BIT key of as many symbols as fit (alphabet compacted to ceil(log2 sigma) bits: 12 symbols
//     of text, 8 of arbitrary bytes) built in text order -- no gather at all; ranks by one scatter.
//   * every later round works on the ACTIVE list only (elements of non-singleton groups, ascending SA slots, head
//     flags in bit 31) and sorts each group where it lies: a workgroup loads the groups that START in its 2048-element
//     chunk into LDS, gathers rank[p + h] once, rank-sorts each group (all-pairs counting, groups <= 2048), writes the
//     suffixes back to their slots and the new ranks to a staging array -- two random accesses per element and no radix
//     pass.  Groups larger than 2048 are DEFERRED to one wide radix sort on (group number, rank[p + h]).  New ranks are
//     applied by a second kernel: a gather of another workgroup must never see a rank of this round (mixing old and new
//     ranks inside one comparison can order two suffixes wrongly).
// The sorter of rounds 1-2 stays as k1_sort_rotations_v1 behind BCE_K1_V1=1 for A/B runs only: a 4-byte first key (4 radix
// passes), then full rounds in which the sequence SA[j]-h (j ascending) is already ordered by rank[p+h], so a STABLE sort
// of it by rank[p] alone orders by (rank[p], rank[p+h]) (the Manber-Myers induction as a stable LSD radix sort).
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

// ---- rank scatters as sorts (inputs far beyond the caches: 10^9 B) ----
// rank[SA[j]] = v and rank[vs[i]] = nr[i] are one random 4-byte store per element into an array of 4 n bytes: at n = 10^9
// every one misses every cache and moves a line for its four bytes (k1_apply_kernel: 45 ms for one scatter of 4 GB, 23 G
// stores / s; 2.3 ms at 10^8).  Measured first, and rejected: ONE radix pass on the top bits of the destination, then stores
// into 16 MB windows -- the stores ran at the same 26 G / s (the eight XCDs' L2s each hold a part of every line of the
// window and write it back partly filled), and the BWT as a byte scatter bwt[rank[p]] = T[p - 1] into 4 MB windows took 28 ms
// against 19 for the gather.  What does pay is to SORT the pairs by destination (three 9/10-bit passes of the pair sort,
// ~4.5 ms each per 10^9 pairs): the scatter of a permutation becomes the sorted value array itself, that of a subset a
// store stream in ascending order.
#ifndef K1_PART_MIN
#define K1_PART_MIN (1u << 27)                   // elements from which the sorted form is used
#endif

// hv[j] = max(nrk[0..j]) = index of j's group head (what k1_apply_kernel scatters), written in place of order j
__global__ __launch_bounds__(K1_T) void k1_headidx_kernel(const uint32_t *__restrict__ nrk, const uint32_t *__restrict__ blockmax, uint32_t n,
                                                          uint32_t per_block, uint32_t *__restrict__ hv) {
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
    if (valid) hv[j] = inc;
    carry = carry > tot ? carry : tot;
  }
}
// dst[key[i]] = val[i] for the keys below `limit` (pairs grouped by destination range)
__global__ __launch_bounds__(K1_T) void k1_scatter32_kernel(const uint32_t *__restrict__ key, const uint32_t *__restrict__ val, uint32_t m,
                                                            uint32_t limit, uint32_t *__restrict__ dst) {
  for (uint64_t i = (uint64_t)blockIdx.x * K1_T + threadIdx.x; i < m; i += (uint64_t)gridDim.x * K1_T) {
    const uint32_t k = key[i];
    if (k < limit) dst[k] = val[i];
  }
}
// the segmented round's results as pairs: key = the suffix (0xFFFFFFFF for deferred elements: they are applied elsewhere)
__global__ __launch_bounds__(K1_T) void k1_seg_pairs_kernel(const uint32_t *__restrict__ vs, const uint8_t *__restrict__ flag, uint8_t defer, uint32_t m,
                                                            uint32_t *__restrict__ key) {
  for (uint64_t i = (uint64_t)blockIdx.x * K1_T + threadIdx.x; i < m; i += (uint64_t)gridDim.x * K1_T)
    key[i] = (flag[i] & defer) ? 0xFFFFFFFFu : vs[i];
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
                                                         uint32_t *__restrict__ out, uint32_t *__restrict__ total, uint32_t mark = 0,
                                                         const uint32_t *__restrict__ sa_in = nullptr, uint32_t *__restrict__ vout = nullptr) {
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
      // mark: bit 31 = this element starts a group (the v2 rounds find the groups of the list from these bits)
      if (k) {
        out[base + ex] = (src ? src[i] : (uint32_t)i) | ((mark && (i == 0 || nrk[i] != 0)) ? 0x80000000u : 0u);
        if (vout) vout[base + ex] = sa_in[i];               // (v2: the suffix at the slot travels with the list)
      }
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
static int k1_sort_rotations_v1(bce_hip_ctx *c, const uint8_t *T, uint32_t n, bool sentinel) {
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


// ---- v2: 64-bit first key + segmented rounds --------------------------------------------------------------------------
constexpr uint32_t K1_SLOT = 0x7FFFFFFFu, K1_HEAD = 0x80000000u;
constexpr uint32_t K1_SC_HIST = 16, K1_SC_CODE = 16 + 256;     // words behind the scalars: byte histogram, symbol code table (256 x u16:
                                                               // with the sentinel 257 symbols can occur)
constexpr uint8_t KF_HEAD = 1, KF_KEEP = 2, KF_DEFER = 4;

__global__ __launch_bounds__(K1_T) void k1_bytehist_kernel(const uint8_t *__restrict__ T, uint32_t n, uint32_t *__restrict__ hist) {
  __shared__ uint32_t lh[256];
  lh[threadIdx.x] = 0;
  __syncthreads();
  for (uint64_t i = (uint64_t)blockIdx.x * K1_T + threadIdx.x; i < n; i += (uint64_t)gridDim.x * K1_T) atomicAdd(&lh[T[i]], 1u);
  __syncthreads();
  if (lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lh[threadIdx.x]);
}

// key[p] = the first nsym symbols of rotation p, `bits` bits each, first symbol most significant; val[p] = p.
// m = number of rotations (n bytes of T, plus the sentinel when m = n + 1: symbol 0, every byte's code is then >= 1).
constexpr int K1I_E = 8;
__global__ __launch_bounds__(K1_T) void k1_init64_kernel(const uint8_t *__restrict__ T, uint32_t m, uint32_t nbytes,
                                                         const uint16_t *__restrict__ code, uint32_t bits, uint32_t nsym,
                                                         uint32_t *__restrict__ lo, uint32_t *__restrict__ hi, uint32_t *__restrict__ val) {
  __shared__ uint16_t sym[K1_T * K1I_E + 16];
  __shared__ uint16_t lut[256];
  lut[threadIdx.x] = code[threadIdx.x];
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * (K1_T * K1I_E);
  for (uint32_t i = threadIdx.x; i < K1_T * K1I_E + 16; i += K1_T) {
    uint64_t q = base + i;
    if (q >= m) q %= m;
    sym[i] = q < nbytes ? lut[T[q]] : (uint16_t)0;           // (q == nbytes only with the sentinel)
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < K1I_E; ++e) {
    const uint32_t o = (uint32_t)e * K1_T + threadIdx.x;
    const uint64_t p = base + o;
    if (p < m) {
      uint64_t k = 0;
      for (uint32_t b = 0; b < nsym; ++b) k = (k << bits) | sym[o + b];
      lo[p] = (uint32_t)k;
      hi[p] = (uint32_t)(k >> 32);
      val[p] = (uint32_t)p;
    }
  }
}

// ---- segmented rounds ----
constexpr int SEG_T = 1024;                    // threads of a workgroup
constexpr int SEG_CH = 2048;                   // list elements whose groups a workgroup owns (groups that START in its chunk)
constexpr int SEG_CAP = 2048;                  // largest group sorted in LDS; larger ones are deferred
constexpr int SEG_W = SEG_CH + SEG_CAP;        // sortable window: list positions [base, base + W)
constexpr int SEG_HW = SEG_W + SEG_CAP;        // head-flag window: list positions [base - CAP, base + W)

// highest set bit at or below pos (and at or above lo_limit), or -1
__device__ __forceinline__ int seg_prev_bit(const uint64_t *b, int pos, int lo_limit) {
  int wi = pos >> 6;
  uint64_t word = b[wi] & (~0ull >> (63 - (pos & 63)));
  const int wl = lo_limit >> 6;
  while (word == 0 && wi > wl) { --wi; word = b[wi]; }
  if (!word) return -1;
  const int r = wi * 64 + 63 - __clzll((long long)word);
  return r >= lo_limit ? r : -1;
}
// lowest set bit above pos (and at or below hi_limit), or -1
__device__ __forceinline__ int seg_next_bit(const uint64_t *b, int pos, int hi_limit) {
  int wi = pos >> 6;
  uint64_t word = (pos & 63) == 63 ? 0ull : (b[wi] & (~0ull << ((pos & 63) + 1)));
  const int wh = hi_limit >> 6;
  while (word == 0 && wi < wh) { ++wi; word = b[wi]; }
  if (!word) return -1;
  const int r = wi * 64 + __ffsll((long long)word) - 1;
  return r <= hi_limit ? r : -1;
}

// One doubling round on the groups that lie whole in LDS.  A[i] = SA slot of active element i (ascending) | head bit.
// Writes: sa[slot] (the group's suffixes in their new order), nr[i] / vs[i] (new rank = slot of the new group's head, and
// the suffix now at position i: applied by k1_seg_apply_kernel), flag[i] (new head, keep = not a singleton); elements of
// groups larger than SEG_CAP get KF_DEFER and are counted in *ndefer.
__global__ __launch_bounds__(SEG_T) void k1_seg_sort_kernel(const uint32_t *__restrict__ A, const uint32_t *__restrict__ V, uint32_t m, uint32_t *__restrict__ sa,
                                                            const uint32_t *__restrict__ rank, uint32_t n, uint32_t h,
                                                            uint32_t *__restrict__ nr, uint32_t *__restrict__ vs,
                                                            uint8_t *__restrict__ flag, uint32_t *__restrict__ ndefer) {
  __shared__ uint64_t hb[SEG_HW / 64 + 1];
  __shared__ uint64_t nbits[SEG_W / 64 + 1];
  __shared__ __attribute__((aligned(16))) uint32_t K[SEG_W];
  __shared__ uint32_t S[SEG_W], SL[SEG_W];
  __shared__ uint32_t s_defer;
  __shared__ int s_cntw;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const int64_t base = (int64_t)blockIdx.x * SEG_CH;
  if (tid == 0) { s_defer = 0; hb[SEG_HW / 64] = 0; nbits[SEG_W / 64] = 0; }
  // 1. head bits of the window (position m is a head: the end of the list) and the slots of the sortable part
#pragma unroll
  for (int it = 0; it < SEG_HW / SEG_T; ++it) {
    const int u = it * SEG_T + (int)tid;
    const int64_t i = base - SEG_CAP + u;
    uint32_t a = 0;
    bool hd = false;
    if (i >= 0) {
      if (i < (int64_t)m) { a = A[i]; hd = (a & K1_HEAD) != 0u; }
      else hd = true;
    }
    const uint64_t bal = __ballot(hd);
    if (lane == 0) hb[u >> 6] = bal;
    if (u >= SEG_CAP) SL[u - SEG_CAP] = a & K1_SLOT;
  }
  __syncthreads();
  // the groups that start in this chunk end at the first head at or after the next chunk's start: nothing beyond it is
  // this block's business (positions of the chunk itself are always looked at: their big groups are flagged here)
  if (tid == 0) {
    const int e = seg_next_bit(hb, SEG_CAP + SEG_CH - 1, SEG_HW - 1);
    const int cw = (e < 0 ? SEG_HW : e) - SEG_CAP;
    s_cntw = cw > SEG_CH ? cw : SEG_CH;
  }
  __syncthreads();
  const int cntw = s_cntw;
  // 2. every position up to there finds its group; the ones of small groups that start in this chunk gather their key
  constexpr int EPT = SEG_W / SEG_T;
  uint32_t k2[EPT], sfx[EPT];
  int gs[EPT], ge[EPT];
  bool mine[EPT];
  uint32_t ndef = 0;
#pragma unroll
  for (int it = 0; it < EPT; ++it) {
    const int w = it * SEG_T + (int)tid, u = w + SEG_CAP;
    const int64_t i = base + w;
    mine[it] = false; gs[it] = 0; ge[it] = 0; k2[it] = 0; sfx[it] = 0;
    if (w < cntw && i < (int64_t)m) {
      const int g0 = seg_prev_bit(hb, u, u - SEG_CAP + 1);            // a head within the last CAP positions, or the group is big
      bool big = g0 < 0;
      int g1 = -1;
      if (!big) {
        const int lim = g0 + SEG_CAP < SEG_HW - 1 ? g0 + SEG_CAP : SEG_HW - 1;
        g1 = seg_next_bit(hb, u, lim);                              // the next head within CAP of the group's start
        big = g1 < 0;
      }
      if (big) {
        if (w < SEG_CH) { flag[i] = KF_DEFER; ++ndef; }               // (each element is flagged by the block of its own chunk)
      } else if (g0 >= SEG_CAP && g0 < SEG_CAP + SEG_CH) {
        mine[it] = true; gs[it] = g0 - SEG_CAP; ge[it] = g1 - SEG_CAP;
      }
    }
  }
#pragma unroll
  for (int it = 0; it < EPT; ++it) if (mine[it]) sfx[it] = V[base + it * SEG_T + (int)tid];
#pragma unroll
  for (int it = 0; it < EPT; ++it)
    if (mine[it]) {
      uint64_t q = (uint64_t)sfx[it] + h;
      if (q >= n) q -= n;
      k2[it] = rank[q];
    }
#pragma unroll
  for (int it = 0; it < EPT; ++it) if (mine[it]) K[it * SEG_T + (int)tid] = k2[it];
  if (ndef) atomicAdd(&s_defer, ndef);
  __syncthreads();
  // 3. rank sort inside each group: position = number of smaller (key, old position) pairs = the members before w with
  //    key <= mine plus the members after w with key < mine.  Two compare-and-add-carry instructions per member, four
  //    members per LDS read (most of a mid-sized group's time is this loop: 65-2048 members, all pairs).
  int pos[EPT];
#pragma unroll
  for (int it = 0; it < EPT; ++it) {
    pos[it] = 0;
    if (mine[it]) {
      const int w = it * SEG_T + (int)tid;
      const uint32_t kw = k2[it];
      uint32_t r = 0;
      {
        int t = gs[it];
        const int e = w;                                              // [gs, w): <=
        for (; t < e && (t & 3); ++t) r += K[t] <= kw ? 1u : 0u;
        for (; t + 4 <= e; t += 4) {
          const uint4 v = *reinterpret_cast<const uint4 *>(&K[t]);
          r += (v.x <= kw ? 1u : 0u) + (v.y <= kw ? 1u : 0u) + (v.z <= kw ? 1u : 0u) + (v.w <= kw ? 1u : 0u);
        }
        for (; t < e; ++t) r += K[t] <= kw ? 1u : 0u;
      }
      {
        int t = w + 1;
        const int e = ge[it];                                         // (w, ge): <
        for (; t < e && (t & 3); ++t) r += K[t] < kw ? 1u : 0u;
        for (; t + 4 <= e; t += 4) {
          const uint4 v = *reinterpret_cast<const uint4 *>(&K[t]);
          r += (v.x < kw ? 1u : 0u) + (v.y < kw ? 1u : 0u) + (v.z < kw ? 1u : 0u) + (v.w < kw ? 1u : 0u);
        }
        for (; t < e; ++t) r += K[t] < kw ? 1u : 0u;
      }
      pos[it] = gs[it] + (int)r;
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < EPT; ++it) if (mine[it]) { K[pos[it]] = k2[it]; S[pos[it]] = sfx[it]; }
  __syncthreads();
  // 4. new heads: position w of its group's new order
#pragma unroll
  for (int it = 0; it < EPT; ++it) {
    const int w = it * SEG_T + (int)tid;
    if ((w & ~63) < cntw) {                                          // (wave-uniform)
      const bool nh = mine[it] && (w == gs[it] || K[w] != K[w - 1]);
      const uint64_t bal = __ballot(nh);
      if (lane == 0) nbits[w >> 6] = bal;
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < EPT; ++it)
    if (mine[it]) {
      const int w = it * SEG_T + (int)tid;
      const int hp = seg_prev_bit(nbits, w, gs[it]);                 // >= gs: the group's first position is a head
      const bool nh = hp == w;
      const bool next_head = (w + 1 == ge[it]) || ((nbits[(w + 1) >> 6] >> ((w + 1) & 63)) & 1ull);
      const int64_t i = base + w;
      const uint32_t s = S[w];
      sa[SL[w]] = s;
      nr[i] = SL[hp];
      vs[i] = s;
      flag[i] = (uint8_t)((nh ? KF_HEAD : 0) | ((nh && next_head) ? 0 : KF_KEEP));
    }
  if (tid == 0 && s_defer) atomicAdd(ndefer, s_defer);
}

// rank[vs[i]] = nr[i] for the elements the segmented kernel sorted (the deferred ones are applied by the global path)
__global__ __launch_bounds__(K1_T) void k1_seg_apply_kernel(const uint32_t *__restrict__ nr, const uint32_t *__restrict__ vs,
                                                            const uint8_t *__restrict__ flag, uint32_t m, uint32_t *__restrict__ rank) {
  for (uint64_t i = (uint64_t)blockIdx.x * K1_T + threadIdx.x; i < m; i += (uint64_t)gridDim.x * K1_T)
    if (!(flag[i] & KF_DEFER)) rank[vs[i]] = nr[i];
}

// compaction of the list by a flag: PASS 0 counts per block, PASS 1 writes.  MODE 0: deferred elements -> their list index (idx)
// and slot (slots); MODE 1: kept elements -> slot | new head bit and the suffix now at that slot (the next round's lists).
template <int PASS, int MODE>
__global__ __launch_bounds__(K1_T) void k1_relist_kernel(const uint32_t *__restrict__ A, const uint32_t *__restrict__ vs,
                                                         const uint8_t *__restrict__ flag, uint32_t m,
                                                         uint32_t per_block, uint32_t nb, uint32_t *__restrict__ blockcnt,
                                                         uint32_t *__restrict__ out, uint32_t *__restrict__ out2, uint32_t *__restrict__ total,
                                                         uint32_t *__restrict__ blockcnt2 = nullptr, uint32_t *__restrict__ out3 = nullptr) {
  const uint64_t beg = (uint64_t)blockIdx.x * per_block;
  uint64_t end = beg + per_block;
  if (end > m) end = m;
  auto take = [&](uint64_t i) -> bool { return MODE == 0 ? (flag[i] & KF_DEFER) != 0 : (flag[i] & KF_KEEP) != 0; };
  if (PASS == 0) {
    uint32_t cnt = 0, hc = 0;
    for (uint64_t i = beg + threadIdx.x; i < end; i += K1_T) {
      const bool k = take(i);
      cnt += k ? 1u : 0u;
      if (MODE == 0) hc += (k && (A[i] & K1_HEAD)) ? 1u : 0u;
    }
    cnt = block_reduce_sum<K1_T>(cnt);
    if (MODE == 0) hc = block_reduce_sum<K1_T>(hc);
    if (threadIdx.x == 0) { blockcnt[blockIdx.x] = cnt; if (MODE == 0) blockcnt2[blockIdx.x] = hc; }
  } else {
    uint32_t b = 0, a = 0, b2 = 0, a2 = 0;
    for (uint32_t k = threadIdx.x; k < nb; k += K1_T) {
      const uint32_t v = blockcnt[k];
      a += v; if (k < blockIdx.x) b += v;
      if (MODE == 0) { const uint32_t v2 = blockcnt2[k]; a2 += v2; if (k < blockIdx.x) b2 += v2; }
    }
    uint32_t base = block_reduce_sum<K1_T>(b);
    const uint32_t all = block_reduce_sum<K1_T>(a);
    uint32_t hbase = 0;
    if (MODE == 0) {
      hbase = block_reduce_sum<K1_T>(b2);
      const uint32_t all2 = block_reduce_sum<K1_T>(a2);
      if (blockIdx.x == 0 && threadIdx.x == 0) total[1] = all2;     // deferred groups
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) total[0] = all;
    for (uint64_t c0 = beg; c0 < end; c0 += K1_T) {
      const uint64_t i = c0 + threadIdx.x;
      const bool k = i < end && take(i);
      const uint32_t a_i = (MODE == 0 && k) ? A[i] : 0u;
      uint32_t tot;
      const uint32_t ex = block_excl_scan_sum<K1_T>(k ? 1u : 0u, &tot);
      uint32_t htot = 0, hex = 0;
      const uint32_t hd = (MODE == 0 && (a_i & K1_HEAD)) ? 1u : 0u;
      if (MODE == 0) hex = block_excl_scan_sum<K1_T>(hd, &htot);
      if (k) {
        if (MODE == 0) { out[base + ex] = (uint32_t)i; out2[base + ex] = a_i & K1_SLOT; out3[base + ex] = hbase + hex + hd - 1u; }
        else { out[base + ex] = (A[i] & K1_SLOT) | ((flag[i] & KF_HEAD) ? K1_HEAD : 0u); out2[base + ex] = vs[i]; }
      }
      base += tot;
      hbase += htot;
    }
  }
}

// ---- the global path (round 3): the elements of groups too large for LDS, all such groups in ONE sort ----------------
// key = (number of the group among the deferred groups) << rbits | rank[p + h], value = p: one stable sort on the bits in
// use orders every deferred group by its second key where it lies (the list is ascending and holds whole groups, so the
// sorted sequence lines up with the slots).  Against v1's active round: one random gather and one random scatter per
// element instead of seven, five wide passes instead of six narrow ones and a gather.
__global__ __launch_bounds__(K1_T) void k1_big_gather_kernel(const uint32_t *__restrict__ idx, const uint32_t *__restrict__ gid,
                                                             const uint32_t *__restrict__ V, const uint32_t *__restrict__ rank,
                                                             uint32_t n, uint32_t nd, uint32_t h, uint32_t rbits,
                                                             uint32_t *__restrict__ lo, uint32_t *__restrict__ hi, uint32_t *__restrict__ val) {
  for (uint64_t j = (uint64_t)blockIdx.x * K1_T + threadIdx.x; j < nd; j += (uint64_t)gridDim.x * K1_T) {
    const uint32_t p = V[idx[j]];
    uint64_t q = (uint64_t)p + h;
    if (q >= n) q -= n;
    const uint64_t key = ((uint64_t)gid[j] << rbits) | rank[q];
    lo[j] = (uint32_t)key;
    hi[j] = (uint32_t)(key >> 32);
    val[j] = p;
  }
}
// the sorted suffixes go back to their slots; nrk[j] = slot where a new group starts (else 0), per-block maxima for k1_act_apply_kernel
__global__ __launch_bounds__(K1_T) void k1_big_heads_kernel(const uint32_t *__restrict__ slots, const uint32_t *__restrict__ lo,
                                                            const uint32_t *__restrict__ hi, const uint32_t *__restrict__ val,
                                                            uint32_t nd, uint32_t per_block, uint32_t *__restrict__ sa,
                                                            uint32_t *__restrict__ nrk, uint32_t *__restrict__ blockmax) {
  const uint64_t beg = (uint64_t)blockIdx.x * per_block;
  uint64_t end = beg + per_block;
  if (end > nd) end = nd;
  uint32_t mx = 0;
  for (uint64_t j = beg + threadIdx.x; j < end; j += K1_T) {
    const uint32_t slot = slots[j];
    sa[slot] = val[j];
    const bool head = j == 0 || lo[j] != lo[j - 1] || hi[j] != hi[j - 1];
    nrk[j] = head ? slot : 0u;
    if (head) mx = slot;
  }
  mx = block_reduce_max<K1_T>(mx);
  if (threadIdx.x == 0) blockmax[blockIdx.x] = mx;
}

// the global path's result (nrk over the deferred list: slot where a new group starts, else 0) as flags of the list elements
__global__ __launch_bounds__(K1_T) void k1_defer_flags_kernel(const uint32_t *__restrict__ nrk, const uint32_t *__restrict__ idx,
                                                              const uint32_t *__restrict__ sorted, uint32_t nd,
                                                              uint8_t *__restrict__ flag, uint32_t *__restrict__ vs) {
  for (uint64_t j = (uint64_t)blockIdx.x * K1_T + threadIdx.x; j < nd; j += (uint64_t)gridDim.x * K1_T) {
    const bool hd = j == 0 || nrk[j] != 0u;
    const bool nx = j + 1 >= nd || nrk[j + 1] != 0u;
    const uint32_t i = idx[j];
    flag[i] = (uint8_t)(KF_DEFER | (hd ? KF_HEAD : 0) | ((hd && nx) ? 0 : KF_KEEP));
    vs[i] = sorted[j];
  }
}

static int k1_sort_rotations(bce_hip_ctx *c, const uint8_t *T, uint32_t n, bool sentinel) {
  if (getenv("BCE_K1_V1")) return k1_sort_rotations_v1(c, T, n, sentinel);
  const size_t b4 = (size_t)n * 4;
  for (int i = 0; i < 2; ++i) { BCE_TRY(ensure(c, c->sa[i], b4)); BCE_TRY(ensure(c, c->key[i], b4)); BCE_TRY(ensure(c, c->khi[i], b4)); }
  BCE_TRY(ensure(c, c->rank, b4));
  BCE_TRY(ensure(c, c->k2, b4));
  BCE_TRY(ensure(c, c->nrk, b4));
  BCE_TRY(ensure(c, c->kflag, (size_t)n + 16));
  for (int i = 0; i < 2; ++i) { BCE_TRY(ensure(c, c->act[i], b4)); BCE_TRY(ensure(c, c->actv[i], b4)); }
  const K1Plan pl = k1_plan(n);
  BCE_TRY(ensure(c, c->blk, (size_t)(K1_MAXB + K1_SC_CODE + 128 + K1_MAXB) * 4));
  uint32_t *blockmax = c->blk.as<uint32_t>();
  uint32_t *scalars = blockmax + K1_MAXB;  // [0] groups, [1] offset, [2] active elements, [3] deferred elements, [4..5] deferred list;
                                           // histogram; code table; a second per-block array
  uint32_t *rank = c->rank.as<uint32_t>(), *k2 = c->k2.as<uint32_t>(), *nrk = c->nrk.as<uint32_t>();
  uint8_t *flag = c->kflag.as<uint8_t>();
  const uint32_t g = grid_for(n);
  const bool trace = getenv("BCE_K1_TRACE") != nullptr;
  double t_prev = now_s();
  auto lap = [&](const char *what, uint64_t h, uint32_t m, uint32_t nd) {
    if (!trace) return;
    (void)hipStreamSynchronize(c->stream);
    const double t = now_s();
    fprintf(stderr, "k1 %p: %-10s h %llu active %u deferred %u (+%.2f ms)\n", (void *)c, what, (unsigned long long)h, m, nd, (t - t_prev) * 1e3);
    t_prev = t;
  };

  // ---- alphabet: the symbols that occur, in byte order, ceil(log2) bits each ----
  const uint32_t nbytes = sentinel ? n - 1u : n;
  BCE_HIP_TRY(c, hipMemsetAsync(scalars + K1_SC_HIST, 0, 256 * 4, c->stream));
  hipLaunchKernelGGL(k1_bytehist_kernel, dim3(g < 1024u ? g : 1024u), dim3(K1_T), 0, c->stream, T, nbytes, scalars + K1_SC_HIST);
  uint32_t hist[256];
  BCE_TRY(read_back(c, hist, scalars + K1_SC_HIST, sizeof hist));
  uint16_t code[256];
  uint32_t sigma = sentinel ? 1u : 0u;
  for (int b = 0; b < 256; ++b) { code[b] = (uint16_t)sigma; if (hist[b]) ++sigma; }    // (absent bytes: any value, never read)
  uint32_t bits = ceil_log2(sigma);
  if (bits < 1) bits = 1;
  uint32_t nsym = 64u / bits;
  if (nsym > 16u) nsym = 16u;
  // (diagnostic: a narrower first key -- more of the order is left to the segmented and active rounds; measured at 10^8 B,
  //  text / natural corpus / binary corpus: 64 bits 16.9 / 33.3 / 28.2 ms, 54: 17.8 / 36.1 / 29.1, 48: 18.7 / 35.9 / 28.9,
  //  40: 19.5 / 38.1 / 30.7, 32: 23.3 / 41.9 / 32.7 -- every pass of the wide sort earns more than it costs)
  if (const char *e = getenv("BCE_K1_KEYBITS")) { const uint32_t v = (uint32_t)strtoul(e, nullptr, 10); if (v >= bits && v / bits < nsym) nsym = v / bits; }
  const uint32_t keybits = nsym * bits;
  {
    if (!c->h_small) BCE_TRY(pin_alloc(c, &c->h_small, 4096));
    memcpy(c->h_small, code, sizeof code);
    BCE_HIP_TRY(c, hipMemcpyAsync(scalars + K1_SC_CODE, c->h_small, sizeof code, hipMemcpyHostToDevice, c->stream));
  }

  // ---- first sort: 64-bit keys of nsym symbols ----
  uint32_t *lo[2] = {c->key[0].as<uint32_t>(), c->key[1].as<uint32_t>()};
  uint32_t *hi[2] = {c->khi[0].as<uint32_t>(), c->khi[1].as<uint32_t>()};
  uint32_t *val[2] = {c->sa[0].as<uint32_t>(), c->sa[1].as<uint32_t>()};
  {
    const uint32_t per = K1_T * K1I_E;
    hipLaunchKernelGGL(k1_init64_kernel, dim3((uint32_t)(((uint64_t)n + per - 1) / per)), dim3(K1_T), 0, c->stream, T, n, nbytes,
                       reinterpret_cast<const uint16_t *>(scalars + K1_SC_CODE), bits, nsym, lo[0], hi[0], val[0]);
  }
  int res = 0;
  BCE_TRY(radix_sort_wide(c, lo, hi, val, n, keybits, &res, 9));     // (9-bit digits: a 10-bit pass costs 1.0 ms at 10^8, a 9-bit one 0.57)
  lap("first sort", nsym, n, 0);
  // ranks: group heads from the sorted keys, one scatter; the first active list
  BCE_HIP_TRY(c, hipMemsetAsync(scalars, 0, 16, c->stream));
  hipLaunchKernelGGL(k1_heads_kernel, dim3(pl.nb), dim3(K1_T), 0, c->stream, hi[res], lo[res], n, pl.per_block, nrk, blockmax, scalars);
  const uint32_t rbits = ceil_log2(n);
  static const uint32_t part_min = getenv("BCE_K1_PART_MIN") ? (uint32_t)strtoul(getenv("BCE_K1_PART_MIN"), nullptr, 10) : K1_PART_MIN;
  const bool sorted_apply = n >= part_min;
  if (sorted_apply) hipLaunchKernelGGL(k1_headidx_kernel, dim3(pl.nb), dim3(K1_T), 0, c->stream, nrk, blockmax, n, pl.per_block, k2);
  else hipLaunchKernelGGL(k1_apply_kernel, dim3(pl.nb), dim3(K1_T), 0, c->stream, nrk, blockmax, n, pl.per_block, val[res], rank);
  uint32_t *act[2] = {c->act[0].as<uint32_t>(), c->act[1].as<uint32_t>()};
  uint32_t *actv[2] = {c->actv[0].as<uint32_t>(), c->actv[1].as<uint32_t>()};
  uint32_t m = 0;
  hipLaunchKernelGGL(k1_active_kernel<0>, dim3(pl.nb), dim3(K1_T), 0, c->stream, nrk, (const uint32_t *)nullptr, n, pl.per_block, pl.nb,
                     blockmax, act[0], scalars + 2, 1u, (const uint32_t *)val[res], actv[0]);
  hipLaunchKernelGGL(k1_active_kernel<1>, dim3(pl.nb), dim3(K1_T), 0, c->stream, nrk, (const uint32_t *)nullptr, n, pl.per_block, pl.nb,
                     blockmax, act[0], scalars + 2, 1u, (const uint32_t *)val[res], actv[0]);
  if (sorted_apply) {
    // rank = the head indices (k2) sorted by suffix: the first pass reads the suffix array and leaves it alone, the last one
    // writes the rank array itself
    const uint32_t b1 = rbits / ((rbits + 9u) / 10u);                       // the digit of a balanced pass
    uint32_t *pk[2] = {val[res], lo[res ^ 1]}, *pv[2] = {k2, rank};
    int pr = 0;
    BCE_TRY(radix_sort_pairs(c, pk, pv, n, 0, b1, &pr, 10));                // one pass: (lo[res^1], rank)
    uint32_t *qk[2] = {lo[res ^ 1], hi[res ^ 1]}, *qv[2] = {rank, val[res ^ 1]};
    BCE_TRY(radix_sort_pairs(c, qk, qv, n, b1, rbits - b1, &pr, 10));
    if (pr) BCE_HIP_TRY(c, hipMemcpyAsync(rank, qv[1], (size_t)n * 4, hipMemcpyDeviceToDevice, c->stream));   // (an odd number of passes behind the first: n > 2^30)
  }
  BCE_TRY(read_back(c, &m, scalars + 2, 4));
  lap("first ranks", nsym, m, 0);

  // ---- segmented rounds ----
  uint64_t h = nsym;
  uint32_t *sa = val[res];
  uint32_t *nr = hi[0], *vs = hi[1];             // (the high key words are free now)
  while (m > 0 && h < n) {
    const uint32_t *A = act[0];
    const uint32_t nblk = (uint32_t)(((uint64_t)m + SEG_CH - 1) / SEG_CH);
    BCE_HIP_TRY(c, hipMemsetAsync(scalars + 3, 0, 4, c->stream));
    hipLaunchKernelGGL(k1_seg_sort_kernel, dim3(nblk), dim3(SEG_T), 0, c->stream, A, (const uint32_t *)actv[0], m, sa, rank, n, (uint32_t)h, nr, vs,
                       flag, scalars + 3);
    uint32_t nd = 0;
    BCE_TRY(read_back(c, &nd, scalars + 3, 4));
    const K1Plan ap = k1_plan(m);
    if (nd) {
      // ---- the groups that do not fit LDS: one sort of all their elements by (group, second key) ----
      BCE_TRY(ensure(c, c->dl[0], (size_t)nd * 4));
      BCE_TRY(ensure(c, c->dl[1], (size_t)nd * 4));
      BCE_TRY(ensure(c, c->dl[2], (size_t)nd * 4));
      BCE_TRY(ensure(c, c->dl[3], (size_t)nd * 4));
      uint32_t *idx = c->dl[0].as<uint32_t>(), *slots = c->dl[1].as<uint32_t>(), *gid = c->dl[2].as<uint32_t>();
      uint32_t *blockcnt2 = scalars + K1_SC_CODE + 128;
      hipLaunchKernelGGL((k1_relist_kernel<0, 0>), dim3(ap.nb), dim3(K1_T), 0, c->stream, A, (const uint32_t *)vs, flag, m, ap.per_block, ap.nb, blockmax, idx, slots,
                         scalars + 4, blockcnt2, gid);
      hipLaunchKernelGGL((k1_relist_kernel<1, 0>), dim3(ap.nb), dim3(K1_T), 0, c->stream, A, (const uint32_t *)vs, flag, m, ap.per_block, ap.nb, blockmax, idx, slots,
                         scalars + 4, blockcnt2, gid);
      uint32_t cnts[2] = {0, 0};
      BCE_TRY(read_back(c, cnts, scalars + 4, 8));
      if (cnts[0] != nd || cnts[1] == 0 || cnts[1] > nd) { snprintf(c->err, sizeof c->err, "k1: deferred list %u / %u groups, expected %u elements", cnts[0], cnts[1], nd); return BCE_HIP_E_INTERNAL; }
      const uint32_t gbits = ceil_log2(cnts[1]);
      uint32_t *bl[2] = {lo[0], lo[1]};
      uint32_t *bh[2] = {val[res ^ 1], k2};
      uint32_t *bv[2] = {nrk, c->dl[3].as<uint32_t>()};
      const uint32_t ga = grid_for(nd);
      hipLaunchKernelGGL(k1_big_gather_kernel, dim3(ga), dim3(K1_T), 0, c->stream, idx, gid, (const uint32_t *)actv[0], rank, n, nd, (uint32_t)h, rbits,
                         bl[0], bh[0], bv[0]);
      int rb = 0;
      BCE_TRY(radix_sort_wide(c, bl, bh, bv, nd, rbits + gbits, &rb, 9));
      const K1Plan dp = k1_plan(nd);
      uint32_t *snrk = bv[rb ^ 1];                 // (the other value buffer is free after the sort)
      hipLaunchKernelGGL(k1_big_heads_kernel, dim3(dp.nb), dim3(K1_T), 0, c->stream, slots, bl[rb], bh[rb], bv[rb], nd, dp.per_block, sa, snrk, blockmax);
      hipLaunchKernelGGL(k1_act_apply_kernel, dim3(dp.nb), dim3(K1_T), 0, c->stream, snrk, blockmax, nd, dp.per_block, bv[rb], rank);
      uint32_t *vjr = bv[rb];
      hipLaunchKernelGGL(k1_defer_flags_kernel, dim3(ga), dim3(K1_T), 0, c->stream, snrk, idx, (const uint32_t *)vjr, nd, flag, vs);
    }
    if (m >= part_min) {
      // (the deferred groups' buffers -- lo[], k2, the other suffix array -- are free again: their kernels are queued ahead)
      const uint32_t b1 = rbits / ((rbits + 9u) / 10u);
      uint32_t *pk[2] = {lo[0], lo[1]}, *pv[2] = {nr, k2};
      hipLaunchKernelGGL(k1_seg_pairs_kernel, dim3(grid_for(m)), dim3(K1_T), 0, c->stream, (const uint32_t *)vs, (const uint8_t *)flag, KF_DEFER, m, pk[0]);
      int pr = 0;
      BCE_TRY(radix_sort_pairs(c, pk, pv, m, 0, b1, &pr, 10));               // one pass: (lo[1], k2); nr is left alone
      uint32_t *qk[2] = {lo[1], lo[0]}, *qv[2] = {k2, val[res ^ 1]};
      BCE_TRY(radix_sort_pairs(c, qk, qv, m, b1, rbits - b1, &pr, 10));
      hipLaunchKernelGGL(k1_scatter32_kernel, dim3(grid_for(m)), dim3(K1_T), 0, c->stream, (const uint32_t *)qk[pr], (const uint32_t *)qv[pr], m, n, rank);
    } else {
      hipLaunchKernelGGL(k1_seg_apply_kernel, dim3(grid_for(m)), dim3(K1_T), 0, c->stream, nr, vs, flag, m, rank);
    }
    hipLaunchKernelGGL((k1_relist_kernel<0, 1>), dim3(ap.nb), dim3(K1_T), 0, c->stream, A, (const uint32_t *)vs, flag, m, ap.per_block, ap.nb, blockmax, act[1],
                       actv[1], scalars + 2);
    hipLaunchKernelGGL((k1_relist_kernel<1, 1>), dim3(ap.nb), dim3(K1_T), 0, c->stream, A, (const uint32_t *)vs, flag, m, ap.per_block, ap.nb, blockmax, act[1],
                       actv[1], scalars + 2);
    const uint32_t m_old = m;
    BCE_TRY(read_back(c, &m, scalars + 2, 4));
    std::swap(act[0], act[1]);
    std::swap(actv[0], actv[1]);
    h <<= 1;
    c->stats.sort_rounds++;
    lap("round", h, m, nd);
    (void)m_old;
  }
  BCE_HIP_TRY(c, hipGetLastError());
  c->k1_unique = m == 0;
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
