// kd_decode.hip -- GPU-assisted decoder (`bce -d`), SURVEY section 8f "next #1".
//
// BCE::decode (bce.cpp:1169-1233) walks the same 8 tries as the encoder (BCE::code mode 0, bce.cpp:1236-1374),
// but the split of every interval comes out of the plane's range decoder, whose state depends on every symbol
// before it: one strictly sequential stream per plane.  What is NOT sequential is everything around it, and
// that is where the reference spends its time (8n nodes, 3 rank look-ups each; then 8 cursor reads per byte for
// unbwt and one cache miss per byte for the inverse BWT).  So, per round:
//   GPU  (1) classify every node from the dense boundary-rank arrays R[p][x] = rank1_p(x) (known at every
//            boundary learnt so far); nodes whose split is forced need no symbol; the others become QUERIES
//            (k, c1, c2, cs) in stream order (count / scan / write, the K3 pattern),
//   host (2) the eight plane decoders answer their queries in order (8 threads: AdaptiveCoder::get,
//            bce.cpp:555-590; the adaptive counters live here because each answer feeds the next),
//   GPU  (3) children + the new boundary rank R[s + x0] from the answers (count / scan / write).
// After the last round: gap fill (between two known boundaries a plane is constant) -> rank granules ->
// wavelet-matrix access = unbwt::bytewise (bce.cpp:1043-1085) -> LF mapping by one radix pass -> inverse BWT by
// 2^18..2^19 concurrent walkers that stop at marked rows (segments are chained on the host, then written in
// place, rotated by `offset`, bce.cpp:1091-1093).
// The archive format, the model and the coder arithmetic are the reference's; parity = decode(reference
// archive) == input (tests/test_gpu_decode.py).  Inputs whose LF mapping is not one cycle (periodic inputs, on which
// the reference's decoder fails) are unrolled from the cycle through row 0.
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"
#include "decoder_core.h"
#include "k3_args.h"
#include "scan_util.h"

namespace bce {

namespace {

struct DecCtl {
  uint32_t cnt[2][8][2];     // [parity][plane][seg] node counts, as EnumCtl
  uint32_t ptot[8][4];       // per plane totals of the scan in flight: child0, child1, queries, escape queries
  uint32_t qbase[8];         // first query of each plane in this round's query buffer
  uint32_t ebase[8];         // first escape record (k > 31) of each plane
  uint32_t ticket, err;      // err: 1 inconsistent archive, 2 node list overflow
  uint32_t next_nodes, pad;
  uint64_t nodes_total;
};
static_assert(sizeof(DecCtl) <= 4096, "read_back() moves it through 4 KB of pinned memory");

struct DecInfo {             // what the host needs after the query pass (pinned host memory)
  uint32_t qbase[8], qtot[8];
  uint32_t ebase[8], etot[8];
  uint32_t cur_nodes, err;
  uint64_t nodes_total;
};

constexpr uint32_t kEscape = 0x80000000u;

struct DecArgs {
  DecCtl *ctl;
  DecInfo *info;             // device pointer of the pinned DecInfo
  Node *nodes;               // [2][8][capP]
  uint32_t *R;               // [8][n + 1] boundary ranks, kUnknown where not learnt yet
  uint32_t *tilecnt, *tileoff;   // [tiles][4]: 0 child0, 1 child1, 2 queries, 3 escape queries
  uint32_t *Q;               // queries of the round: k | ctx << 5 with the context resolved (k <= 31), or kEscape
  uint4 *E;                  // escape queries (k > 31) in the same order: (k, c1, c2, cs)
  const PlaneCfg *cfg;       // [8] the archive's context-bit tables (the preamble of each stream)
  const uint32_t *res;       // answers, same indexing
  uint32_t capP, n, par;
  uint32_t zeros[8];         // zeros of plane p = C[(p+1)&7]: the child1 lists of plane p start there
  // Mailbox of the wave tail kernel (pinned, host-coherent; nullptr = leave at every query round):
  //   [0] device -> host: number of the query round whose queries are in Q / E / info
  //   [1] host -> device: number of the query round whose answers are in res
  //   [2] device -> host: set to 1 when the kernel has left
  uint32_t *mbox;
  uint32_t seq_base;         // number of the first query round of this launch (numbers never repeat within a decode)
  unsigned long long *words; // dec_small_kernel: one tagged count word per tile
  uint32_t round;            // ... and the round number the tags are made of
  // Six-launch rounds taken in two parts (the busiest plane first, see decompress_device): the planes this launch works
  // on, the order in which the planes' queries lie in Q (4 bits each, first plane lowest), whether this children pass is
  // the round's last (it alone adds up the round)
  uint32_t pmask = 0xFFu, order = 0x76543210u, final = 1u;
};

__device__ __forceinline__ Node *dec_nodes(const DecArgs &a, uint32_t par, uint32_t p) {
  return a.nodes + ((size_t)(par * 8u + p)) * a.capP;
}

__device__ __forceinline__ void dec_tile_prefix(const DecArgs &a, uint32_t tp[9]) {
  uint32_t acc = 0;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    tp[p] = acc;
    const uint32_t m = a.ctl->cnt[a.par][p][0] + a.ctl->cnt[a.par][p][1];
    acc += (m + K3_TILE - 1) / K3_TILE;
  }
  tp[8] = acc;
}

// Exclusive ranks of up to three flag sets over the nodes of a tile in list order (item, wave, lane), and the
// tile totals.  Two barriers; `lds` may be reused afterwards.
__device__ __forceinline__ void tile_ranks(const uint32_t (&f0)[K3_NPT], const uint32_t (&f1)[K3_NPT],
                                           const uint32_t (&f2)[K3_NPT], uint32_t (*lds)[4][3], uint32_t (&r0)[K3_NPT],
                                           uint32_t (&r1)[K3_NPT], uint32_t (&r2)[K3_NPT], uint32_t (&tot)[3]) {
  const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int it = 0; it < K3_NPT; ++it) {
    const uint64_t b0 = __ballot(f0[it]), b1 = __ballot(f1[it]), b2 = __ballot(f2[it]);
    r0[it] = (uint32_t)__popcll(b0 & lt);
    r1[it] = (uint32_t)__popcll(b1 & lt);
    r2[it] = (uint32_t)__popcll(b2 & lt);
    if (lane == 0) {
      lds[it][w][0] = (uint32_t)__popcll(b0);
      lds[it][w][1] = (uint32_t)__popcll(b1);
      lds[it][w][2] = (uint32_t)__popcll(b2);
    }
  }
  __syncthreads();
  uint32_t run0 = 0, run1 = 0, run2 = 0;
#pragma unroll
  for (int it = 0; it < K3_NPT; ++it) {
    uint32_t b0 = run0, b1 = run1, b2 = run2;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) {
      const uint32_t x0 = lds[it][ww][0], x1 = lds[it][ww][1], x2 = lds[it][ww][2];
      if ((uint32_t)ww < w) { b0 += x0; b1 += x1; b2 += x2; }
      run0 += x0; run1 += x1; run2 += x2;
    }
    r0[it] += b0; r1[it] += b1; r2[it] += b2;
  }
  tot[0] = run0; tot[1] = run1; tot[2] = run2;
  __syncthreads();
}

// One node of BCE::code, decode mode (bce.cpp:1261-1351), split in two: what the ranks alone say ...
struct DCls { uint32_t s1, n1x, kind, mn, mx, bad; };   // kind 0: all zeros, 1: all ones, 2: forced split, 3: coded split
__device__ __forceinline__ DCls dec_classify(const Node &nd, const uint32_t *__restrict__ R, uint32_t n) {
  DCls c;
  const uint32_t x = nd.x0 + nd.x1;
  c.bad = ((uint64_t)nd.s + x > n || nd.x0 == 0 || nd.x1 == 0) ? 1u : 0u;
  const uint32_t s1 = c.bad ? 0u : R[nd.s], e1 = c.bad ? 0u : R[nd.s + x];
  if (s1 == kUnknown || e1 == kUnknown || e1 < s1 || e1 - s1 > x) c.bad = 1u;
  c.s1 = s1;
  c.n1x = c.bad ? 0u : e1 - s1;
  c.mn = c.mx = 0;
  if (c.n1x == 0) c.kind = 0;                                    // :1274-1279
  else if (c.n1x == x) c.kind = 1;                               // :1282-1287
  else {
    uint32_t mn = nd.x0 - c.n1x, mx = c.n1x - nd.x1;             // :1290-1294
    mn = ((int32_t)mn < 0) ? 0u : mn;
    mx = ((int32_t)mx < 0) ? 0u : mx;
    c.mn = mn;
    c.mx = nd.x0 - mx;
    c.kind = c.mx != c.mn ? 3u : 2u;
  }
  return c;
}
// ... and the children once the split n0x0 is known (:1337-1350).  Returns the value for R[s + x0].
__device__ __forceinline__ uint32_t dec_children(const Node &nd, const DCls &c, uint32_t n0x0, uint32_t zi, uint32_t &has0,
                                                 Node &c0, uint32_t &has1, Node &c1) {
  const uint32_t x = nd.x0 + nd.x1, s0 = nd.s - c.s1;
  has0 = has1 = 0;
  c0 = Node{0, 0, 0}; c1 = Node{0, 0, 0};
  if (c.kind == 0) { has0 = 1; c0 = Node{s0, nd.x0, nd.x1}; return c.s1; }
  if (c.kind == 1) { has1 = 1; c1 = Node{zi + c.s1, nd.x0, nd.x1}; return c.s1 + nd.x0; }
  const uint32_t n0x = x - c.n1x;
  const uint32_t n0x1 = n0x - n0x0;
  if (n0x0 && n0x1) { has0 = 1; c0 = Node{s0, n0x0, n0x1}; }
  const uint32_t n1x1 = nd.x1 - n0x1;
  const uint32_t n1x0 = c.n1x - n1x1;
  if (n1x0 && n1x1) { has1 = 1; c1 = Node{zi + c.s1, n1x0, n1x1}; }
  return c.s1 + n1x0;
}

// MODE 0: count the queries of each tile.  1: write them.  2: count the children.  3: write children and ranks.
template <int MODE>
__global__ __launch_bounds__(K3_T) void dec_tiles_kernel(DecArgs a) {
  __shared__ uint32_t tp[9];
  __shared__ uint32_t lds_cnt[K3_NPT][4][3];
  if (a.ctl->err) return;
  const uint32_t tid = threadIdx.x;
  if (tid == 0) dec_tile_prefix(a, tp);
  __syncthreads();
  const uint32_t T = tp[8];
  for (uint32_t tile = blockIdx.x; tile < T; tile += gridDim.x) {
    uint32_t p = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) p += (tile >= tp[k]) ? 1u : 0u;
    if (!((a.pmask >> p) & 1u)) continue;                        // (a whole tile: uniform over the block)
    const uint32_t ti = tile - tp[p];
    const uint32_t c0n = a.ctl->cnt[a.par][p][0], M = c0n + a.ctl->cnt[a.par][p][1];
    const Node *src = dec_nodes(a, a.par, p);
    uint32_t *R = a.R + (size_t)p * ((size_t)a.n + 1);
    const uint32_t zi = a.zeros[p];
    Node nd[K3_NPT];
    DCls cl[K3_NPT];
    uint32_t valid[K3_NPT], isq[K3_NPT], ise[K3_NPT], zero[K3_NPT];
    uint32_t bad = 0;
#pragma unroll
    for (int it = 0; it < K3_NPT; ++it) {
      const uint32_t q = ti * K3_TILE + (uint32_t)it * K3_T + tid;
      valid[it] = q < M ? 1u : 0u;
      const uint32_t qq = valid[it] ? q : ti * K3_TILE;
      nd[it] = src[qq < c0n ? qq : (a.capP - 1u - (qq - c0n))];
    }
#pragma unroll
    for (int it = 0; it < K3_NPT; ++it) {
      cl[it] = dec_classify(nd[it], R, a.n);
      bad |= cl[it].bad & valid[it];
      isq[it] = (valid[it] && cl[it].kind == 3u) ? 1u : 0u;
      ise[it] = (isq[it] && cl[it].mx - cl[it].mn + 1u > (uint32_t)kMaxK) ? 1u : 0u;
      zero[it] = 0;
    }
    uint32_t qr[K3_NPT], er[K3_NPT], d2[K3_NPT], tot[3];
    tile_ranks(isq, ise, zero, lds_cnt, qr, er, d2, tot);
    if (bad) a.ctl->err = 1;
    if (MODE == 0) {
      if (tid == 0) { a.tilecnt[(size_t)tile * 4 + 2] = tot[0]; a.tilecnt[(size_t)tile * 4 + 3] = tot[1]; }
      continue;
    }
    const uint32_t qb = a.ctl->qbase[p] + a.tileoff[(size_t)tile * 4 + 2];
    if (MODE == 1) {
      const uint32_t eb = a.ctl->ebase[p] + a.tileoff[(size_t)tile * 4 + 3];
#pragma unroll
      for (int it = 0; it < K3_NPT; ++it)
        if (isq[it]) {
          // get(k, _0x, _x1, _x) :1304; the context of get_context (:671-677) is resolved here when k needs no escape
          const uint32_t x = nd[it].x0 + nd[it].x1, k = cl[it].mx - cl[it].mn + 1u, c1 = x - cl[it].n1x, c2 = nd[it].x1;
          if (k <= (uint32_t)kMaxK) {
            const uint32_t b = a.cfg[p].bits[k];
            const uint32_t ctxv = context_index(b, c1, c2, x);
            a.Q[qb + qr[it]] = k | (ctxv << 5);
          } else {
            a.Q[qb + qr[it]] = kEscape;
            a.E[eb + er[it]] = make_uint4(k, c1, c2, x);
          }
        }
      continue;
    }
    uint32_t has0[K3_NPT], has1[K3_NPT], rval[K3_NPT];
    Node c0[K3_NPT], c1[K3_NPT];
#pragma unroll
    for (int it = 0; it < K3_NPT; ++it) {
      uint32_t v = cl[it].mn;
      if (isq[it]) v += a.res[qb + qr[it]];
      if (valid[it] && cl[it].kind >= 2u && v > cl[it].mx) { bad = 1; v = cl[it].mn; }
      rval[it] = dec_children(nd[it], cl[it], v, zi, has0[it], c0[it], has1[it], c1[it]);
      has0[it] &= valid[it];
      has1[it] &= valid[it];
    }
    if (bad) a.ctl->err = 1;
    uint32_t r0[K3_NPT], r1[K3_NPT];
    tile_ranks(has0, has1, zero, lds_cnt, r0, r1, d2, tot);
    if (MODE == 2) {
      if (tid == 0) { a.tilecnt[(size_t)tile * 4 + 0] = tot[0]; a.tilecnt[(size_t)tile * 4 + 1] = tot[1]; }
      continue;
    }
    const uint32_t o0 = a.tileoff[(size_t)tile * 4 + 0], o1 = a.tileoff[(size_t)tile * 4 + 1];
    Node *dst = dec_nodes(a, a.par ^ 1u, (p + 1u) & 7u);
#pragma unroll
    for (int it = 0; it < K3_NPT; ++it) {
      if (has0[it]) dst[o0 + r0[it]] = c0[it];
      if (has1[it]) dst[a.capP - 1u - (o1 + r1[it])] = c1[it];
      if (valid[it] && !cl[it].bad) R[nd[it].s + nd[it].x0] = rval[it];        // ranks[i].set(s + _x0, s1 + _1x0) :1277,1285,1350
    }
  }
}

// Per-plane exclusive scan of the tile counts (block p = plane p) and the round's bookkeeping by the block that
// finishes last.  QUERY: queries -> tileoff[.][2], query bases, DecInfo for the host.  !QUERY: children ->
// tileoff[.][0..1], next round's list sizes.
template <bool QUERY>
__global__ __launch_bounds__(1024) void dec_scan_kernel(DecArgs a) {
  __shared__ uint32_t tp[9];
  __shared__ uint32_t s_last;
  DecCtl *ctl = a.ctl;
  const uint32_t tid = threadIdx.x, p = blockIdx.x;
  if (!((a.pmask >> p) & 1u)) return;                            // (the whole block)
  if (tid == 0) dec_tile_prefix(a, tp);
  __syncthreads();
  uint32_t r0 = 0, r1 = 0;
  if (!ctl->err) {
    for (uint32_t base = tp[p]; base < tp[p + 1]; base += 1024) {
      const uint32_t t = base + tid;
      const bool valid = t < tp[p + 1];
      if (QUERY) {
        const uint32_t v2 = valid ? a.tilecnt[(size_t)t * 4 + 2] : 0u, v3 = valid ? a.tilecnt[(size_t)t * 4 + 3] : 0u;
        uint64_t tot;
        const uint64_t ex = block_excl_scan_sum64<1024>((uint64_t)v2 | ((uint64_t)v3 << 32), &tot);
        if (valid) {
          a.tileoff[(size_t)t * 4 + 2] = r0 + (uint32_t)ex;
          a.tileoff[(size_t)t * 4 + 3] = r1 + (uint32_t)(ex >> 32);
        }
        r0 += (uint32_t)tot; r1 += (uint32_t)(tot >> 32);
      } else {
        const uint32_t v0 = valid ? a.tilecnt[(size_t)t * 4 + 0] : 0u, v1 = valid ? a.tilecnt[(size_t)t * 4 + 1] : 0u;
        uint64_t tot;
        const uint64_t ex = block_excl_scan_sum64<1024>((uint64_t)v0 | ((uint64_t)v1 << 32), &tot);
        if (valid) {
          a.tileoff[(size_t)t * 4 + 0] = r0 + (uint32_t)ex;
          a.tileoff[(size_t)t * 4 + 1] = r1 + (uint32_t)(ex >> 32);
        }
        r0 += (uint32_t)tot; r1 += (uint32_t)(tot >> 32);
      }
    }
  }
  if (tid == 0) {
    if (QUERY) { ctl->ptot[p][2] = r0; ctl->ptot[p][3] = r1; } else { ctl->ptot[p][0] = r0; ctl->ptot[p][1] = r1; }
    __threadfence();
    s_last = atomicAdd(&ctl->ticket, 1u) == (uint32_t)__popc(a.pmask) - 1u ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last || tid != 0) return;
  __threadfence();
  ctl->ticket = 0;
  const volatile uint32_t (*pt)[4] = ctl->ptot;
  uint64_t curn = 0;
  for (int q = 0; q < 8; ++q) curn += (uint64_t)ctl->cnt[a.par][q][0] + ctl->cnt[a.par][q][1];
  if (QUERY) {
    uint32_t acc = 0, eacc = 0;
    for (int i = 0; i < 8; ++i) {                                 // (planes of an earlier part of this round lie first: their totals stand)
      const uint32_t q = (a.order >> (4 * i)) & 7u;
      if ((a.pmask >> q) & 1u) {
        ctl->qbase[q] = acc;
        ctl->ebase[q] = eacc;
        a.info->qbase[q] = acc;
        a.info->qtot[q] = pt[q][2];
        a.info->ebase[q] = eacc;
        a.info->etot[q] = pt[q][3];
      }
      acc += pt[q][2];
      eacc += pt[q][3];
    }
    a.info->cur_nodes = (uint32_t)curn;
    a.info->nodes_total = ctl->nodes_total;
    __threadfence_system();
    a.info->err = ctl->err;
  } else {
    uint64_t nextn = 0;
    bool ovf = false;
    for (uint32_t q = 0; q < 8; ++q) {
      const uint32_t qn = (q + 1u) & 7u;
      if ((a.pmask >> q) & 1u) {
        ctl->cnt[a.par ^ 1u][qn][0] = pt[q][0];
        ctl->cnt[a.par ^ 1u][qn][1] = pt[q][1];
        if ((uint64_t)pt[q][0] + pt[q][1] > a.capP) ovf = true;
      }
      nextn += (uint64_t)pt[q][0] + pt[q][1];
    }
    if (ovf && !ctl->err) ctl->err = 2;
    if (a.final) {                                               // (every plane's totals are this round's by now)
      ctl->nodes_total += curn;
      ctl->next_nodes = (uint32_t)nextn;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------
// Rounds of up to ~2 M nodes in TWO launches instead of six (round 3).  Between the first wide rounds and the tail lie
// thousands of rounds of 2 K .. 1 M nodes (natural corpus: 4 900, binary corpus: 4 100) in which count / scan / write
// kernels of ~10 us each, three times per round, are pure launch overhead.  Here a pass is ONE kernel, as in
// k3_small_kernel: block = tile, classified once; the tile's counts go out in a 64-bit word tagged with the pass number
// (nothing to reset), the block waits for the words of the earlier tiles (all of them for the query offsets, those of
// its plane for the children), sums them and writes; the block of the last tile folds the words into the control block
// and the host's DecInfo.  A waiter only waits for smaller block indices (in-order dispatch, as k3_small_kernel); the
// wait is bounded and ends in err = 4 instead of a hang.
// ---------------------------------------------------------------------------------------------------------
constexpr uint32_t DS_MAXTILES = 2048;
constexpr uint32_t DS_MAXNODES = (DS_MAXTILES - 8u) * K3_TILE;
__device__ __forceinline__ uint64_t ds_ld(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint64_t ds_wait(const unsigned long long *p, uint64_t epoch, uint64_t w, DecCtl *ctl) {
  uint32_t spins = 0;
  while ((w >> 33) != epoch) {
    __builtin_amdgcn_s_sleep(1);
    w = ds_ld(p);
    if (++spins == (1u << 22)) { ctl->err = 4; return epoch << 33; }
  }
  return w;
}
template <bool QUERY>
__global__ __launch_bounds__(K3_T) void dec_small_kernel(DecArgs a) {
  __shared__ uint32_t tp[9];
  __shared__ uint32_t lds_cnt[K3_NPT][4][3];
  __shared__ unsigned long long s_acc[2];
  __shared__ unsigned long long s_tot[8];
  DecCtl *ctl = a.ctl;
  if (ctl->err) return;
  const uint32_t tid = threadIdx.x;
  if (tid == 0) dec_tile_prefix(a, tp);
  if (tid < 2) s_acc[tid] = 0;
  if (tid < 8) s_tot[tid] = 0;
  __syncthreads();
  const uint32_t T = tp[8], tile = blockIdx.x;
  if (tile >= T) return;
  const uint64_t epoch = ((uint64_t)a.round * 2u + (QUERY ? 1u : 2u)) & 0x7FFFFFFFull;
  uint32_t p = 0;
#pragma unroll
  for (int k = 1; k < 8; ++k) p += (tile >= tp[k]) ? 1u : 0u;
  const uint32_t ti = tile - tp[p];
  const uint32_t c0n = ctl->cnt[a.par][p][0], M = c0n + ctl->cnt[a.par][p][1];
  const Node *src = dec_nodes(a, a.par, p);
  uint32_t *R = a.R + (size_t)p * ((size_t)a.n + 1);
  const uint32_t zi = a.zeros[p];
  Node nd[K3_NPT];
  DCls cl[K3_NPT];
  uint32_t valid[K3_NPT], isq[K3_NPT], ise[K3_NPT], zero[K3_NPT];
  uint32_t bad = 0;
#pragma unroll
  for (int it = 0; it < K3_NPT; ++it) {
    const uint32_t q = ti * K3_TILE + (uint32_t)it * K3_T + tid;
    valid[it] = q < M ? 1u : 0u;
    const uint32_t qq = valid[it] ? q : ti * K3_TILE;
    nd[it] = src[qq < c0n ? qq : (a.capP - 1u - (qq - c0n))];
  }
#pragma unroll
  for (int it = 0; it < K3_NPT; ++it) {
    cl[it] = dec_classify(nd[it], R, a.n);
    bad |= cl[it].bad & valid[it];
    isq[it] = (valid[it] && cl[it].kind == 3u) ? 1u : 0u;
    ise[it] = (isq[it] && cl[it].mx - cl[it].mn + 1u > (uint32_t)kMaxK) ? 1u : 0u;
    zero[it] = 0;
  }
  uint32_t qr[K3_NPT], er[K3_NPT], d2[K3_NPT], tot[3];
  tile_ranks(isq, ise, zero, lds_cnt, qr, er, d2, tot);
  uint32_t has0[K3_NPT], has1[K3_NPT], rval[K3_NPT], r0[K3_NPT], r1[K3_NPT];
  Node c0[K3_NPT], c1[K3_NPT];
  uint32_t qb = 0;
  if (!QUERY) {
    qb = a.tileoff[(size_t)tile * 4 + 2];                      // (left there by the query pass of this round)
#pragma unroll
    for (int it = 0; it < K3_NPT; ++it) {
      uint32_t v = cl[it].mn;
      if (isq[it]) v += a.res[qb + qr[it]];
      if (valid[it] && cl[it].kind >= 2u && v > cl[it].mx) { bad = 1; v = cl[it].mn; }
      rval[it] = dec_children(nd[it], cl[it], v, zi, has0[it], c0[it], has1[it], c1[it]);
      has0[it] &= valid[it];
      has1[it] &= valid[it];
    }
    tile_ranks(has0, has1, zero, lds_cnt, r0, r1, d2, tot);
  }
  if (bad) { ctl->err = 1; __threadfence(); }
  // [10:0] first count, [21:11] second (queries / escapes, or child0 / child1: <= 1024 each), [63:33] pass tag
  if (tid == 0)
    __hip_atomic_store(&a.words[tile], (epoch << 33) | ((uint64_t)tot[1] << 11) | (uint64_t)tot[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // look back: the query pass needs every earlier tile (the query buffer runs through the planes), the children pass
  // the earlier tiles of its own plane
  {
    constexpr int LB = (int)(DS_MAXTILES / K3_T);
    const uint32_t first = QUERY ? 0u : tp[p];
    uint64_t wv[LB], s0 = 0, s1 = 0;
#pragma unroll
    for (int q = 0; q < LB; ++q) { const uint32_t j = tid + (uint32_t)q * K3_T; wv[q] = (j >= first && j < tile) ? ds_ld(&a.words[j]) : 0ull; }
#pragma unroll
    for (int q = 0; q < LB; ++q) {
      const uint32_t j = tid + (uint32_t)q * K3_T;
      if (j >= first && j < tile) {
        wv[q] = ds_wait(&a.words[j], epoch, wv[q], ctl);
        s0 += wv[q] & 0x7FFu; s1 += (wv[q] >> 11) & 0x7FFu;
      }
    }
    if (s0) atomicAdd(&s_acc[0], (unsigned long long)s0);
    if (s1) atomicAdd(&s_acc[1], (unsigned long long)s1);
  }
  __syncthreads();
  const uint32_t o0 = (uint32_t)s_acc[0], o1 = (uint32_t)s_acc[1];
  if (QUERY) {
    if (tid == 0) a.tileoff[(size_t)tile * 4 + 2] = o0;
#pragma unroll
    for (int it = 0; it < K3_NPT; ++it)
      if (isq[it]) {
        const uint32_t x = nd[it].x0 + nd[it].x1, k = cl[it].mx - cl[it].mn + 1u, cc1 = x - cl[it].n1x, cc2 = nd[it].x1;
        if (k <= (uint32_t)kMaxK) {
          const uint32_t b = a.cfg[p].bits[k];
          a.Q[o0 + qr[it]] = k | (context_index(b, cc1, cc2, x) << 5);
        } else {
          a.Q[o0 + qr[it]] = kEscape;
          a.E[o1 + er[it]] = make_uint4(k, cc1, cc2, x);
        }
      }
  } else {
    Node *dst = dec_nodes(a, a.par ^ 1u, (p + 1u) & 7u);
#pragma unroll
    for (int it = 0; it < K3_NPT; ++it) {
      if (has0[it]) dst[o0 + r0[it]] = c0[it];
      if (has1[it]) dst[a.capP - 1u - (o1 + r1[it])] = c1[it];
      if (valid[it] && !cl[it].bad) R[nd[it].s + nd[it].x0] = rval[it];
    }
  }
  // ---- the block of the last tile folds the pass into the control block (and the host's DecInfo) ----
  if (tile != T - 1u) return;
  {
    constexpr int LB = (int)(DS_MAXTILES / K3_T);
    uint64_t wv[LB];
#pragma unroll
    for (int q = 0; q < LB; ++q) { const uint32_t j = tid + (uint32_t)q * K3_T; wv[q] = j < T ? ds_ld(&a.words[j]) : 0ull; }
#pragma unroll
    for (int q = 0; q < LB; ++q) {
      const uint32_t j = tid + (uint32_t)q * K3_T;
      if (j < T) {
        uint32_t pj = 0;
#pragma unroll
        for (int k = 1; k < 8; ++k) pj += (j >= tp[k]) ? 1u : 0u;
        wv[q] = ds_wait(&a.words[j], epoch, wv[q], ctl);
        atomicAdd(&s_tot[pj], (unsigned long long)((wv[q] & 0x7FFu) | (((wv[q] >> 11) & 0x7FFu) << 32)));
      }
    }
  }
  __syncthreads();
  if (tid != 0) return;
  uint64_t curn = 0;
  for (int q = 0; q < 8; ++q) curn += (uint64_t)ctl->cnt[a.par][q][0] + ctl->cnt[a.par][q][1];
  if (QUERY) {
    uint32_t acc = 0, eacc = 0;
    for (int q = 0; q < 8; ++q) {
      const uint32_t tq = (uint32_t)s_tot[q], te = (uint32_t)(s_tot[q] >> 32);
      ctl->qbase[q] = acc; ctl->ebase[q] = eacc;
      a.info->qbase[q] = acc; a.info->qtot[q] = tq;
      a.info->ebase[q] = eacc; a.info->etot[q] = te;
      acc += tq; eacc += te;
    }
    a.info->cur_nodes = (uint32_t)curn;
    a.info->nodes_total = ctl->nodes_total;
    __threadfence_system();
    a.info->err = ctl->err;
  } else {
    uint64_t nextn = 0;
    bool ovf = false;
    for (uint32_t q = 0; q < 8; ++q) {
      const uint32_t qn = (q + 1u) & 7u, t0 = (uint32_t)s_tot[q], t1 = (uint32_t)(s_tot[q] >> 32);
      ctl->cnt[a.par ^ 1u][qn][0] = t0;
      ctl->cnt[a.par ^ 1u][qn][1] = t1;
      nextn += (uint64_t)t0 + t1;
      if ((uint64_t)t0 + t1 > a.capP) ovf = true;
    }
    if (ovf && !ctl->err) ctl->err = 2;
    ctl->nodes_total += curn;
    ctl->next_nodes = (uint32_t)nextn;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Deep tails.  A long repeat is a chain of forced nodes (all zeros / all ones / forced split): thousands to
// millions of rounds in which no decoder is asked anything.  One persistent workgroup keeps the lists in LDS and
// runs those rounds on the device (classify from R, children, the new ranks), and stops IN FRONT of the first
// round that holds a query, does not fit, or is empty; the host then runs that round the normal way.
// ---------------------------------------------------------------------------------------------------------
constexpr int DT_T = 1024;
constexpr int DT_NPT = 4;
constexpr uint32_t DT_CAP = DT_T * DT_NPT;      // nodes of all planes together
constexpr uint32_t DT_ENTER = 2048;             // the host tries the tail kernel at or below this many nodes

// `resume` != 0: the first round is the one a previous launch stopped at; its queries were answered by the host
// (a.res, in the order they were emitted).  When the kernel stops at a round with queries it EMITS them (a.Q / a.E /
// a.info, all host-visible) so that one launch + one sync is all a query round costs in this regime; rounds_done[1]
// = 1 then, 2 if the children outgrow the kernel, 3 on an inconsistent node, 0 when nothing is left.
__global__ __launch_bounds__(DT_T) void dec_tail_kernel(DecArgs a, uint32_t max_rounds, uint32_t *rounds_done, uint32_t resume) {
  __shared__ Node buf[2][DT_CAP];
  __shared__ uint32_t cnt[2][8][2];
  __shared__ uint32_t off[9], noff[9];
  __shared__ uint64_t ws[DT_T / 64];
  __shared__ uint64_t pstart[9];
  DecCtl *ctl = a.ctl;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
  uint32_t par = a.par, cur = 0, executed = 0;
  if (tid < 16) cnt[par][tid >> 1][tid & 1] = ctl->cnt[par][tid >> 1][tid & 1];
  __syncthreads();
  if (tid == 0) {
    uint32_t acc = 0;
    for (int p = 0; p < 8; ++p) { off[p] = acc; acc += cnt[par][p][0] + cnt[par][p][1]; }
    off[8] = acc;
  }
  __syncthreads();
  uint32_t total = off[8];
  if (total > DT_CAP || total == 0 || ctl->err) {
    if (tid == 0) {
      rounds_done[0] = 0; rounds_done[1] = total > DT_CAP ? 2u : 0u; rounds_done[2] = 0; rounds_done[3] = a.seq_base; rounds_done[4] = resume & 1u;
      __threadfence_system();
      if (a.mbox) __hip_atomic_store(&a.mbox[2], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  for (uint32_t q = tid; q < total; q += DT_T) {
    uint32_t p = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) p += (q >= off[k]) ? 1u : 0u;
    const uint32_t i = q - off[p], c0 = cnt[par][p][0];
    buf[0][q] = dec_nodes(a, par, p)[i < c0 ? i : (a.capP - 1u - (i - c0))];
  }
  uint64_t nodes_total = ctl->nodes_total;
  uint32_t why = 0;
  uint32_t seq = a.seq_base, pending = 0;                     // query rounds are numbered (see DecArgs::mbox)
  bool have_ans = (resume & 1u) != 0;                         // the answers of the round about to run are in res
  __shared__ uint32_t got_flag;
  __syncthreads();
  for (;;) {
    if (total == 0 || executed >= max_rounds) break;
    if ((resume & 2u) && executed && total <= 64) break;    // few nodes again: the wave kernel is three times faster per round
    Node nd[DT_NPT];
    DCls cl[DT_NPT];
    uint32_t pl[DT_NPT];
    bool valid[DT_NPT];
    int stop = 0, badn = 0;
    uint64_t qmine = 0;                                       // queries | escape queries << 32 among my nodes
#pragma unroll
    for (int it = 0; it < DT_NPT; ++it) {
      const uint32_t q = tid * DT_NPT + (uint32_t)it;
      valid[it] = q < total;
      nd[it] = valid[it] ? buf[cur][q] : Node{0u, 1u, 1u};
      uint32_t p = 0;
#pragma unroll
      for (int k = 1; k < 8; ++k) p += (q >= off[k]) ? 1u : 0u;
      pl[it] = valid[it] ? p : 0u;
      cl[it] = dec_classify(nd[it], a.R + (size_t)pl[it] * ((size_t)a.n + 1), a.n);
      if (valid[it] && cl[it].bad) badn = 1;
      if (valid[it] && cl[it].kind == 3u) {
        stop = 1;
        qmine += 1ull | ((cl[it].mx - cl[it].mn + 1u > (uint32_t)kMaxK) ? (1ull << 32) : 0ull);
      }
    }
    if (__syncthreads_or(badn)) { why = 3; break; }
    const bool answered = have_ans;                          // this round's queries came back from the host
    uint32_t qx = 0;                                         // exclusive query rank of my first node (list order = stream order)
    if (__syncthreads_or(stop)) {
      // exclusive scan of (queries | escapes << 32) in list order, and the per-plane starts
      uint64_t qinc = wave_incl_sum64(qmine);
      if (lane == 63) ws[wid] = qinc;
      __syncthreads();
      uint64_t qbase = 0, qtot = 0;
#pragma unroll
      for (int i = 0; i < DT_T / 64; ++i) { const uint64_t t = ws[i]; if ((uint32_t)i < wid) qbase += t; qtot += t; }
      const uint64_t qex = qbase + qinc - qmine;
      __syncthreads();
      qx = (uint32_t)qex;
      if (!answered) {
        // emit the round's queries for the host and leave; the lists stay as they are
        if (tid < 9) pstart[tid] = qtot;
        __syncthreads();
        {
          uint64_t run = qex;
#pragma unroll
          for (int it = 0; it < DT_NPT; ++it) {
            const uint32_t q = tid * DT_NPT + (uint32_t)it;
            if (valid[it] && q == off[pl[it]]) pstart[pl[it]] = run;
            if (valid[it] && cl[it].kind == 3u) {
              const uint32_t x = nd[it].x0 + nd[it].x1, k = cl[it].mx - cl[it].mn + 1u, c1v = x - cl[it].n1x, c2v = nd[it].x1;
              if (k <= (uint32_t)kMaxK) {
                const uint32_t b = a.cfg[pl[it]].bits[k];
                const uint32_t ctxv = context_index(b, c1v, c2v, x);
                a.Q[(uint32_t)run] = k | (ctxv << 5);
                run += 1ull;
              } else {
                a.Q[(uint32_t)run] = kEscape;
                a.E[(uint32_t)(run >> 32)] = make_uint4(k, c1v, c2v, x);
                run += 1ull | (1ull << 32);
              }
            }
          }
        }
        __syncthreads();
        if (tid == 0) {
          for (int p = 7; p >= 0; --p) if (off[p] == off[p + 1]) pstart[p] = pstart[p + 1];
          for (int p = 0; p < 8; ++p) {
            a.info->qbase[p] = (uint32_t)pstart[p];
            a.info->qtot[p] = (uint32_t)(pstart[p + 1] - pstart[p]);
            a.info->ebase[p] = (uint32_t)(pstart[p] >> 32);
            a.info->etot[p] = (uint32_t)((pstart[p + 1] - pstart[p]) >> 32);
          }
          a.info->cur_nodes = total;
          a.info->err = 0;
        }
        pending = seq;
        bool got = false;
        if (a.mbox) {                                          // stay resident: see dec_tail64_kernel
          __threadfence_system();
          __syncthreads();
          if (tid == 0) {
            __hip_atomic_store(&a.mbox[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            const uint64_t t0 = wall_clock64();
            uint32_t g = 0;
            for (;;) {
              if (__hip_atomic_load(&a.mbox[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == seq) { g = 1; break; }
              if (wall_clock64() - t0 > 200000ull) break;
              __builtin_amdgcn_s_sleep(4);
            }
            got_flag = g;
          }
          __syncthreads();
          got = got_flag != 0;
          __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
        if (!got) { why = 1; break; }
        ++seq; pending = 0; have_ans = true;
      }
    }
    uint32_t has0[DT_NPT], has1[DT_NPT], rval[DT_NPT];
    Node c0[DT_NPT], c1[DT_NPT];
    uint64_t mine = 0;
    int over = 0;
#pragma unroll
    for (int it = 0; it < DT_NPT; ++it) {
      uint32_t v = cl[it].mn;
      if (valid[it] && cl[it].kind == 3u) { v += __hip_atomic_load(const_cast<uint32_t *>(&a.res[qx]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); ++qx; if (v > cl[it].mx) { over = 1; v = cl[it].mn; } }
      rval[it] = dec_children(nd[it], cl[it], v, a.zeros[pl[it]], has0[it], c0[it], has1[it], c1[it]);
      if (!valid[it]) has0[it] = has1[it] = 0;
      mine += (uint64_t)has0[it] | ((uint64_t)has1[it] << 32);
    }
    if (__syncthreads_or(over)) { why = 3; break; }          // an answer outside [mn, mx]: inconsistent archive
    // block exclusive scan of (child0 | child1 << 32) in list order
    uint64_t inc = wave_incl_sum64(mine);
    if (lane == 63) ws[wid] = inc;
    __syncthreads();
    uint64_t wbase = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < DT_T / 64; ++i) { const uint64_t t = ws[i]; if ((uint32_t)i < wid) wbase += t; tot += t; }
    const uint64_t ex = wbase + inc - mine;
    if (tid < 9) pstart[tid] = tot;
    __syncthreads();
    {
      uint64_t run = ex;
#pragma unroll
      for (int it = 0; it < DT_NPT; ++it) {
        const uint32_t q = tid * DT_NPT + (uint32_t)it;
        if (valid[it] && q == off[pl[it]]) pstart[pl[it]] = run;
        run += (uint64_t)has0[it] | ((uint64_t)has1[it] << 32);
      }
    }
    __syncthreads();
    if (tid == 0) {
      for (int p = 7; p >= 0; --p) if (off[p] == off[p + 1]) pstart[p] = pstart[p + 1];
      uint32_t acc = 0;
      for (int pn = 0; pn < 8; ++pn) {
        const int p = (pn + 7) & 7;
        const uint64_t d = pstart[p + 1] - pstart[p];
        cnt[par ^ 1u][pn][0] = (uint32_t)d;
        cnt[par ^ 1u][pn][1] = (uint32_t)(d >> 32);
        noff[pn] = acc;
        acc += (uint32_t)d + (uint32_t)(d >> 32);
      }
      noff[8] = acc;
    }
    __syncthreads();
    if (noff[8] > DT_CAP) { why = 2; break; }               // nothing of this round has been written yet
    {
      uint64_t run = ex;
#pragma unroll
      for (int it = 0; it < DT_NPT; ++it) {
        if (valid[it]) {
          const uint32_t p = pl[it], pn = (p + 1u) & 7u;
          const uint64_t rel = run - pstart[p];
          if (has0[it]) buf[cur ^ 1u][noff[pn] + (uint32_t)rel] = c0[it];
          if (has1[it]) buf[cur ^ 1u][noff[pn] + cnt[par ^ 1u][pn][0] + (uint32_t)(rel >> 32)] = c1[it];
          (a.R + (size_t)p * ((size_t)a.n + 1))[nd[it].s + nd[it].x0] = rval[it];
        }
        run += (uint64_t)has0[it] | ((uint64_t)has1[it] << 32);
      }
    }
    nodes_total += total;
    __syncthreads();                                         // also orders the R stores before the next round's loads
    if (tid < 9) off[tid] = noff[tid];
    __syncthreads();
    total = off[8];
    par ^= 1u; cur ^= 1u; ++executed; have_ans = false;
  }
  __syncthreads();
  for (uint32_t q = tid; q < total; q += DT_T) {
    uint32_t p = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) p += (q >= off[k]) ? 1u : 0u;
    const uint32_t i = q - off[p], c0 = cnt[par][p][0];
    dec_nodes(a, par, p)[i < c0 ? i : (a.capP - 1u - (i - c0))] = buf[cur][q];
  }
  if (tid < 16) ctl->cnt[par][tid >> 1][tid & 1] = cnt[par][tid >> 1][tid & 1];
  if (tid == 0) {
    ctl->nodes_total = nodes_total;
    ctl->next_nodes = total;
    rounds_done[0] = executed;
    rounds_done[1] = why;
    rounds_done[2] = pending;
    rounds_done[3] = seq;
    rounds_done[4] = have_ans ? 1u : 0u;
    __threadfence_system();
    if (a.mbox) __hip_atomic_store(&a.mbox[2], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// The same for at most 64 nodes (the very deep tails: a handful of chains): ONE wave, one node per lane, the new
// order from ballots instead of a block scan -- no workgroup barriers on the round's critical path.
// rounds_done[1] says why it stopped: 1 query / inconsistent node, 2 more than 64 children, 0 nothing left / max.
// Like dec_tail_kernel it emits the queries of the round it stops at and resumes with the answers (`resume`).
__global__ __launch_bounds__(64) void dec_tail64_kernel(DecArgs a, uint32_t max_rounds, uint32_t *rounds_done, uint32_t resume) {
  __shared__ Node nbuf[64];
  __shared__ uint32_t pbuf[64];
  DecCtl *ctl = a.ctl;
  const uint32_t lane = threadIdx.x;
  const uint64_t below = (1ull << lane) - 1ull;
  uint32_t par = a.par, executed = 0, why = 0;
  uint32_t cnt[8][2], total = 0;
#pragma unroll
  for (int p = 0; p < 8; ++p) { cnt[p][0] = ctl->cnt[par][p][0]; cnt[p][1] = ctl->cnt[par][p][1]; total += cnt[p][0] + cnt[p][1]; }
  if (total > 64 || total == 0 || ctl->err) {
    if (lane == 0) {
      rounds_done[0] = 0; rounds_done[1] = total > 64 ? 2u : 0u; rounds_done[2] = 0; rounds_done[3] = a.seq_base; rounds_done[4] = resume & 1u;
      __threadfence_system();
      if (a.mbox) __hip_atomic_store(&a.mbox[2], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  Node nd{0u, 1u, 1u};
  uint32_t pl = 0;
  {
    uint32_t acc = 0;
#pragma unroll
    for (uint32_t p = 0; p < 8; ++p) {
      const uint32_t m = cnt[p][0] + cnt[p][1];
      if (lane >= acc && lane < acc + m) {
        const uint32_t i = lane - acc;
        nd = dec_nodes(a, par, p)[i < cnt[p][0] ? i : (a.capP - 1u - (i - cnt[p][0]))];
        pl = p;
      }
      acc += m;
    }
  }
  uint64_t nodes_total = ctl->nodes_total;
  uint32_t seq = a.seq_base, pending = 0;                     // query rounds are numbered; `pending` = the one I left at
  bool have_ans = (resume & 1u) != 0;                         // the answers of the round about to run are in res
  for (;;) {
    if (total == 0 || executed >= max_rounds) break;
    const bool valid = lane < total;
    const DCls cl = dec_classify(nd, a.R + (size_t)pl * ((size_t)a.n + 1), a.n);
    if (__ballot(valid && cl.bad)) { why = 3; break; }
    const bool isq = valid && cl.kind == 3u;
    const uint32_t kq = cl.mx - cl.mn + 1u;
    const bool ise = isq && kq > (uint32_t)kMaxK;
    const uint64_t mq = __ballot(isq), me = __ballot(ise);
    const uint32_t qidx = (uint32_t)__popcll(mq & below), eidx = (uint32_t)__popcll(me & below);   // lane order = stream order
    const bool answered = have_ans;                            // this round's queries came back from the host
    if (mq && !answered) {
      // emit the round's queries (host-visible buffers) and leave; the lists stay as they are
      if (isq) {
        const uint32_t x = nd.x0 + nd.x1, c1v = x - cl.n1x, c2v = nd.x1;
        if (!ise) {
          const uint32_t b = a.cfg[pl].bits[kq];
          a.Q[qidx] = kq | ((context_index(b, c1v, c2v, x)) << 5);
        } else {
          a.Q[qidx] = kEscape;
          a.E[eidx] = make_uint4(kq, c1v, c2v, x);
        }
      }
#pragma unroll
      for (uint32_t p = 0; p < 8; ++p) {
        const uint64_t pm = __ballot(valid && pl == p);
        const uint64_t lowp = pm ? ((pm & (0ull - pm)) - 1ull) : ~0ull;      // lanes before plane p's first node
        if (lane == p) {
          a.info->qbase[p] = (uint32_t)__popcll(mq & lowp & (pm ? ~0ull : 0ull)) + (pm ? 0u : 0u);
          a.info->qtot[p] = (uint32_t)__popcll(mq & pm);
          a.info->ebase[p] = (uint32_t)__popcll(me & lowp & (pm ? ~0ull : 0ull));
          a.info->etot[p] = (uint32_t)__popcll(me & pm);
        }
      }
      if (lane == 0) { a.info->cur_nodes = total; a.info->err = 0; }
      pending = seq;
      bool got = false;
      if (a.mbox) {
        // Stay resident: publish the round's number, wait for the host to post the same number back (it polls the
        // mailbox while the kernel runs), at most ~2 ms -- then leave as without a mailbox; the host answers and relaunches.
        __threadfence_system();
        if (lane == 0) __hip_atomic_store(&a.mbox[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint64_t t0 = wall_clock64();                    // 100 MHz
        for (;;) {
          uint32_t v = 0;
          if (lane == 0) v = __hip_atomic_load(&a.mbox[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
          v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
          if (v == seq) { got = true; break; }
          if (wall_clock64() - t0 > 200000ull) break;
          __builtin_amdgcn_s_sleep(4);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
      }
      if (!got) { why = 1; break; }
      ++seq; pending = 0; have_ans = true;
    }
    uint32_t has0, has1;
    Node c0, c1;
    uint32_t v = cl.mn;
    if (isq) v += __hip_atomic_load(const_cast<uint32_t *>(&a.res[qidx]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // host memory, possibly written while we run
    if (__ballot(isq && v > cl.mx)) { why = 3; break; }        // an answer outside [mn, mx]: inconsistent archive
    const uint32_t rval = dec_children(nd, cl, v, a.zeros[pl], has0, c0, has1, c1);
    if (!valid) has0 = has1 = 0;
    const uint64_t m0 = __ballot(has0), m1 = __ballot(has1);
    if (__popcll(m0) + __popcll(m1) > 64) { why = 2; break; }
    // next lists: plane pn takes the children of plane pn - 1: child0s in order, then child1s in order
    uint32_t base = 0, my0 = 0, my1 = 0, ncnt[8][2];
#pragma unroll
    for (uint32_t pn = 0; pn < 8; ++pn) {
      const uint32_t p = (pn + 7u) & 7u;
      const uint64_t pm = __ballot(valid && pl == p);
      const uint32_t t0 = (uint32_t)__popcll(m0 & pm), t1 = (uint32_t)__popcll(m1 & pm);
      if (pl == p) {
        my0 = base + (uint32_t)__popcll(m0 & pm & below);
        my1 = base + t0 + (uint32_t)__popcll(m1 & pm & below);
      }
      ncnt[pn][0] = t0; ncnt[pn][1] = t1;
      base += t0 + t1;
    }
    if (has0) { nbuf[my0] = c0; pbuf[my0] = (pl + 1u) & 7u; }
    if (has1) { nbuf[my1] = c1; pbuf[my1] = (pl + 1u) & 7u; }
    if (valid) (a.R + (size_t)pl * ((size_t)a.n + 1))[nd.s + nd.x0] = rval;
    nodes_total += total;
    __syncthreads();                                         // one wave: orders LDS and the R stores before the next loads
    total = base;
    nd = lane < total ? nbuf[lane] : Node{0u, 1u, 1u};
    pl = lane < total ? pbuf[lane] : 0u;
#pragma unroll
    for (int p = 0; p < 8; ++p) { cnt[p][0] = ncnt[p][0]; cnt[p][1] = ncnt[p][1]; }
    __syncthreads();
    par ^= 1u; ++executed; have_ans = false;
  }
  // hand the state back
  {
    uint32_t acc = 0;
#pragma unroll
    for (uint32_t p = 0; p < 8; ++p) {
      const uint32_t m = cnt[p][0] + cnt[p][1];
      if (lane >= acc && lane < acc + m && lane < total) {
        const uint32_t i = lane - acc;
        dec_nodes(a, par, p)[i < cnt[p][0] ? i : (a.capP - 1u - (i - cnt[p][0]))] = nd;
      }
      acc += m;
    }
  }
  if (lane < 16) ctl->cnt[par][lane >> 1][lane & 1] = cnt[lane >> 1][lane & 1];
  if (lane == 0) {
    ctl->nodes_total = nodes_total;
    ctl->next_nodes = total;
    rounds_done[0] = executed;
    rounds_done[1] = why;
    rounds_done[2] = pending;                                  // the query round whose queries are out (why == 1)
    rounds_done[3] = seq;                                      // the next number
    rounds_done[4] = have_ans ? 1u : 0u;                       // the round I stopped at has been answered (the decoders moved on)
    __threadfence_system();
    if (a.mbox) __hip_atomic_store(&a.mbox[2], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ---------------------------------------------------------------------------------------------------------
// After the last round: R -> plane bits -> rank granules.  Between two known boundaries a < b a plane is
// constant: all ones iff R[b] - R[a] == b - a, all zeros iff R[b] == R[a] (anything else: inconsistent archive).
// ---------------------------------------------------------------------------------------------------------
constexpr int FG_T = 256;
constexpr uint32_t FG_PER = 32;                        // positions per thread = one 32-bit word of the plane
constexpr uint32_t FG_CHUNK = FG_T * FG_PER;           // 8192 positions per block

struct FillArgs {
  const uint32_t *R;        // [8][n + 1]
  uint32_t *cmax;           // [8][chunks]: last known index + 1 inside the chunk (0 = none), then its exclusive prefix max
  uint8_t *type;            // [8][n + 1]: at a known index a: 1 if the gap that starts at a is all ones
  uint32_t *words;          // [8][nwords] plane bits, LSB first
  uint32_t *rankw;          // [8][nwords] rank1 at the first position of each word
  uint32_t n, chunks, nwords;
  uint32_t *err;
};

__global__ __launch_bounds__(FG_T) void fill_chunkmax_kernel(FillArgs a) {
  const uint32_t p = blockIdx.y, c = blockIdx.x;
  const uint32_t *R = a.R + (size_t)p * ((size_t)a.n + 1);
  uint32_t best = 0;
  const uint64_t base = (uint64_t)c * FG_CHUNK;
  for (uint32_t i = threadIdx.x; i < FG_CHUNK; i += FG_T) {
    const uint64_t q = base + i;
    if (q <= a.n && R[q] != kUnknown) best = (uint32_t)q + 1u;
  }
  uint32_t tot;
  (void)block_incl_scan_max<FG_T>(best, &tot);
  if (threadIdx.x == 0) a.cmax[(size_t)p * a.chunks + c] = tot;
}

// exclusive prefix max over the chunks of one plane (block = plane)
__global__ __launch_bounds__(1024) void fill_chunkscan_kernel(FillArgs a) {
  uint32_t *cm = a.cmax + (size_t)blockIdx.x * a.chunks;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < a.chunks; base += 1024) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < a.chunks ? cm[i] : 0u;
    uint32_t tot;
    const uint32_t inc = block_incl_scan_max<1024>(v, &tot);
    __shared__ uint32_t sh[1024];
    sh[threadIdx.x] = inc;
    __syncthreads();
    uint32_t ex = threadIdx.x ? sh[threadIdx.x - 1] : 0u;
    ex = ex > carry ? ex : carry;
    __syncthreads();
    if (i < a.chunks) cm[i] = ex;
    carry = carry > tot ? carry : tot;
  }
}

// PASS 0: gap types at the known indices.  PASS 1: plane words + word ranks.
template <int PASS>
__global__ __launch_bounds__(FG_T) void fill_kernel(FillArgs a) {
  __shared__ uint32_t sh[FG_T];
  const uint32_t p = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
  const uint32_t *R = a.R + (size_t)p * ((size_t)a.n + 1);
  uint8_t *type = a.type + (size_t)p * ((size_t)a.n + 1);
  const uint64_t q0 = (uint64_t)c * FG_CHUNK + (uint64_t)tid * FG_PER;
  // last known index + 1 among my positions, then the same for everything before me
  uint32_t r[FG_PER];
  uint32_t mine = 0;
#pragma unroll
  for (uint32_t i = 0; i < FG_PER; ++i) {
    const uint64_t q = q0 + i;
    r[i] = q <= a.n ? R[q] : kUnknown;
    if (r[i] != kUnknown) mine = (uint32_t)q + 1u;
  }
  uint32_t tot;
  const uint32_t inc = block_incl_scan_max<FG_T>(mine, &tot);
  sh[tid] = inc;
  __syncthreads();
  uint32_t last = tid ? sh[tid - 1] : 0u;                     // known index + 1 before my range, inside the chunk
  const uint32_t carry = a.cmax[(size_t)p * a.chunks + c];
  last = last > carry ? last : carry;                         // ... or in an earlier chunk (index 0 is always known)
  if (PASS == 0) {
    uint32_t a_idx = last ? last - 1u : 0u;
    uint32_t ra = last ? R[a_idx] : 0u;
#pragma unroll
    for (uint32_t i = 0; i < FG_PER; ++i) {
      const uint64_t q = q0 + i;
      if (q == 0 || q > a.n || r[i] == kUnknown) { if (q == 0 && r[i] != kUnknown) { a_idx = 0; ra = r[i]; } continue; }
      const uint32_t ones = r[i] - ra, wdt = (uint32_t)q - a_idx;
      if (r[i] < ra || (ones != 0 && ones != wdt)) *a.err = 1;   // a mixed gap that was never split
      type[a_idx] = ones == wdt ? 1 : 0;
      a_idx = (uint32_t)q; ra = r[i];
    }
  } else {
    uint32_t a_idx = last ? last - 1u : 0u;
    uint32_t ra = R[a_idx];
    uint32_t ty = type[a_idx];
    uint32_t word = 0;
    const uint32_t rank_first = ra + ty * ((uint32_t)q0 - a_idx);   // valid when q0 <= n (checked below)
    uint32_t rank0 = rank_first;
#pragma unroll
    for (uint32_t i = 0; i < FG_PER; ++i) {
      const uint64_t q = q0 + i;
      if (q >= a.n) break;
      if (r[i] != kUnknown) { a_idx = (uint32_t)q; ra = r[i]; ty = type[a_idx]; if (i == 0) rank0 = ra; }
      word |= ty << i;
    }
    const uint64_t w = q0 / 32;
    if (w < a.nwords) {
      a.words[(size_t)p * a.nwords + w] = word;
      a.rankw[(size_t)p * a.nwords + w] = q0 <= a.n ? rank0 : 0u;
    }
  }
}

__global__ void gran_from_words_kernel(FillArgs a, Granule *gran, uint32_t ngran) {
  const uint32_t p = blockIdx.y;
  for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < ngran; g += gridDim.x * blockDim.x) {
    const uint32_t *W = a.words + (size_t)p * a.nwords, *RW = a.rankw + (size_t)p * a.nwords;
    const uint64_t w = (uint64_t)g * 3;
    Granule G;
    G.w0 = w < a.nwords ? W[w] : 0u;
    G.w1 = w + 1 < a.nwords ? W[w + 1] : 0u;
    G.w2 = w + 2 < a.nwords ? W[w + 2] : 0u;
    G.cum = w < a.nwords ? RW[w] : 0u;
    gran[(size_t)p * ngran + g] = G;
  }
}

// unbwt::bytewise (bce.cpp:1043-1085) as wavelet-matrix access: byte i = the bits met while following position
// i through the eight stable partitions (the reference's cursor heap D does the same walk for all i in order).
__global__ void access_kernel(const Granule *__restrict__ gran, uint32_t ngran, uint32_t n, const uint32_t *zeros8,
                              uint8_t *__restrict__ bwt) {
  uint32_t zeros[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) zeros[j] = zeros8[j];
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    uint32_t pos = i, chr = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t g = div96(pos), o = pos - g * 96u;
      const Granule G = gran[(size_t)j * ngran + g];
      const uint32_t r1 = granule_rank1(G, o);
      const uint32_t wsel = o >> 5, word = wsel == 0 ? G.w0 : (wsel == 1 ? G.w1 : G.w2);
      const uint32_t b = (word >> (o & 31u)) & 1u;
      chr |= b << j;
      pos = b ? zeros[j] + r1 : pos - r1;
    }
    bwt[i] = (uint8_t)chr;
  }
}

// ---- inverse BWT: LF by one stable radix pass on the byte, then concurrent walkers between marked rows ----
__global__ void lf_keys_kernel(const uint8_t *__restrict__ bwt, uint32_t n, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { keys[i] = bwt[i]; vals[i] = i; }
}
__global__ void lf_scatter_kernel(const uint32_t *__restrict__ sorted_rows, uint32_t n, uint32_t *__restrict__ lf) {
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) lf[sorted_rows[k]] = k;
}
// walker j starts at row j << sh and stops in front of the next row that is a multiple of 2^sh
__global__ void walk_len_kernel(const uint32_t *__restrict__ lf, uint32_t m, uint32_t sh, uint32_t *__restrict__ len,
                                uint32_t *__restrict__ endrow) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const uint32_t mask = (1u << sh) - 1u;
  uint32_t row = j << sh, l = 0;
  do { row = lf[row]; ++l; } while (row & mask);
  len[j] = l; endrow[j] = row;
}
// text[i] = bwt[row_i] along the walk; text position i lands at out[(i + off) % n]  (std::rotate, bce.cpp:1093)
__global__ void walk_write_kernel(const uint32_t *__restrict__ lf, const uint8_t *__restrict__ bwt, uint32_t m, uint32_t sh,
                                  const uint32_t *__restrict__ len, const uint32_t *__restrict__ dest_end, uint32_t n,
                                  uint32_t off, uint8_t *__restrict__ out) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  uint32_t row = j << sh;
  const uint32_t l = len[j];
  uint64_t i = dest_end[j];                                   // one past my last text position
  for (uint32_t t = 0; t < l; ++t) {
    --i;
    uint64_t o = i + off;
    if (o >= n) o -= n;
    out[o] = bwt[row];
    row = lf[row];
  }
}

// periodic inputs: the walk from row 0 goes round ONE cycle of the LF mapping (length lc) again and again; V holds that
// cycle as the last lc text positions would, and text[i] = V[lc - 1 - ((n - 1 - i) mod lc)]
__global__ void expand_cycle_kernel(const uint8_t *__restrict__ V, uint32_t lc, uint32_t n, uint32_t off, uint8_t *__restrict__ out) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    uint64_t o = (uint64_t)i + off;
    if (o >= n) o -= n;
    out[o] = V[lc - 1u - ((n - 1u - i) % lc)];
  }
}

// ---- the libdivsufsort seam: inverse_bw_transform(T, U, A, n, idx) (include/divsufsort_hip.h; bce.cpp:1091) ----
// T = the BWT divbwt produces (n bytes, the suffix-0 row left out at 1-based position idx).  With the row put back as a
// sentinel the n + 1 rows are the BWT of the rotations of T$: LF by one stable radix pass on 9-bit symbols (0 = $), then
// the decoder's walk from row 0 -- the rotation that starts with $, preceded by the last text byte -- backwards through
// the text.  One cycle of n + 1 rows always (the sentinel is unique).
__global__ void seam_rows_kernel(const uint8_t *__restrict__ U, uint32_t n, uint32_t idx, uint8_t *__restrict__ rows,
                                 uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += gridDim.x * blockDim.x) {
    const uint32_t b = i == idx ? 0u : (uint32_t)U[i < idx ? i : i - 1u];
    rows[i] = (uint8_t)b;
    keys[i] = i == idx ? 0u : b + 1u;
    vals[i] = i;
  }
}

}  // namespace

int kd_inverse_bw_transform(bce_hip_ctx *c, const uint8_t *T_host, uint8_t *U_host, uint32_t n, uint32_t idx) {
  if (n == 0 || n >= 0x7FFFFFFFu || idx == 0 || idx > n) return BCE_HIP_E_ARG;
  const uint32_t m_rows = n + 1u;
  const size_t b4 = (size_t)m_rows * 4;
  BCE_TRY(ensure(c, c->text, m_rows));
  BCE_TRY(ensure(c, c->bwt, m_rows));
  BCE_TRY(ensure(c, c->ptmp[0], m_rows));
  for (int i = 0; i < 2; ++i) { BCE_TRY(ensure(c, c->sa[i], b4)); BCE_TRY(ensure(c, c->key[i], b4)); }
  BCE_TRY(ensure(c, c->rank, b4));
  c->stage = 0; c->k1_valid = false; c->enum_active = false;          // the context's compression state is gone
  uint8_t *d_in = c->ptmp[0].as<uint8_t>(), *rows = c->bwt.as<uint8_t>();
  BCE_HIP_TRY(c, hipMemcpyAsync(d_in, T_host, n, hipMemcpyHostToDevice, c->stream));
  uint32_t *key[2] = {c->key[0].as<uint32_t>(), c->key[1].as<uint32_t>()};
  uint32_t *val[2] = {c->sa[0].as<uint32_t>(), c->sa[1].as<uint32_t>()};
  uint32_t *lf = c->rank.as<uint32_t>();
  const uint32_t gn = (uint32_t)((((uint64_t)m_rows + 255) / 256) < 8192 ? (((uint64_t)m_rows + 255) / 256) : 8192);
  hipLaunchKernelGGL(seam_rows_kernel, dim3(gn), dim3(256), 0, c->stream, d_in, n, idx, rows, key[0], val[0]);
  int res = 0;
  BCE_TRY(radix_sort_pairs(c, key, val, m_rows, 0, 9, &res, 9));
  hipLaunchKernelGGL(lf_scatter_kernel, dim3(gn), dim3(256), 0, c->stream, val[res], m_rows, lf);
  uint32_t sh = 0;
  while (((uint64_t)m_rows >> sh) > (1u << 19)) ++sh;          // at most 2^19 walkers
  while (sh < 8 && ((uint64_t)m_rows >> (sh + 1)) >= 4096) ++sh;   // ... of at least 256 rows each while a few thousand are left (see decompress_device)
  const uint32_t m = (uint32_t)((((uint64_t)m_rows - 1) >> sh) + 1);
  BCE_TRY(ensure(c, c->key[0], (size_t)3 * m * 4 + 16 > b4 ? (size_t)3 * m * 4 + 16 : b4));
  uint32_t *d_len = c->key[0].as<uint32_t>(), *d_end = d_len + m, *d_dest = d_len + 2 * (size_t)m;
  hipLaunchKernelGGL(walk_len_kernel, dim3((m + 63) / 64), dim3(64), 0, c->stream, lf, m, sh, d_len, d_end);
  std::vector<uint32_t> h_len(m), h_end(m), h_dest(m, 0);
  BCE_HIP_TRY(c, hipMemcpyAsync(h_len.data(), d_len, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
  BCE_HIP_TRY(c, hipMemcpyAsync(h_end.data(), d_end, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
  BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
  BCE_HIP_TRY(c, hipGetLastError());
  uint64_t lc = 0;
  {
    uint32_t cur = 0;
    std::vector<uint8_t> visited(m, 0);
    std::vector<uint32_t> order;
    do {
      const uint32_t j = cur >> sh;
      if (visited[j]) return BCE_HIP_E_ARG;                      // not the BWT of anything (T / idx inconsistent)
      visited[j] = 1;
      order.push_back(j);
      lc += h_len[j];
      cur = h_end[j];
    } while (cur != 0);
    uint64_t pos = lc;
    for (uint32_t j : order) { h_dest[j] = (uint32_t)pos; pos -= h_len[j]; }
    for (uint32_t j = 0; j < m; ++j) if (!visited[j]) h_len[j] = 0;
  }
  if (lc != m_rows) return BCE_HIP_E_ARG;                        // several LF cycles: T / idx inconsistent
  BCE_HIP_TRY(c, hipMemcpyAsync(d_dest, h_dest.data(), (size_t)m * 4, hipMemcpyHostToDevice, c->stream));
  BCE_HIP_TRY(c, hipMemcpyAsync(d_len, h_len.data(), (size_t)m * 4, hipMemcpyHostToDevice, c->stream));
  // the walk's positions are those of $T: shifted by n (= -1 mod n + 1) the text lands at [0, n), the sentinel's slot at n
  hipLaunchKernelGGL(walk_write_kernel, dim3((m + 63) / 64), dim3(64), 0, c->stream, lf, rows, m, sh, d_len, d_dest, m_rows, n,
                     c->text.as<uint8_t>());
  BCE_HIP_TRY(c, hipMemcpyAsync(U_host, c->text.p, n, hipMemcpyDeviceToHost, c->stream));
  BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
  BCE_HIP_TRY(c, hipGetLastError());
  return BCE_HIP_OK;
}

namespace {

// ---- the host's part of a round: the eight plane decoders answer their queries, in parallel ----
struct QueryPool {
  std::vector<Decoder> *dec = nullptr;
  const uint32_t *Q = nullptr;
  const uint4 *E = nullptr;
  uint32_t *res = nullptr;
  DecInfo info;
  std::thread th[8];
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  struct Job { const uint32_t *q; const uint4 *e; uint32_t *r; uint32_t cnt; };
  double t_begin[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_end[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double busy[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // seconds each plane's thread has spent answering, and how many queries (BCE_DEC_TIMING)
  uint64_t asked_n[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  Job job[8] = {};
  uint64_t asked[8] = {0, 0, 0, 0, 0, 0, 0, 0}, answered[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool stop = false;

  static void answer(Decoder &d, const uint32_t *q, const uint4 *e, uint32_t *r, uint32_t cnt) {
    static_assert(sizeof(Decoder::Esc) == sizeof(uint4) && Decoder::kEscapeQuery == kEscape, "query formats");
    d.answer_batch(q, reinterpret_cast<const Decoder::Esc *>(e), r, cnt);
  }
  void worker(int p) {
    uint64_t seen = 0;
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_go.wait(lk, [&] { return stop || asked[p] != seen; });
        if (stop) return;
        seen = asked[p];
        j = job[p];
      }
      const double ta = now_s();
      answer((*dec)[p], j.q, j.e, j.r, j.cnt);
      t_begin[p] = ta;
      t_end[p] = now_s();
      busy[p] += t_end[p] - ta;
      asked_n[p] += j.cnt;
      {
        std::lock_guard<std::mutex> g(mu);
        answered[p] = seen;
      }
      cv_done.notify_all();
    }
  }
  void start() { for (int p = 0; p < 8; ++p) th[p] = std::thread([this, p] { worker(p); }); }
  // the planes of `mask` start on jobs[p] (queries, escape records, where the answers go, how many)
  void run_async(const Job jobs[8], uint32_t mask) {
    {
      std::lock_guard<std::mutex> g(mu);
      for (int p = 0; p < 8; ++p)
        if ((mask >> p) & 1u) { job[p] = jobs[p]; ++asked[p]; }
    }
    cv_go.notify_all();
  }
  void wait(uint32_t mask) {
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { for (int p = 0; p < 8; ++p) if (((mask >> p) & 1u) && answered[p] != asked[p]) return false; return true; });
  }
  uint32_t done_locked(uint32_t mask) const {
    uint32_t fin = 0;
    for (int p = 0; p < 8; ++p) if (((mask >> p) & 1u) && answered[p] == asked[p]) fin |= 1u << p;
    return fin;
  }
  uint32_t done(uint32_t mask) { std::lock_guard<std::mutex> g(mu); return done_locked(mask); }      // which of `mask` have answered
  uint32_t wait_any(uint32_t mask) {                                                                 // ... at least one of them (mask != 0)
    std::unique_lock<std::mutex> lk(mu);
    uint32_t fin = 0;
    cv_done.wait(lk, [&] { fin = done_locked(mask); return fin != 0 || mask == 0; });
    return fin;
  }
  void run(const DecInfo &in) {
    uint64_t total = 0;
    for (int p = 0; p < 8; ++p) total += in.qtot[p];
    if (total < 2048) {                                          // not worth waking anybody
      for (int p = 0; p < 8; ++p) answer((*dec)[p], Q + in.qbase[p], E + in.ebase[p], res + in.qbase[p], in.qtot[p]);
      return;
    }
    Job jobs[8];
    for (int p = 0; p < 8; ++p) jobs[p] = Job{Q + in.qbase[p], E + in.ebase[p], res + in.qbase[p], in.qtot[p]};
    run_async(jobs, 0xFFu);
    wait(0xFFu);
  }
  ~QueryPool() {
    { std::lock_guard<std::mutex> g(mu); stop = true; }
    cv_go.notify_all();
    for (int p = 0; p < 8; ++p) if (th[p].joinable()) th[p].join();
  }
};

struct Pinned {
  void *p = nullptr;
  size_t cap = 0;
  void **keep_p = nullptr;      // where the buffer lives on after this object (a slot of the context), if anywhere
  size_t *keep_cap = nullptr;
  Pinned() = default;
  Pinned(void **kp, size_t *kc) : p(*kp), cap(*kc), keep_p(kp), keep_cap(kc) {}
  Pinned(const Pinned &) = delete;
  Pinned &operator=(const Pinned &) = delete;
  int ensure(bce_hip_ctx *c, size_t bytes, size_t least = (size_t)16 << 20) {
    if (bytes <= cap) return BCE_HIP_OK;
    size_t want = 2 * cap > bytes ? 2 * cap : bytes;          // pinning is slow (~0.15 s per GB): grow geometrically
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
    if (want < least) want = least;
    const double t0 = now_s();
    // (the runtime's own pinned memory: kernels write here while the host reads -- see big_host_alloc in common.h)
    BCE_HIP_TRY(c, hipHostMalloc(&p, want, hipHostMallocCoherent | hipHostMallocMapped));   // the wave tail kernel reads answers written while it runs
    c->pin_s += now_s() - t0; c->pin_bytes += want; c->pin_calls++;
    cap = want;
    return BCE_HIP_OK;
  }
  ~Pinned() {
    if (keep_p) { *keep_p = p; *keep_cap = cap; }
    else if (p) (void)hipHostFree(p);
  }
};

struct Mailbox {               // a few host-coherent words the wave tail kernel and the host exchange while the kernel runs
  uint32_t *host = nullptr, *dev = nullptr;
  int open(bce_hip_ctx *c) {
    BCE_HIP_TRY(c, hipHostMalloc(reinterpret_cast<void **>(&host), 64, hipHostMallocCoherent | hipHostMallocMapped));
    for (int i = 0; i < 16; ++i) host[i] = 0;
    BCE_HIP_TRY(c, hipHostGetDevicePointer(reinterpret_cast<void **>(&dev), host, 0));
    return BCE_HIP_OK;
  }
  ~Mailbox() { if (host) (void)hipHostFree(host); }
};

// The very deep tail on the host.  A chain of a few nodes that goes on for millions of rounds (a megabyte run, a
// whole-file duplicate) costs the wave kernel ~2 us per round -- the latency of one dependent load after the other
// -- while a CPU core does the same round out of its caches in ~0.1 us.  So once the wave kernel has spent
// max(kHostTailMin, n / 1000) rounds on <= 64 nodes, the boundary ranks (8 x 4(n+1) B) and the node lists come to the host, the rounds
// are finished here exactly as `bce -ds` runs them (decoder.cpp, BCE::code mode 0, bce.cpp:1246-1371, same decoders),
// and the ranks go back for the plane fill.  The copies are ~2 x 60 ms per 10^8 bytes.
__global__ void dec_scatter_kernel(uint32_t *__restrict__ R, const uint64_t *__restrict__ idx, const uint32_t *__restrict__ val, uint64_t m) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < m; i += (uint64_t)gridDim.x * blockDim.x) R[idx[i]] = val[i];
}

// Eight CPUs of the calling thread's L3 domain that its affinity mask allows (empty: fewer than eight, or no sysfs):
// where the eight threads of dec_host_tail sit so that their per-round barrier stays inside one CCX.
static std::vector<int> tail_cpus() {
  std::vector<int> ccx;
  cpu_set_t aff;
  if (getenv("BCE_DEC_TAIL_NOPIN") || sched_getaffinity(0, sizeof aff, &aff) != 0) return ccx;
  char path[128];
  snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", sched_getcpu());
  if (FILE *f = fopen(path, "r")) {
    char buf[512] = {0};
    if (fgets(buf, sizeof buf, f)) {
      for (char *q = buf; *q && ccx.size() < 8;) {
        char *e;
        long lo = strtol(q, &e, 10), hi = lo;
        if (e == q) break;
        if (*e == '-') { q = e + 1; hi = strtol(q, &e, 10); }
        for (long v = lo; v <= hi && ccx.size() < 8; ++v) if (v >= 0 && v < CPU_SETSIZE && CPU_ISSET((int)v, &aff)) ccx.push_back((int)v);
        if (*e != ',') break;
        q = e + 1;
      }
    }
    fclose(f);
  }
  if (ccx.size() != 8) ccx.clear();
  return ccx;
}

// The host tail's pinned copy of the boundary ranks is 32 (n + 1) bytes; pinning them takes ~0.15 s per GB the first time a
// context needs them (3.2 GB at 10^8 bytes: 0.5 s, a third of a whole decode).  Once a decode looks like it will end in
// a long tail (see the caller) the allocation is started on a thread of its own, beside the GPU's rounds.
struct BigPin {
  bce_hip_ctx *c = nullptr;
  std::thread th;
  void *p = nullptr;
  size_t bytes = 0;
  bool registered = false;
  bool running = false;
  double t_start = 0, t_done = 0;
  void start(bce_hip_ctx *ctx, size_t want) {
    if (running || ctx->h_big_cap >= want) return;
    c = ctx; bytes = want;
    const int dev = ctx->device;
    try {
      t_start = now_s();
      th = std::thread([this, dev] { p = big_host_alloc(bytes, dev, &registered); t_done = now_s(); });
      running = true;
    } catch (...) { running = false; }                        // (no thread: dec_host_tail allocates as before)
  }
  void settle() {                                              // the buffer, if it came, becomes the context's
    if (!running) return;
    const double t0 = now_s();
    th.join();
    running = false;
    if (getenv("BCE_DEC_TIMING")) fprintf(stderr, "gpu decode: pinned %.1f GB beside the rounds: %.3f s, the tail waited %.3f s of them\n", bytes / 1e9, t_done - t_start, now_s() - t0);
    if (p) {
      if (c->h_big) big_host_free(c, c->h_big, c->h_big_cap, c->h_big_registered);
      c->h_big = p; c->h_big_cap = bytes; c->h_big_registered = registered; p = nullptr;
      if (registered) c->reg_maps++;
    }
  }
  ~BigPin() { settle(); }
};

constexpr uint32_t kHostTailMin = 20000;                    // ... rounds, or n / 1000 if that is more (the copies cost ~n)

int dec_host_tail(bce_hip_ctx *c, const DecArgs &a, const DecCtl &ctl, std::vector<Decoder> &dec, uint32_t n, uint32_t *round,
                  uint64_t *nodes_total, uint64_t *queries_total, bool *bad_out) {
  const uint32_t par = *round & 1u;
  std::vector<Node> cur[8][2], nxt[8][2];
  BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (uint32_t p = 0; p < 8; ++p) {
    const uint32_t c0 = ctl.cnt[par][p][0], c1 = ctl.cnt[par][p][1];
    const Node *base = a.nodes + ((size_t)(par * 8u + p)) * a.capP;
    cur[p][0].resize(c0);
    if (c0) BCE_HIP_TRY(c, hipMemcpy(cur[p][0].data(), base, (size_t)c0 * sizeof(Node), hipMemcpyDeviceToHost));
    cur[p][1].resize(c1);
    if (c1) {
      BCE_HIP_TRY(c, hipMemcpy(cur[p][1].data(), base + (a.capP - c1), (size_t)c1 * sizeof(Node), hipMemcpyDeviceToHost));
      for (uint32_t i = 0; i < c1 / 2; ++i) std::swap(cur[p][1][i], cur[p][1][c1 - 1u - i]);     // child1 lists grow downwards
    }
  }
  const size_t stride = (size_t)n + 1;
  // 3.2 GB at 10^8 bytes: into pinned memory the context keeps (a copy into fresh pageable memory ran at a fifth of the bus)
  const double tcp0 = now_s();
  if (c->h_big_cap < 8 * stride * 4) {
    if (c->h_big) { big_host_free(c, c->h_big, c->h_big_cap, c->h_big_registered); c->h_big = nullptr; c->h_big_cap = 0; }
    c->h_big = big_host_alloc(8 * stride * 4, c->device, &c->h_big_registered);
    if (c->h_big && c->h_big_registered) c->reg_maps++;
    if (!c->h_big) { snprintf(c->err, sizeof c->err, "host tail: no pinned memory for the boundary ranks"); return BCE_HIP_E_NOMEM; }
    c->h_big_cap = 8 * stride * 4;
  }
  uint32_t *Rh = static_cast<uint32_t *>(c->h_big);
  const double tcp1 = now_s();
  BCE_HIP_TRY(c, hipMemcpyAsync(Rh, a.R, 8 * stride * 4, hipMemcpyDeviceToHost, c->stream));
  BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (getenv("BCE_DEC_TIMING")) fprintf(stderr, "gpu decode: host tail: pinned buffer %.3f s, boundary ranks to the host %.3f s (%.1f GB)\n", tcp1 - tcp0, now_s() - tcp1, 8 * stride * 4 / 1e9);
  // what the host learns goes back as (index, value) pairs, kept per plane (the planes are worked on side by side)
  // (each plane's output on cache lines of its own: the headers of neighbouring std::vectors share a line, and eight threads
  //  appending to neighbouring vectors took 13 us per round for what is 3 us of work)
  struct alignas(128) PlaneOut {
    std::vector<Node> o0, o1;                  // the children: next round's lists of plane i + 1
    std::vector<uint64_t> wi;
    std::vector<uint32_t> wv;
    uint64_t nodes = 0, queries = 0;
  };
  std::unique_ptr<PlaneOut[]> po(new PlaneOut[8]);
  // beyond that many pairs (per plane) the whole array goes back: the pinned copy runs at the bus' rate (3.2 GB in 56 ms),
  // while a pair costs two appends here, a place in two pageable arrays and a scattered store there (~7 ns: 18 M pairs of
  // the natural corpus took longer than the whole array)
  const size_t pairs_max = std::min<size_t>(stride / 8 + 1, (size_t)1 << 18);
  std::atomic<bool> bad_flag{false};
  // One plane of one round (BCE::code mode 0, bce.cpp:1261-1351): reads and writes plane i's boundary ranks only, appends to the
  // lists of plane i + 1 only -- the planes of a round are independent, which is the reference's own OpenMP split (:1250-1252).
  auto do_plane = [&](uint32_t i) {
    uint32_t *R = Rh + (size_t)i * stride;
    const uint32_t zi = a.zeros[i];
    PlaneOut &P = po[i];
    // a child is plane i + 1's node in the NEXT round: its three boundary ranks are asked for now (two random lines of a
    // 4(n + 1)-byte array: DRAM), so that they are in the cache (this core's, or the L3 it shares with the plane's thread)
    // by the time the round gets there -- in the chains a round is ~10 nodes and was mostly waiting for these lines
    const uint32_t *Rn = Rh + (size_t)((i + 1u) & 7u) * stride;
    auto ask = [&](uint32_t cs, uint32_t cx0, uint32_t cx1) {
      __builtin_prefetch(Rn + cs, 0, 3);
      __builtin_prefetch(Rn + cs + cx0 + cx1, 0, 3);
      __builtin_prefetch(Rn + cs + cx0, 1, 3);
    };
    std::vector<Node> &o0 = P.o0, &o1 = P.o1;
    std::vector<uint64_t> &wi = P.wi;
    std::vector<uint32_t> &wv = P.wv;
    uint64_t nn = 0, qq = 0;
    for (int j = 0; j < 2; ++j)
      for (const Node &nd : cur[i][j]) {
        const uint32_t s = nd.s, x0 = nd.x0, x1 = nd.x1, x = x0 + x1;
        if ((uint64_t)s + x > n || R[s] == kUnknown || R[s + x] == kUnknown) { bad_flag.store(true); return; }
        const uint32_t s1 = R[s], n1x = R[s + x] - s1, s0 = s - s1;
        if (n1x > x) { bad_flag.store(true); return; }
        uint32_t n1x0;
        if (!n1x) { o0.push_back(Node{s0, x0, x1}); ask(s0, x0, x1); n1x0 = 0; }
        else if (n1x == x) { o1.push_back(Node{zi + s1, x0, x1}); ask(zi + s1, x0, x1); n1x0 = x0; }
        else {
          const uint32_t n0x = x - n1x;
          uint32_t mn = x0 - n1x, mx = n1x - x1;
          mn = ((int32_t)mn < 0) ? 0u : mn;
          mx = ((int32_t)mx < 0) ? 0u : mx;
          mx = x0 - mx;
          uint32_t n0x0 = mn;
          if (mx != mn) { n0x0 = mn + dec[i].get_adaptive(mx - mn + 1, n0x, x1, x); ++qq; }
          if (n0x0 > mx) { bad_flag.store(true); return; }
          const uint32_t n0x1 = n0x - n0x0;
          if (n0x0 && n0x1) { o0.push_back(Node{s0, n0x0, n0x1}); ask(s0, n0x0, n0x1); }
          const uint32_t n1x1 = x1 - n0x1;
          n1x0 = n1x - n1x1;
          if (n1x0 && n1x1) { o1.push_back(Node{zi + s1, n1x0, n1x1}); ask(zi + s1, n1x0, n1x1); }
        }
        R[s + x0] = s1 + n1x0;
        if (wi.size() <= pairs_max) { wi.push_back((uint64_t)i * stride + s + x0); wv.push_back(s1 + n1x0); }
        ++nn;
      }
    P.nodes += nn; P.queries += qq;
  };
  // Rounds of a few nodes run on this thread (a round of two nodes is ~0.2 us: no barrier is worth that); from kParMin nodes
  // on the eight planes go to eight threads -- this one and seven helpers that spin on the round number (and doze off when
  // nothing has come for a while).  The GPU's resident tail kernels managed 35-160 ns per node on such rounds (dependent
  // loads, a mailbox round trip per query round); a host core does a node in ~40-100 ns, eight of them side by side.
  constexpr size_t kParMin = 16;              // (a round on eight threads costs ~1 us of barrier; a node ~0.1 us)
  constexpr uint32_t kParkAfter = 4096;       // serial rounds in a row after which the helpers stop spinning
  struct Pool {
    std::atomic<uint64_t> epoch{0};
    std::atomic<uint32_t> done{0};
    std::atomic<bool> quit{false}, parked{false};
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::thread> th;
  } pool;
  bool threaded = !getenv("BCE_DEC_TAIL_SERIAL") && std::thread::hardware_concurrency() >= 8u;
  // The eight threads meet at a barrier every round: on cores that share an L3 (one CCX of the EPYC hosts) that is ~1 us, across
  // the package 4-5 us -- more than the round's work.  So for the duration of the tail they sit on eight CPUs of this
  // thread's L3 domain, if the affinity mask has that many (otherwise wherever the scheduler puts them).
  std::vector<int> ccx;
  cpu_set_t old_aff;
  bool repin = false;
  if (threaded && sched_getaffinity(0, sizeof old_aff, &old_aff) == 0) {
    ccx = tail_cpus();
    if (ccx.size() == 8) {
      cpu_set_t one; CPU_ZERO(&one); CPU_SET(ccx[0], &one);
      repin = sched_setaffinity(0, sizeof one, &one) == 0;
    }
  }
  struct Unpin { bool on; cpu_set_t *aff; ~Unpin() { if (on) (void)sched_setaffinity(0, sizeof *aff, aff); } } unpin{repin, &old_aff};
  if (threaded) try {
    for (uint32_t w = 1; w < 8; ++w)
      pool.th.emplace_back([&, w] {
        if (repin) { cpu_set_t one; CPU_ZERO(&one); CPU_SET(ccx[w], &one); (void)sched_setaffinity(0, sizeof one, &one); }
        uint64_t seen = 0;
        uint32_t idle = 0;
        for (;;) {
          const uint64_t e = pool.epoch.load(std::memory_order_acquire);
          if (e != seen) {
            seen = e;
            do_plane(w);
            pool.done.fetch_add(1, std::memory_order_release);
            idle = 0;
            continue;
          }
          if (pool.quit.load(std::memory_order_acquire)) return;
          if (pool.parked.load(std::memory_order_acquire)) {             // the chains have begun: sleep until rounds get wide again
            std::unique_lock<std::mutex> lk(pool.mu);
            pool.cv.wait(lk, [&] { return !pool.parked.load() || pool.quit.load(); });
            continue;
          }
          (void)idle;
          __builtin_ia32_pause();
        }
      });
  } catch (...) {                                                          // no threads to be had: the rounds run on this one
    { std::lock_guard<std::mutex> lk(pool.mu); pool.quit.store(true, std::memory_order_release); }
    pool.cv.notify_all();
    for (auto &t : pool.th) t.join();
    pool.th.clear();
    threaded = false;
  }
  uint64_t par_rounds = 0, par_nodes = 0, ser_nodes = 0;
  double par_time = 0, ser_time = 0;
  const bool tail_timing = getenv("BCE_DEC_TIMING") != nullptr;
  uint32_t rounds = 0, serial_run = 0;
  auto unpark = [&] { { std::lock_guard<std::mutex> lk(pool.mu); pool.parked.store(false); } pool.cv.notify_all(); };
  // `live`: the planes that have nodes this round.  The chains of the deep tail are a node or two in one or two planes for
  // hundreds of thousands of rounds: such a round touches those planes' lists only (a round over all eight planes' sixteen
  // lists, empty or not, cost more than its two nodes).
  uint32_t live = 0;
  for (uint32_t i = 0; i < 8; ++i) if (!cur[i][0].empty() || !cur[i][1].empty()) live |= 1u << i;
  while (live && !bad_flag.load(std::memory_order_relaxed)) {
    size_t tot = 0;
    for (uint32_t m = live; m; m &= m - 1) { const int i = __builtin_ctz(m); tot += cur[i][0].size() + cur[i][1].size(); }
    const bool par_round = threaded && tot >= kParMin;
    const double tr0 = tail_timing && (par_round || (rounds & 63u) == 0) ? now_s() : 0.0;
    if (par_round) {
      if (pool.parked.load(std::memory_order_relaxed)) unpark();
      serial_run = 0;
      pool.done.store(0, std::memory_order_relaxed);
      pool.epoch.fetch_add(1, std::memory_order_release);
      do_plane(0);
      while (pool.done.load(std::memory_order_acquire) < 7u) __builtin_ia32_pause();
      ++par_rounds;
    } else {
      for (uint32_t m = live; m && !bad_flag.load(std::memory_order_relaxed); m &= m - 1) do_plane((uint32_t)__builtin_ctz(m));
      if (threaded && ++serial_run == kParkAfter) pool.parked.store(true, std::memory_order_release);
    }
    if (tail_timing) {                                                     // (serial rounds: one in 64 is timed -- the clock costs as much as the round)
      if (par_round) { par_time += now_s() - tr0; par_nodes += tot; }
      else { if ((rounds & 63u) == 0) ser_time += 64.0 * (now_s() - tr0); ser_nodes += tot; }
    }
    ++rounds;
    uint32_t next_live = 0;
    for (uint32_t m = live; m; m &= m - 1) {                               // plane i's children are plane i + 1's nodes
      const int i = __builtin_ctz(m), q = (i + 1) & 7;
      cur[q][0].swap(po[i].o0); po[i].o0.clear();
      cur[q][1].swap(po[i].o1); po[i].o1.clear();
      if (!cur[q][0].empty() || !cur[q][1].empty()) next_live |= 1u << q;
    }
    for (uint32_t m = live; m; m &= m - 1) {                               // a plane that was worked on and got nothing new
      const int q = __builtin_ctz(m);
      if (!((live >> ((q + 7) & 7)) & 1u)) { cur[q][0].clear(); cur[q][1].clear(); }
    }
    live = next_live;
  }
  { std::lock_guard<std::mutex> lk(pool.mu); pool.quit.store(true, std::memory_order_release); }
  pool.cv.notify_all();
  for (auto &t : pool.th) t.join();
  const bool bad = bad_flag.load();
  uint64_t nodes = 0, queries = 0;
  bool whole = false;
  std::vector<uint64_t> widx;
  std::vector<uint32_t> wval;
  for (int i = 0; i < 8; ++i) {
    nodes += po[i].nodes; queries += po[i].queries;
    if (po[i].wi.size() > pairs_max) whole = true;
  }
  if (!whole)
    for (int i = 0; i < 8; ++i) { widx.insert(widx.end(), po[i].wi.begin(), po[i].wi.end()); wval.insert(wval.end(), po[i].wv.begin(), po[i].wv.end()); }
  if (tail_timing) fprintf(stderr, "gpu decode: host tail: %llu of %u rounds on eight threads (%llu nodes, %.3f s), the others on one (%llu nodes, %.3f s)\n",
                           (unsigned long long)par_rounds, rounds, (unsigned long long)par_nodes, par_time, (unsigned long long)ser_nodes, ser_time);
  *bad_out = bad;
  if (bad) return BCE_HIP_OK;
  const double tcp2 = now_s();
  if (getenv("BCE_DEC_TIMING")) fprintf(stderr, "gpu decode: host tail: %u rounds, %llu nodes in %.3f s\n", rounds, (unsigned long long)nodes, tcp2 - tcp0);
  if (whole) {                                                           // (more than an eighth of everything: the whole array)
    BCE_HIP_TRY(c, hipMemcpyAsync(a.R, Rh, 8 * stride * 4, hipMemcpyHostToDevice, c->stream));
    BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
  } else if (!widx.empty()) {
    const size_t m = widx.size();
    BCE_TRY(ensure(c, c->skey[0], m * 8));
    BCE_TRY(ensure(c, c->skey[1], m * 4));
    BCE_HIP_TRY(c, hipMemcpy(c->skey[0].p, widx.data(), m * 8, hipMemcpyHostToDevice));
    BCE_HIP_TRY(c, hipMemcpy(c->skey[1].p, wval.data(), m * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(dec_scatter_kernel, dim3((unsigned)((m + 255) / 256 < 4096 ? (m + 255) / 256 : 4096)), dim3(256), 0, c->stream,
                       a.R, c->skey[0].as<uint64_t>(), c->skey[1].as<uint32_t>(), (uint64_t)m);
    BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
    BCE_HIP_TRY(c, hipGetLastError());
  }
  *round += rounds; *nodes_total += nodes; *queries_total += queries;
  return BCE_HIP_OK;
}

// The node lists of a decode: at most n / 2 + 2 nodes each (the worst case: disjoint intervals of width >= 2), and to begin with an
// eighth of n (what k3_begin starts the encoder with: text fills 0.02-0.03 n, random bytes 0.15-0.3 n) or what the context's
// buffer already holds.  A round whose children would not fit stops the decode before it writes them (the scan of the
// children pass checks; the one-launch kernel, which checks afterwards, is only used where twice the round's nodes fit) and
// the decode starts again with twice the room (decompress_device_body).  Test knob 12 / BCE_HIP_CAPP_DIV as for the encoder.
uint32_t dec_full_capP(uint32_t n) { return (uint32_t)((uint64_t)n / 2 + 2); }
uint32_t dec_capP(const bce_hip_ctx *c, uint32_t n, size_t archive_bytes) {
  const uint64_t full = dec_full_capP(n);
  uint64_t div = 8;
  bool forced = false;
  if (c->dbg_capp_div) { div = c->dbg_capp_div; forced = true; }
  if (const char *e = getenv("BCE_HIP_CAPP_DIV")) { const uint64_t v = strtoull(e, nullptr, 10); if (v >= 1) { div = v; forced = true; } }
  uint64_t cap = (uint64_t)n / div + 4096;
  // (the one-launch rounds need twice a round's nodes to fit: 4 M nodes per list at least, i.e. the worst case up to 8 MB of
  //  input -- with n / 8 a 1 MB input left them for its widest rounds and decoded in 32 ms instead of 16)
  if (!forced && cap < ((uint64_t)4 << 20)) cap = (uint64_t)4 << 20;
  // How full the lists get goes with how well the input compresses -- text (archive = 0.23 n) fills 0.03 n per list, random
  // bytes (1.0 n) 0.2-0.3 n -- and the archive's size is known before the first round: 0.25 x archive bytes nodes per list
  // spares a high-entropy archive the decodes that run out of room and start again (1.5 * 10^9 random bytes: one of 40 s).
  if (!forced) { const uint64_t by_ratio = (uint64_t)(0.25 * (double)archive_bytes) + 4096; if (by_ratio > cap) cap = by_ratio; }
  const uint64_t held = c->nlist[0].cap / (16 * sizeof(Node));
  if (held > cap && !forced) cap = held;
  if (c->dec_cap_next > cap) cap = c->dec_cap_next;             // (a decode that ran out of room: this much the next time)
  return (uint32_t)(cap < full ? cap : full);
}

}  // namespace

}  // namespace bce

using namespace bce;

// `bce -d` on the GPU: archive -> original bytes (see the header of this file).  The context is only used for its
// device, stream and scratch buffers; any compression state in it is dropped.
static int decompress_device_body(bce_hip_ctx *c, const uint8_t *archive, size_t len, uint8_t *out, size_t cap, size_t *out_len);
extern "C" int bce_hip_decompress_device(bce_hip_ctx *c, const uint8_t *archive, size_t len, uint8_t *out, size_t cap,
                                         size_t *out_len) {
  return bce_guarded(c, [&] { return decompress_device_body(c, archive, len, out, cap, out_len); });
}
static int decompress_device_once(bce_hip_ctx *c, const uint8_t *archive, size_t len, uint8_t *out, size_t cap, size_t *out_len);
static int decompress_device_body(bce_hip_ctx *c, const uint8_t *archive, size_t len, uint8_t *out, size_t cap, size_t *out_len) {
  if (!c) return BCE_HIP_E_ARG;
  c->dec_cap_next = 0;
  for (;;) {
    c->dec_list_overflow = false;
    const int rc = decompress_device_once(c, archive, len, out, cap, out_len);
    if (rc != BCE_HIP_E_OVERFLOW || !c->dec_list_overflow) { c->dec_cap_next = 0; return rc; }
    // a round did not fit the node lists (nothing of it was written): the same decode again with twice the room
    const uint32_t full = dec_full_capP((uint32_t)*out_len);
    if (c->capP >= full) { c->dec_cap_next = 0; return rc; }
    c->dec_cap_next = (uint64_t)c->capP * 2 < full ? (uint64_t)c->capP * 2 : full;
    c->dec_restarts++;
    if (getenv("BCE_DEC_TIMING") || getenv("BCE_ALLOC_TRACE")) fprintf(stderr, "gpu decode: a round does not fit the node lists of %u nodes: once more with %llu\n", c->capP, (unsigned long long)c->dec_cap_next);
    (void)hipDeviceSynchronize();
    c->err[0] = 0;
  }
}
static int decompress_device_once(bce_hip_ctx *c, const uint8_t *archive, size_t len, uint8_t *out, size_t cap, size_t *out_len) {
  if (!c || !archive || !out_len) return BCE_HIP_E_ARG;
  ArchiveHead hd;
  if (parse_archive(archive, len, hd, /*header_only=*/true) != 0) return BCE_HIP_E_ARG;
  *out_len = hd.n;
  if (!out) return BCE_HIP_OK;
  if (cap < hd.n) return BCE_HIP_E_OVERFLOW;
  if (parse_archive(archive, len, hd, false) != 0) return BCE_HIP_E_ARG;
  BCE_HIP_TRY(c, hipSetDevice(c->device));
  c->coder->drain();
  c->stage = 0; c->enum_active = false; c->k1_valid = false;   // the scratch buffers below belong to the decoder now
  c->phase = 4;
  struct PhaseEnd { bce_hip_ctx *c; ~PhaseEnd() { c->phase = 0; } } phase_end{c};
  const uint32_t n = hd.n;
  const bool timing = getenv("BCE_DEC_TIMING") != nullptr;
  double tp0 = now_s();

  // ---- buffers ----
  const size_t rstride = (size_t)n + 1;
  c->capP = dec_capP(c, n, len);
  BCE_TRY(ensure(c, c->nlist[0], (size_t)16 * c->capP * sizeof(Node)));
  BCE_TRY(ensure(c, c->ctl, sizeof(DecCtl) > sizeof(EnumCtl) ? sizeof(DecCtl) : sizeof(EnumCtl)));
  const size_t max_tiles = (size_t)8 * ((c->capP + K3_TILE - 1) / K3_TILE + 1);
  BCE_TRY(ensure(c, c->tilecnt, max_tiles * 16));
  BCE_TRY(ensure(c, c->tileoff, max_tiles * 16));
  DevBuf &Rbuf = c->dfs;                                        // 8 x (n + 1) boundary ranks
  BCE_TRY(ensure(c, Rbuf, 8 * rstride * 4));
  uint32_t *R = Rbuf.as<uint32_t>();
  BCE_HIP_TRY(c, hipMemsetAsync(R, 0xFF, 8 * rstride * 4, c->stream));
  DecCtl ctl;
  memset(&ctl, 0, sizeof ctl);
  for (int i = 0; i < 8; ++i) {
    const uint32_t zero = 0, ones = n - hd.C[i];               // R[p][0] = 0, R[(i+7)&7][n] = n - C[i]  (:1207-1211)
    BCE_HIP_TRY(c, hipMemcpyAsync(R + (size_t)i * rstride, &zero, 4, hipMemcpyHostToDevice, c->stream));
    BCE_HIP_TRY(c, hipMemcpyAsync(R + (size_t)((i + 7) & 7) * rstride + n, &ones, 4, hipMemcpyHostToDevice, c->stream));
    BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));            // stack temporaries
    if (hd.C[i] && n - hd.C[i]) {                               // :1214-1216
      const Node root = {0u, hd.C[i], n - hd.C[i]};
      BCE_HIP_TRY(c, hipMemcpyAsync(c->nlist[0].as<Node>() + (size_t)i * c->capP, &root, sizeof root, hipMemcpyHostToDevice, c->stream));
      BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
      ctl.cnt[0][i][0] = 1;
    }
  }
  BCE_HIP_TRY(c, hipMemcpyAsync(c->ctl.p, &ctl, sizeof ctl, hipMemcpyHostToDevice, c->stream));
  BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
  // (query, escape-record and answer buffers stay with the context: pinning their ~150 MB anew was 0.05 s of every decode)
  Pinned pin_info, pin_q(&c->dec_pin[0], &c->dec_pin_cap[0]), pin_e(&c->dec_pin[1], &c->dec_pin_cap[1]), pin_res(&c->dec_pin[2], &c->dec_pin_cap[2]);
  struct Events {
    hipEvent_t e[8] = {};
    ~Events() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); }
  } ev;
  uint32_t prev_qtot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t split_rounds = 0;
  double dbg_r[5] = {0, 0, 0, 0, 0};
  uint32_t dbg_last[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double t_split = 0, ts_issue = 0, ts_first = 0, ts_lanes = 0, ts_wait = 0, ts_rb = 0;
  const bool no_split = getenv("BCE_DEC_NO_SPLIT") != nullptr;
  BCE_TRY(pin_info.ensure(c, sizeof(DecInfo), 4096));          // (a few words: not the 16 MB the query buffers start with -- 4 ms of pinning per decode)
  DecInfo *info = static_cast<DecInfo *>(pin_info.p);
  memset(info, 0, sizeof *info);

  DecArgs a;
  a.ctl = c->ctl.as<DecCtl>();
  a.info = info;                                                // pinned host memory is device-accessible at the same address
  a.nodes = c->nlist[0].as<Node>();
  a.R = R;
  a.tilecnt = c->tilecnt.as<uint32_t>();
  a.tileoff = c->tileoff.as<uint32_t>();
  a.capP = c->capP; a.n = n;
  for (int p = 0; p < 8; ++p) a.zeros[p] = hd.C[(p + 1) & 7];
  {
    PlaneCfg hcfg[8];
    for (int p = 0; p < 8; ++p) hcfg[p] = hd.dec[p].cfg;
    BCE_TRY(ensure(c, c->dcfg, sizeof hcfg));
    BCE_HIP_TRY(c, hipMemcpy(c->dcfg.p, hcfg, sizeof hcfg, hipMemcpyHostToDevice));
    a.cfg = c->dcfg.as<PlaneCfg>();
  }
  DevBuf &Qbuf = c->skey[0], &Ebuf = c->skey[1], &Rsbuf = c->sesc;   // queries / escape queries / answers of a round (device side)
  QueryPool pool;
  pool.dec = &hd.dec;
  pool.start();

  if (timing) { fprintf(stderr, "gpu decode: setup %.3f s\n", now_s() - tp0); tp0 = now_s(); }
  // ---- the rounds (BCE::code mode 0, :1246-1371) ----
  uint64_t cur_nodes = 0;
  for (int i = 0; i < 8; ++i) cur_nodes += ctl.cnt[0][i][0];
  uint32_t round = 0;
  uint64_t nodes_total = 0, queries_total = 0;
  double t_q = 0, t_copy = 0, t_host = 0, t_c = 0;
  uint64_t tail_rounds = 0;
  bool answered_pending = false;                              // pin_res holds the answers of the round about to run
  Mailbox mbox;                                               // of the wave tail kernel (see DecArgs::mbox)
  if (!getenv("BCE_DEC_NO_MAILBOX")) BCE_TRY(mbox.open(c));
  uint32_t next_seq = 1, last_answered = 0;
  const bool host_tail_ok = !getenv("BCE_DEC_NO_HOST_TAIL");
  // (ski rental: the copies cost ~1.2 ns per input byte, a round ~1.0 us less on the host than in the wave kernel, so the
  //  switch pays once n / 830 rounds are still to come -- which nobody knows -- and is made after that many have gone by)
  const uint32_t kHostTailAfter = n / 2500u > kHostTailMin ? n / 2500u : kHostTailMin;
  // Query-heavy tails (executables: something is coded in half of the rounds) go to the host as a whole: a query round costs the
  // resident kernels a mailbox round trip (~10 us), the host -- eight threads, one per plane -- nothing.  Whether a tail is
  // query-heavy is measured: the first kProbeRounds rounds of the tail kernels count their mailbox rounds.
  constexpr uint32_t kProbeRounds = 1024;
  uint32_t probe_rounds = 0;
  uint64_t probe_mbox0 = 0;
  bool probe_done = getenv("BCE_DEC_NO_PROBE") != nullptr, query_heavy = false;
  if (getenv("BCE_DEC_FORCE_HOST_TAIL")) { probe_done = true; query_heavy = true; }
  // With eight CPUs on one L3 for its threads the host does a node of a tail round in ~20 ns (measured: 18.8 M nodes of the
  // binary corpus in 0.37 s, against 1.4 s in the resident kernels; the natural corpus' 20 M nodes in 0.6 s against 1.1 s):
  // then every tail that is worth the copy goes there and nothing is probed.
  // ... every LONG tail, that is: the copy and the threads cost ~0.1 s, which a tail of a few hundred rounds (text) does not have
  // to spare -- so the resident kernels always get the first kProbeRounds rounds.
  const bool host_has_ccx = std::thread::hardware_concurrency() >= 8u && !getenv("BCE_DEC_TAIL_SERIAL") && tail_cpus().size() == 8;
  uint64_t wide_hist[32] = {0}, wide_nodes[32] = {0}, small_rounds = 0;
  double wide_time[32] = {0};
  constexpr uint64_t kDirectNodes = 1u << 18;
  const bool no_small = getenv("BCE_DEC_NO_SMALL") != nullptr;
  BCE_TRY(ensure(c, c->smwords, (size_t)DS_MAXTILES * 8));
  BCE_HIP_TRY(c, hipMemsetAsync(c->smwords.p, 0, (size_t)DS_MAXTILES * 8, c->stream));      // pass tags of an earlier decode
  a.words = c->smwords.as<unsigned long long>();
  uint64_t mbox_rounds = 0, launches_wave = 0, launches_wg = 0, rounds_wg = 0, nodes_wave = 0, nodes_wg = 0, rounds_wave = 0;
  double t_wave = 0, t_wg = 0;
  BCE_TRY(ensure(c, c->runs, 64));
  uint32_t *d_rounds = c->runs.as<uint32_t>();
  // Rounds of a few thousand nodes cost the GPU ~40 us each (two launches, two syncs) whatever they hold; eight host threads
  // do 8192 nodes in about that time and 1024 in a tenth of it.  Where the host takes the tail anyway (a CCX for its
  // threads) it takes it from here on, provided the tail is LONG -- at least 512 more rounds at the present width, and worth
  // the copy of the ranks: text's few hundred tail rounds stay on the device.
  BigPin big_pin;
  uint32_t host_enter = 8192;
  if (const char *e = getenv("BCE_DEC_HOST_ENTER")) host_enter = (uint32_t)strtoul(e, nullptr, 10);
  while (cur_nodes) {
    double t0 = now_s();
    a.par = round & 1u;
    {
      const uint64_t left = 8ull * (n - 1u) - nodes_total;
      // past the half-way mark with 64 rounds' worth of the present width still to come: the widths of text fall
      // geometrically from their peak (the ratio stays under ~50 until next to nothing is left), a long tail does not
      if (host_tail_ok && host_has_ccx && !big_pin.running && left <= 4ull * (n - 1u) && left >= (n >> 6) + (1u << 18) &&
          left >= 64ull * cur_nodes && !getenv("BCE_DEC_NO_EARLY_PIN"))
        big_pin.start(c, 8 * ((size_t)n + 1) * 4);
      if (host_tail_ok && host_has_ccx && !answered_pending && !getenv("BCE_DEC_NO_TAIL") && cur_nodes <= host_enter &&
          left >= (n >> 6) + (1u << 18) && left >= 512ull * cur_nodes && left <= 8ull * (n - 1u) / 8u) {
        bool bad = false;
        const double th = now_s();
        const uint32_t r0 = round;
        big_pin.settle();
        BCE_TRY(dec_host_tail(c, a, ctl, hd.dec, n, &round, &nodes_total, &queries_total, &bad));
        if (bad) { snprintf(c->err, sizeof c->err, "decode: inconsistent archive (round %u)", round); return BCE_HIP_E_INTERNAL; }
        if (timing) fprintf(stderr, "gpu decode: %u rounds of the tail on the host from %llu nodes a round on, %.3f s with the copies\n", round - r0, (unsigned long long)cur_nodes, now_s() - th);
        t_c += now_s() - t0;
        cur_nodes = 0;
        break;
      }
    }
    if (cur_nodes <= DT_ENTER && !getenv("BCE_DEC_NO_TAIL")) {
      // Few nodes: the tail kernels run the forced rounds on the device; at a round with queries the workgroup kernel
      // emits them straight into pinned memory, the host answers (inline: they are few) and the kernel resumes with
      // the answers -- one launch and one sync per query round.
      BCE_TRY(pin_q.ensure(c, (size_t)(DT_CAP + 16) * 4));
      BCE_TRY(pin_e.ensure(c, (size_t)(DT_CAP + 16) * sizeof(uint4)));
      BCE_TRY(pin_res.ensure(c, (size_t)(DT_CAP + 16) * 4));
      DecArgs at = a;
      at.Q = static_cast<uint32_t *>(pin_q.p);
      at.E = static_cast<uint4 *>(pin_e.p);
      at.res = static_cast<const uint32_t *>(pin_res.p);
      bool resume = false, force_wg = false;
      answered_pending = false;
      auto answer_round = [&]() -> uint64_t {                   // the queries in pin_q / pin_e -> answers in pin_res
        const DecInfo in = *info;
        uint64_t qt = 0;
        for (int p = 0; p < 8; ++p) {
          QueryPool::answer(hd.dec[p], at.Q + in.qbase[p], at.E + in.ebase[p], static_cast<uint32_t *>(pin_res.p) + in.qbase[p], in.qtot[p]);
          qt += in.qtot[p];
        }
        return qt;
      };
      for (;;) {
        uint32_t done[5] = {0, 0, 0, 0, 0};
        if (host_tail_ok && query_heavy && !resume && !answered_pending && cur_nodes && cur_nodes <= DT_CAP &&
            8ull * (n - 1u) - nodes_total >= (n >> 6) + (1u << 18) && 8ull * (n - 1u) - nodes_total <= 8ull * (n - 1u) / 8u) {
          // the whole tail on the host (see above): nothing of this round has been asked or answered yet
          bool bad = false;
          const double th = now_s();
          const uint32_t r0 = round;
          big_pin.settle();
        BCE_TRY(dec_host_tail(c, a, ctl, hd.dec, n, &round, &nodes_total, &queries_total, &bad));
          if (bad) { snprintf(c->err, sizeof c->err, "decode: inconsistent archive (round %u)", round); return BCE_HIP_E_INTERNAL; }
          if (timing) fprintf(stderr, "gpu decode: %u rounds of the tail on the host, %.3f s with the copies\n", round - r0, now_s() - th);
          cur_nodes = 0;
          break;
        }
        const bool wave = !force_wg && cur_nodes <= 64;
        at.par = round & 1u;
        const double t_launch = now_s();
        at.mbox = mbox.dev;
        at.seq_base = next_seq;
        if (at.mbox) __atomic_store_n(&mbox.host[2], 0u, __ATOMIC_RELEASE);
        const uint64_t all_nodes = 8ull * (n - 1u);
        const bool probing = host_tail_ok && !probe_done && all_nodes - nodes_total >= (n >> 6) + (1u << 18);   // (a tail worth a copy of the ranks)
        if (probing && probe_rounds == 0) probe_mbox0 = mbox_rounds;
        const uint32_t probe_left = kProbeRounds > probe_rounds ? kProbeRounds - probe_rounds : 1u;
        const uint32_t max_wave = !host_tail_ok ? (1u << 30) : (probing && probe_left < kHostTailAfter ? probe_left : kHostTailAfter);
        if (wave) hipLaunchKernelGGL(dec_tail64_kernel, dim3(1), dim3(64), 0, c->stream, at, max_wave, d_rounds, resume ? 1u : 0u);
        else hipLaunchKernelGGL(dec_tail_kernel, dim3(1), dim3(DT_T), 0, c->stream, at, probing ? probe_left : (1u << 30), d_rounds, (resume ? 1u : 0u) | 2u);
        if (at.mbox) {
          // (before the copies below are queued: a device-to-host copy into pageable memory blocks the host until the kernel is done)
          // the tail kernel stays resident over query rounds: serve its mailbox until it says it has left
          const double t_poll = now_s();
          for (uint32_t spins = 0;; ++spins) {
            if (__atomic_load_n(&mbox.host[0], __ATOMIC_ACQUIRE) == next_seq) {
              queries_total += answer_round();
              __atomic_store_n(&mbox.host[1], next_seq, __ATOMIC_RELEASE);
              last_answered = next_seq++;
              ++mbox_rounds;
              continue;
            }
            if (__atomic_load_n(&mbox.host[2], __ATOMIC_ACQUIRE)) break;
            if ((spins & 1023u) == 1023u && (hipStreamQuery(c->stream) != hipErrorNotReady || now_s() - t_poll > 30.0)) break;   // a failed launch never raises the flag
            __builtin_ia32_pause();
          }
        }
        BCE_TRY(read_back(c, done, d_rounds, 20));
        BCE_TRY(read_back(c, &ctl, c->ctl.p, sizeof ctl));
        BCE_HIP_TRY(c, hipGetLastError());
        if (getenv("BCE_DEC_TRACE")) fprintf(stderr, "tail: round %u wave %d resume %d -> done %u why %u next %u\n", round, (int)wave, (int)resume, done[0], done[1], ctl.next_nodes);
        round += done[0]; tail_rounds += done[0];
        if (wave) { ++launches_wave; rounds_wave += done[0]; nodes_wave += ctl.nodes_total - nodes_total; t_wave += now_s() - t_launch; }
        else { ++launches_wg; rounds_wg += done[0]; nodes_wg += ctl.nodes_total - nodes_total; t_wg += now_s() - t_launch; }
        cur_nodes = ctl.next_nodes;
        nodes_total = ctl.nodes_total;
        // a resumed launch that could not even start its round (children outgrow the kernel, or an inconsistency):
        // the decoders have ALREADY answered this round's queries -- whoever runs the round must reuse the answers
        const bool stuck = done[4] != 0 && done[1] >= 2;          // the round the kernel stopped at has been answered already
        resume = false;
        force_wg = false;
        if (done[1] == 1) {                                      // queries of round `round` are in pin_q / pin_e
          // (a wave kernel that gave up waiting may have been answered through the mailbox at the same moment)
          if (!(at.mbox && done[2] != 0 && done[2] == last_answered)) {
            queries_total += answer_round();
            if (at.mbox) { last_answered = done[2]; next_seq = done[2] + 1u; }
          }
          resume = true;
          continue;
        }
        if (wave && done[1] == 2 && cur_nodes <= DT_CAP) {       // the children outgrow one wave: the workgroup kernel
          force_wg = true;
          resume = stuck;                                        // ... with the answers, if this round already has them
          continue;
        }
        answered_pending = stuck;
        if (probing) {
          probe_rounds += done[0];
          if (probe_rounds >= kProbeRounds) { probe_done = true; query_heavy = host_has_ccx || (mbox_rounds - probe_mbox0) * 3u >= probe_rounds * 2u; /* two rounds in three ask the decoders something */ }
        }
        if (host_tail_ok && done[1] == 0 && !stuck && cur_nodes &&
            ((wave && done[0] >= kHostTailAfter && cur_nodes <= 64) || (query_heavy && cur_nodes <= DT_CAP))) {
          // a long chain of a few nodes: the rest of the rounds on the host (dec_host_tail)
          bool bad = false;
          const double th = now_s();
          const uint32_t r0 = round;
          big_pin.settle();
        BCE_TRY(dec_host_tail(c, a, ctl, hd.dec, n, &round, &nodes_total, &queries_total, &bad));
          if (bad) { snprintf(c->err, sizeof c->err, "decode: inconsistent archive (round %u)", round); return BCE_HIP_E_INTERNAL; }
          if (timing) fprintf(stderr, "gpu decode: %u rounds of the deep tail on the host, %.3f s with the copies\n", round - r0, now_s() - th);
          cur_nodes = 0;
          break;
        }
        if (done[1] == 0 && done[0] && cur_nodes && cur_nodes <= DT_ENTER) continue;   // handed back (few nodes again) or out of max_rounds
        break;                                                   // nothing left, too many nodes (2) or an inconsistent node (3)
      }
      a.par = round & 1u;
      t_c += now_s() - t0;
      t0 = now_s();
      if (c->progress) c->progress(nodes_total, 8ull * n, c->progress_user);
      if (!cur_nodes) break;
    }
    // Rounds of up to kDirectNodes nodes exchange their queries and answers through pinned host memory (as the tail kernels do):
    // the kernels write / read it over the bus, and the round is two launches and two syncs with no copy in between.
    // (dec_small_kernel writes the children before it knows their number: only where twice the round's nodes fit a list)
    const bool small_round = cur_nodes <= DS_MAXNODES && !no_small && 2ull * cur_nodes <= c->capP;
    const bool direct = small_round && cur_nodes <= kDirectNodes && !answered_pending;
    if (direct) {
      BCE_TRY(pin_q.ensure(c, (size_t)(cur_nodes + 16) * 4));
      BCE_TRY(pin_e.ensure(c, (size_t)(cur_nodes + 16) * sizeof(uint4)));
      BCE_TRY(pin_res.ensure(c, (size_t)(cur_nodes + 16) * 4));
      a.Q = static_cast<uint32_t *>(pin_q.p);
      a.E = static_cast<uint4 *>(pin_e.p);
      a.res = static_cast<const uint32_t *>(pin_res.p);
    } else {
      BCE_TRY(ensure(c, Qbuf, (size_t)(cur_nodes + 16) * 4));
      BCE_TRY(ensure(c, Ebuf, (size_t)(cur_nodes + 16) * sizeof(uint4)));
      BCE_TRY(ensure(c, Rsbuf, (size_t)(cur_nodes + 16) * 4));
      a.Q = Qbuf.as<uint32_t>();
      a.E = Ebuf.as<uint4>();
      a.res = Rsbuf.as<uint32_t>();
    }
    uint64_t want = (cur_nodes + K3_TILE - 1) / K3_TILE + 8;
    const uint32_t grid = (uint32_t)(want < 2048 ? want : 2048);
    uint32_t hb = 0;
    const double t_round0 = timing ? now_s() : 0.0;
    if (timing) { while ((2ull << hb) <= cur_nodes && hb < 31) ++hb; wide_hist[hb]++; wide_nodes[hb] += cur_nodes; }
    const bool small = small_round;                                // one launch per pass (dec_small_kernel)
    a.round = round;
    if (!small && !answered_pending && !no_split) {
      // A six-launch round plane by plane.  The round's time is the busiest planes' sequential decoders (text: planes 0
      // and 1 hold half of all queries) with the query passes in front of them and the children passes behind.  The planes
      // of a round do not meet (plane p's children are plane p + 1's nodes of the NEXT round), so each plane is a lane of
      // its own -- query pass, copy out, decoder, answers in, children pass -- and the lanes start in the order of their
      // node counts: the busiest decoder starts as soon as ITS queries are out and only its own children pass is
      // left when it is done; everything else runs beside it.
      const double ts0 = now_s();
      for (int i = 0; i < 8; ++i) if (!ev.e[i]) BCE_HIP_TRY(c, hipEventCreateWithFlags(&ev.e[i], hipEventDisableTiming));
      uint32_t ord[8] = {0, 1, 2, 3, 4, 5, 6, 7};
      // (the bulk of the queries moves from plane p to plane p + 1 with every round -- a node's children are the next plane's
      //  nodes -- and the busiest decoder with it: plane p is as busy as plane p - 1 was last round.  Node counts say less:
      //  how many of a plane's nodes are forced differs from plane to plane.)
      uint64_t weight[8];
      for (int q = 0; q < 8; ++q) weight[q] = ((uint64_t)prev_qtot[(q + 7) & 7] << 1) + ((ctl.cnt[round & 1u][q][0] + ctl.cnt[round & 1u][q][1]) ? 1u : 0u);
      std::stable_sort(ord, ord + 8, [&](uint32_t x, uint32_t y) { return weight[x] > weight[y]; });
      uint32_t order = 0;
      for (int i = 0; i < 8; ++i) order |= ord[i] << (4 * i);
      BCE_TRY(pin_q.ensure(c, (size_t)(cur_nodes + 16) * 4));      // (a plane's queries lie at its base: the bases are exact, the size is a bound)
      BCE_TRY(pin_res.ensure(c, (size_t)(cur_nodes + 16) * 4));
      a.order = order;
      a.final = 0;
      auto ask = [&](int i) -> int {                               // the query pass of lane i
        a.pmask = 1u << ord[i];
        a.final = 0;
        hipLaunchKernelGGL((dec_tiles_kernel<0>), dim3(grid), dim3(K3_T), 0, c->stream, a);
        hipLaunchKernelGGL((dec_scan_kernel<true>), dim3(8), dim3(1024), 0, c->stream, a);
        hipLaunchKernelGGL((dec_tiles_kernel<1>), dim3(grid), dim3(K3_T), 0, c->stream, a);
        BCE_HIP_TRY(c, hipEventRecord(ev.e[i], c->stream));
        return BCE_HIP_OK;
      };
      BCE_TRY(ask(0));                                             // (launching costs the host ~60 us a lane: the busiest decoder does not wait for all eight)
      BCE_TRY(ask(1));
      const double tsa = now_s();
      ts_issue += tsa - ts0;
      QueryPool::Job jobs[8] = {};
      struct Settle {                                              // no decoder outlives this round (its buffers, an early return)
        QueryPool &pool; uint32_t mask;
        ~Settle() { if (mask) pool.wait(mask); }
      } settle{pool, 0u};
      uint64_t qtotal = 0;
      uint32_t new_qtot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      uint32_t launched = 0, children_done = 0;                    // planes whose decoders run / whose children pass is queued
      auto children = [&](uint32_t p, bool last) -> int {
        if (jobs[p].cnt) BCE_HIP_TRY(c, hipMemcpyAsync(Rsbuf.as<uint32_t>() + (jobs[p].r - static_cast<uint32_t *>(pin_res.p)), jobs[p].r, (size_t)jobs[p].cnt * 4, hipMemcpyHostToDevice, c->stream));
        a.pmask = 1u << p;
        a.final = last ? 1u : 0u;
        hipLaunchKernelGGL((dec_tiles_kernel<2>), dim3(grid), dim3(K3_T), 0, c->stream, a);
        hipLaunchKernelGGL((dec_scan_kernel<false>), dim3(8), dim3(1024), 0, c->stream, a);
        hipLaunchKernelGGL((dec_tiles_kernel<3>), dim3(grid), dim3(K3_T), 0, c->stream, a);
        children_done |= 1u << p;
        return BCE_HIP_OK;
      };
      for (int i = 0; i < 8; ++i) {
        const uint32_t p = ord[i];
        if (i > 0 && i + 1 < 8) BCE_TRY(ask(i + 1));               // one lane ahead of the one being waited for
        BCE_HIP_TRY(c, hipEventSynchronize(ev.e[i]));
        if (i == 0) ts_first += now_s() - tsa;
        const DecInfo in = *info;                                  // (plane p's fields and those of the planes before it; the rest are being written)
        if (in.err == 2) { snprintf(c->err, sizeof c->err, "decode: node list overflow (capP=%u)", c->capP); c->dec_list_overflow = true; return BCE_HIP_E_OVERFLOW; }
        if (in.err) { snprintf(c->err, sizeof c->err, "decode: inconsistent archive (round %u)", round); return BCE_HIP_E_INTERNAL; }
        const uint32_t qn = in.qtot[p], en = in.etot[p], qb = in.qbase[p], eb = in.ebase[p];
        new_qtot[p] = qn;
        qtotal += qn;
        if ((size_t)eb + en + 1 > pin_e.cap / sizeof(uint4)) {     // escape records are few, their number is not known ahead: grow between decoders
          if (settle.mask) { pool.wait(settle.mask); }
          BCE_TRY(pin_e.ensure(c, 2 * ((size_t)eb + en + 1) * sizeof(uint4)));
        }
        if (qn) {
          BCE_HIP_TRY(c, hipMemcpyAsync(static_cast<uint32_t *>(pin_q.p) + qb, a.Q + qb, (size_t)qn * 4, hipMemcpyDeviceToHost, c->copy_stream));
          if (en) BCE_HIP_TRY(c, hipMemcpyAsync(static_cast<uint4 *>(pin_e.p) + eb, a.E + eb, (size_t)en * sizeof(uint4), hipMemcpyDeviceToHost, c->copy_stream));
          BCE_HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
          jobs[p] = QueryPool::Job{static_cast<const uint32_t *>(pin_q.p) + qb, static_cast<const uint4 *>(pin_e.p) + eb, static_cast<uint32_t *>(pin_res.p) + qb, qn};
          pool.run_async(jobs, 1u << p);
          settle.mask |= 1u << p;
          launched |= 1u << p;
        } else {
          jobs[p] = QueryPool::Job{nullptr, nullptr, static_cast<uint32_t *>(pin_res.p) + qb, 0u};
          BCE_TRY(children(p, i == 7 && (launched & ~children_done) == 0u));   // nothing to ask: its children pass at once (the round's last one only if nobody is out)
        }
        // decoders that have finished meanwhile: their children passes go out between the copies
        const uint32_t fin = pool.done(launched & ~children_done);
        for (uint32_t q = 0; q < 8; ++q)
          if ((fin >> q) & 1u) { settle.mask &= ~(1u << q); BCE_TRY(children(q, i == 7 && (children_done | (1u << q)) == 0xFFu)); }
      }
      const double tsb = now_s();
      ts_lanes += tsb - tsa;
      while (children_done != 0xFFu) {
        const uint32_t fin = pool.wait_any(launched & ~children_done);
        for (uint32_t q = 0; q < 8; ++q)
          if ((fin >> q) & 1u) { settle.mask &= ~(1u << q); BCE_TRY(children(q, (children_done | (1u << q)) == 0xFFu)); }
      }
      a.pmask = 0xFFu; a.order = 0x76543210u; a.final = 1u;
      const double tsc = now_s();
      if (timing) {
        double last = 0; int lastp = -1;
        for (int q = 0; q < 8; ++q) if (((launched >> q) & 1u) && pool.t_end[q] > last) { last = pool.t_end[q]; lastp = q; }
        if ((launched >> ord[0]) & 1u) { dbg_r[0] += pool.t_begin[ord[0]] - ts0; dbg_r[1] += pool.t_end[ord[0]] - ts0; }
        if (lastp >= 0) { dbg_r[2] += last - ts0; dbg_r[3] += pool.t_begin[lastp] - ts0; dbg_last[lastp]++; }
        dbg_r[4] += tsc - ts0;
      }
      ts_wait += tsc - tsb;
      BCE_TRY(read_back(c, &ctl, c->ctl.p, sizeof ctl));
      ts_rb += now_s() - tsc;
      BCE_HIP_TRY(c, hipGetLastError());
      if (ctl.err) {
        snprintf(c->err, sizeof c->err, ctl.err == 2 ? "decode: node list overflow (capP=%u)" : "decode: inconsistent archive (round %u)", ctl.err == 2 ? c->capP : round);
        c->dec_list_overflow = ctl.err == 2;
        return ctl.err == 2 ? BCE_HIP_E_OVERFLOW : BCE_HIP_E_INTERNAL;
      }
      for (int q = 0; q < 8; ++q) prev_qtot[q] = new_qtot[q];
      ++split_rounds;
      t_split += now_s() - ts0;
      if (timing) wide_time[hb] += now_s() - t_round0;
      cur_nodes = ctl.next_nodes;
      nodes_total = ctl.nodes_total;
      if (c->progress) c->progress(nodes_total, 8ull * n, c->progress_user);
      queries_total += qtotal;
      ++round;
      continue;
    }
    if (small) {
      hipLaunchKernelGGL((dec_small_kernel<true>), dim3(grid), dim3(K3_T), 0, c->stream, a);
      ++small_rounds;
    } else {
      hipLaunchKernelGGL((dec_tiles_kernel<0>), dim3(grid), dim3(K3_T), 0, c->stream, a);
      hipLaunchKernelGGL((dec_scan_kernel<true>), dim3(8), dim3(1024), 0, c->stream, a);
      hipLaunchKernelGGL((dec_tiles_kernel<1>), dim3(grid), dim3(K3_T), 0, c->stream, a);
    }
    BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
    BCE_HIP_TRY(c, hipGetLastError());
    const DecInfo in = *info;
    { const double t1 = now_s(); t_q += t1 - t0; t0 = t1; }
    if (in.err == 2) { snprintf(c->err, sizeof c->err, "decode: node list overflow (capP=%u)", c->capP); c->dec_list_overflow = true; return BCE_HIP_E_OVERFLOW; }
    if (in.err) { snprintf(c->err, sizeof c->err, "decode: inconsistent archive (round %u)", round); return BCE_HIP_E_INTERNAL; }
    uint64_t qtotal = 0, etotal = 0;
    for (int p = 0; p < 8; ++p) { qtotal += in.qtot[p]; etotal += in.etot[p]; prev_qtot[p] = in.qtot[p]; }
    if (qtotal && answered_pending) {
      // the tail kernel emitted exactly these queries (same order) and the decoders answered them: do not ask twice
      BCE_HIP_TRY(c, hipMemcpyAsync(Rsbuf.p, pin_res.p, qtotal * 4, hipMemcpyHostToDevice, c->stream));
    } else if (qtotal && direct) {
      pool.Q = static_cast<const uint32_t *>(pin_q.p);
      pool.E = static_cast<const uint4 *>(pin_e.p);
      pool.res = static_cast<uint32_t *>(pin_res.p);
      pool.run(in);
      { const double t1 = now_s(); t_host += t1 - t0; t0 = t1; }
    } else if (qtotal) {
      BCE_TRY(pin_q.ensure(c, qtotal * 4));
      BCE_TRY(pin_e.ensure(c, (etotal + 1) * sizeof(uint4)));
      BCE_TRY(pin_res.ensure(c, qtotal * 4));
      BCE_HIP_TRY(c, hipMemcpyAsync(pin_q.p, a.Q, qtotal * 4, hipMemcpyDeviceToHost, c->stream));
      if (etotal) BCE_HIP_TRY(c, hipMemcpyAsync(pin_e.p, a.E, etotal * sizeof(uint4), hipMemcpyDeviceToHost, c->stream));
      BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
      { const double t1 = now_s(); t_copy += t1 - t0; t0 = t1; }
      pool.Q = static_cast<const uint32_t *>(pin_q.p);
      pool.E = static_cast<const uint4 *>(pin_e.p);
      pool.res = static_cast<uint32_t *>(pin_res.p);
      pool.run(in);
      { const double t1 = now_s(); t_host += t1 - t0; t0 = t1; }
      BCE_HIP_TRY(c, hipMemcpyAsync(Rsbuf.p, pin_res.p, qtotal * 4, hipMemcpyHostToDevice, c->stream));
    }
    answered_pending = false;
    if (small) {
      hipLaunchKernelGGL((dec_small_kernel<false>), dim3(grid), dim3(K3_T), 0, c->stream, a);
    } else {
      hipLaunchKernelGGL((dec_tiles_kernel<2>), dim3(grid), dim3(K3_T), 0, c->stream, a);
      hipLaunchKernelGGL((dec_scan_kernel<false>), dim3(8), dim3(1024), 0, c->stream, a);
      hipLaunchKernelGGL((dec_tiles_kernel<3>), dim3(grid), dim3(K3_T), 0, c->stream, a);
    }
    // the next round's size: read the control block (the query pass of the next round would tell, but its grid needs it)
    BCE_TRY(read_back(c, &ctl, c->ctl.p, sizeof ctl));
    BCE_HIP_TRY(c, hipGetLastError());
    if (ctl.err) {
      if (ctl.err == 4) snprintf(c->err, sizeof c->err, "decode: a two-launch round waited too long for a predecessor tile (round %u)", round);
      else snprintf(c->err, sizeof c->err, ctl.err == 2 ? "decode: node list overflow (capP=%u)" : "decode: inconsistent archive (round %u)",
                    ctl.err == 2 ? c->capP : round);
      c->dec_list_overflow = ctl.err == 2;
      return ctl.err == 2 ? BCE_HIP_E_OVERFLOW : BCE_HIP_E_INTERNAL;
    }
    t_c += now_s() - t0;
    if (timing) wide_time[hb] += now_s() - t_round0;
    cur_nodes = ctl.next_nodes;
    nodes_total = ctl.nodes_total;
    if (c->progress) c->progress(nodes_total, 8ull * n, c->progress_user);
    queries_total += qtotal;
    ++round;
  }
  if (timing) { fprintf(stderr, "gpu decode: %u rounds (%llu of them in the tail kernels, %llu query rounds answered through the mailbox), %llu nodes, %llu queries: %.3f s (query pass %.3f, copy out %.3f, host decoders %.3f, children pass %.3f)\n",
                        round, (unsigned long long)tail_rounds, (unsigned long long)mbox_rounds, (unsigned long long)nodes_total, (unsigned long long)queries_total, now_s() - tp0, t_q, t_copy, t_host, t_c); tp0 = now_s(); }

  if (timing) fprintf(stderr, "gpu decode: %llu six-launch rounds plane by plane (busiest first): %.3f s (launching the query passes %.3f, the first plane's %.3f, all lanes started after %.3f, waiting for decoders %.3f, last children pass %.3f)\n",
                      (unsigned long long)split_rounds, t_split, ts_issue, ts_first, ts_lanes, ts_wait, ts_rb);
  if (timing) {
    fprintf(stderr, "gpu decode: decoder threads (answers through the pool only):");
    for (int p = 0; p < 8; ++p) fprintf(stderr, " plane %d %.1f M in %.3f s (%.1f ns);", p, pool.asked_n[p] * 1e-6, pool.busy[p], pool.asked_n[p] ? pool.busy[p] * 1e9 / pool.asked_n[p] : 0.0);
    fprintf(stderr, "\n");
  }
  if (timing) fprintf(stderr, "gpu decode: lanes: busiest plane's decoder began %.3f, ended %.3f; the last decoder began %.3f, ended %.3f; all children passes queued %.3f (sums over the rounds, from the round's start); last to end: %u %u %u %u %u %u %u %u\n",
                      dbg_r[0], dbg_r[1], dbg_r[3], dbg_r[2], dbg_r[4], dbg_last[0], dbg_last[1], dbg_last[2], dbg_last[3], dbg_last[4], dbg_last[5], dbg_last[6], dbg_last[7]);
  if (timing) {
    fprintf(stderr, "gpu decode: %llu rounds in two launches (dec_small_kernel); rounds outside the tail by node count:", (unsigned long long)small_rounds);
    for (int b = 0; b < 32; ++b) if (wide_hist[b]) fprintf(stderr, " [2^%d) %llu rounds %.1f M nodes %.3f s;", b, (unsigned long long)wide_hist[b], wide_nodes[b] * 1e-6, wide_time[b]);
    fprintf(stderr, "\n");
  }
  if (timing) fprintf(stderr, "gpu decode: tail kernels: wave %llu launches (%llu rounds, %llu nodes) %.3f s, workgroup %llu launches (%llu rounds, %llu nodes) %.3f s\n",
                      (unsigned long long)launches_wave, (unsigned long long)rounds_wave, (unsigned long long)nodes_wave, t_wave,
                      (unsigned long long)launches_wg, (unsigned long long)rounds_wg, (unsigned long long)nodes_wg, t_wg);
  // ---- R -> planes -> granules -> BWT bytes ----
  FillArgs f;
  f.R = R; f.n = n;
  f.chunks = (uint32_t)(((uint64_t)n + 1 + FG_CHUNK - 1) / FG_CHUNK);
  f.nwords = (uint32_t)(((uint64_t)n + 31) / 32) + 3;
  const uint32_t ngran = (uint32_t)((uint64_t)n / 96) + 2;
  BCE_TRY(ensure(c, c->blk, (size_t)8 * f.chunks * 4 + 64));
  f.cmax = c->blk.as<uint32_t>();
  f.err = f.cmax + (size_t)8 * f.chunks;
  BCE_TRY(ensure(c, c->sa[0], 8 * rstride));                   // gap types
  f.type = c->sa[0].as<uint8_t>();
  BCE_TRY(ensure(c, c->key[0], (size_t)8 * f.nwords * 4));
  BCE_TRY(ensure(c, c->key[1], (size_t)8 * f.nwords * 4));
  f.words = c->key[0].as<uint32_t>();
  f.rankw = c->key[1].as<uint32_t>();
  BCE_TRY(ensure(c, c->gran, (size_t)8 * ngran * sizeof(Granule)));
  BCE_HIP_TRY(c, hipMemsetAsync(f.err, 0, 4, c->stream));
  BCE_HIP_TRY(c, hipMemsetAsync(f.words, 0, (size_t)8 * f.nwords * 4, c->stream));
  BCE_HIP_TRY(c, hipMemsetAsync(f.rankw, 0, (size_t)8 * f.nwords * 4, c->stream));
  hipLaunchKernelGGL(fill_chunkmax_kernel, dim3(f.chunks, 8), dim3(FG_T), 0, c->stream, f);
  hipLaunchKernelGGL(fill_chunkscan_kernel, dim3(8), dim3(1024), 0, c->stream, f);
  hipLaunchKernelGGL((fill_kernel<0>), dim3(f.chunks, 8), dim3(FG_T), 0, c->stream, f);
  hipLaunchKernelGGL((fill_kernel<1>), dim3(f.chunks, 8), dim3(FG_T), 0, c->stream, f);
  {
    uint32_t gb = (ngran + 255) / 256;
    hipLaunchKernelGGL(gran_from_words_kernel, dim3(gb < 4096 ? gb : 4096, 8), dim3(256), 0, c->stream, f, c->gran.as<Granule>(), ngran);
  }
  uint32_t ferr = 0;
  BCE_TRY(read_back(c, &ferr, f.err, 4));
  BCE_HIP_TRY(c, hipGetLastError());
  if (ferr) { snprintf(c->err, sizeof c->err, "decode: a mixed gap was never split"); return BCE_HIP_E_INTERNAL; }
  BCE_TRY(ensure(c, c->bwt, n));
  BCE_TRY(ensure(c, c->text, n));
  BCE_TRY(ensure(c, c->stat, 64));
  uint32_t *dz = c->stat.as<uint32_t>();
  BCE_HIP_TRY(c, hipMemcpyAsync(dz, a.zeros, 32, hipMemcpyHostToDevice, c->stream));
  BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
  const uint32_t gn = (uint32_t)((((uint64_t)n + 255) / 256) < 8192 ? (((uint64_t)n + 255) / 256) : 8192);
  hipLaunchKernelGGL(access_kernel, dim3(gn), dim3(256), 0, c->stream, c->gran.as<Granule>(), ngran, n, dz, c->bwt.as<uint8_t>());
  if (timing) { BCE_HIP_TRY(c, hipStreamSynchronize(c->stream)); fprintf(stderr, "gpu decode: planes + unbwt %.3f s\n", now_s() - tp0); tp0 = now_s(); }

  // ---- inverse BWT ----
  const size_t b4 = (size_t)n * 4;
  for (int i = 0; i < 2; ++i) { BCE_TRY(ensure(c, c->sa[i], b4)); BCE_TRY(ensure(c, c->key[i], b4)); }
  BCE_TRY(ensure(c, c->rank, b4));
  uint32_t *key[2] = {c->key[0].as<uint32_t>(), c->key[1].as<uint32_t>()};
  uint32_t *val[2] = {c->sa[0].as<uint32_t>(), c->sa[1].as<uint32_t>()};
  uint32_t *lf = c->rank.as<uint32_t>();
  hipLaunchKernelGGL(lf_keys_kernel, dim3(gn), dim3(256), 0, c->stream, c->bwt.as<uint8_t>(), n, key[0], val[0]);
  int res = 0;
  BCE_TRY(radix_sort_pairs(c, key, val, n, 0, 8, &res, 8));
  hipLaunchKernelGGL(lf_scatter_kernel, dim3(gn), dim3(256), 0, c->stream, val[res], n, lf);
  uint32_t sh = 0;
  while (((uint64_t)n >> sh) > (1u << 19)) ++sh;                // at most 2^19 walkers
  // ... and at least 256 rows per walker while that leaves a few thousand of them: the host chains the segments one after
  // the other (a random access each), which for one-row segments of a 1 MB input was 10 ms of a 36 ms decode
  while (sh < 8 && ((uint64_t)n >> (sh + 1)) >= 4096) ++sh;
  const uint32_t m = (uint32_t)((((uint64_t)n - 1) >> sh) + 1);
  BCE_TRY(ensure(c, c->key[0], (size_t)3 * m * 4 + 16));         // the sort's key buffers are free again
  uint32_t *d_len = c->key[0].as<uint32_t>(), *d_end = d_len + m, *d_dest = d_len + 2 * (size_t)m;
  hipLaunchKernelGGL(walk_len_kernel, dim3((m + 63) / 64), dim3(64), 0, c->stream, lf, m, sh, d_len, d_end);
  std::vector<uint32_t> h_len(m), h_end(m), h_dest(m, 0);
  BCE_HIP_TRY(c, hipMemcpyAsync(h_len.data(), d_len, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
  BCE_HIP_TRY(c, hipMemcpyAsync(h_end.data(), d_end, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
  BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
  BCE_HIP_TRY(c, hipGetLastError());
  // chain the segments from row 0 (= the last text position) until the walk is back at row 0: one cycle through all
  // n rows for a primitive input; a shorter cycle (length lc, the input's period pattern in BWT terms) otherwise
  uint64_t lc = 0;
  {
    uint32_t cur = 0;
    std::vector<uint8_t> visited(m, 0);
    std::vector<uint32_t> order;
    do {
      const uint32_t j = cur >> sh;
      if (visited[j]) return BCE_HIP_E_INTERNAL;                 // cannot happen: LF is a permutation
      visited[j] = 1;
      order.push_back(j);
      lc += h_len[j];
      cur = h_end[j];
    } while (cur != 0);
    uint64_t pos = lc;
    for (uint32_t j : order) { h_dest[j] = (uint32_t)pos; pos -= h_len[j]; }
    for (uint32_t j = 0; j < m; ++j) if (!visited[j]) h_len[j] = 0;      // rows off the cycle are never written
  }
  if (lc == 0 || lc > n || n % lc) { snprintf(c->err, sizeof c->err, "decode: LF cycle of length %llu in %u rows", (unsigned long long)lc, n); return BCE_HIP_E_INTERNAL; }
  const uint32_t off = hd.offset % n;
  const bool single_cycle = lc == n;
  BCE_HIP_TRY(c, hipMemcpyAsync(d_dest, h_dest.data(), (size_t)m * 4, hipMemcpyHostToDevice, c->stream));
  BCE_HIP_TRY(c, hipMemcpyAsync(d_len, h_len.data(), (size_t)m * 4, hipMemcpyHostToDevice, c->stream));
  if (single_cycle) {
    hipLaunchKernelGGL(walk_write_kernel, dim3((m + 63) / 64), dim3(64), 0, c->stream, lf, c->bwt.as<uint8_t>(), m, sh, d_len, d_dest,
                       n, off, c->text.as<uint8_t>());
  } else {
    // periodic input (the reference's decoder returns zeros here): write the cycle once, then unroll it
    uint8_t *V = reinterpret_cast<uint8_t *>(val[0]);            // the sort's value buffers are free again
    hipLaunchKernelGGL(walk_write_kernel, dim3((m + 63) / 64), dim3(64), 0, c->stream, lf, c->bwt.as<uint8_t>(), m, sh, d_len, d_dest,
                       (uint32_t)lc, 0u, V);
    hipLaunchKernelGGL(expand_cycle_kernel, dim3(gn), dim3(256), 0, c->stream, V, (uint32_t)lc, n, off, c->text.as<uint8_t>());
  }
  BCE_HIP_TRY(c, hipMemcpyAsync(out, c->text.p, n, hipMemcpyDeviceToHost, c->stream));
  BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
  BCE_HIP_TRY(c, hipGetLastError());
  if (timing) fprintf(stderr, "gpu decode: inverse BWT (%s, %u walkers) %.3f s\n", single_cycle ? "one cycle" : "periodic", m, now_s() - tp0);
  if (timing) fprintf(stderr, "gpu decode: this context so far: device allocations %u calls %.1f MB %.3f s, pinned (query / answer buffers) %u calls %.1f MB %.3f s\n",
                      c->alloc_calls, c->alloc_bytes / 1e6, c->alloc_s, c->pin_calls, c->pin_bytes / 1e6, c->pin_s);
  return BCE_HIP_OK;
}
