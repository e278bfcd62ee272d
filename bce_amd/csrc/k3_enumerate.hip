// k3_enumerate.hip -- K3: the interval split/count enumeration ("interval-count kernel").
// Replaces the round loop of BCE::code, mode 1 (bce.cpp:1236-1374; node body :1261-1351).
//
// One round = one pass over every plane's node list (all 8 planes in the same launches).  A node is an
// absolute (s, x0, x1) triple; lists are sorted by s, children come out sorted, so every round is a
// streaming read of triples + 2-3 rank gathers per node + two stable compactions (child0 / child1
// lists of the next plane) + one compaction of symbol records.  The gamma-coded pArray queues
// (bce.cpp:226-356) do not exist here.
//
// Node lists of a plane live in one buffer of cap nodes: the child0 list (positions < zeros, the
// reference's Q[i][0]) grows up from index 0, the child1 list (Q[i][1]) grows DOWN from cap-1, so the
// two lists share the capacity whatever their split is.  The two parities of a round (the lists it reads, the
// lists it writes) are buffers of their own with capacities of their own (K3Args::cap): the one a round is
// about to write holds nothing and is replaced by a larger one when the round does not fit (k3_grow_lists).
//
// Wide rounds: count kernel -> per-plane scan (last block folds the totals into the control block) -> write
// kernel, all parameterised by the round parity and driven by a device-resident control block, so the host can
// queue many rounds without reading anything back.  If the symbol buffer could overflow, the scan marks the
// round as skipped (need_flush) and every later queued round becomes a no-op; the host flushes the model (K4)
// and resumes from that round.  Narrow rounds: k3_tail_kernel below (one persistent workgroup, lists in LDS).
// The end of the enumeration: k3_dfs.hip (depth-first walkers with chain skipping).
#include <stdlib.h>

#include "common.h"
#include "scan_util.h"
#include "k3_args.h"

namespace bce {

// tile -> plane lookup table: tp[p] = first tile of plane p, tp[8] = total tiles; cn[p] = the plane's two list lengths
// (read once per block: a load from the control block on every tile's path is a dependent L2 round trip)
__device__ __forceinline__ void tile_prefix(const K3Args &a, uint32_t tp[9], uint32_t (*cn)[2] = nullptr, uint32_t *skip = nullptr) {
  // (the flags ride in the same batch of loads as the counts: the kernel's prologue is one L2 round trip, not two)
  if (skip) *skip = a.ctl->need_flush | a.ctl->overflow;
  uint32_t acc = 0;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    tp[p] = acc;
    const uint32_t m0 = a.ctl->cnt[a.par][p][0], m1 = a.ctl->cnt[a.par][p][1];
    if (cn) { cn[p][0] = m0; cn[p][1] = m1; }
    acc += (m0 + m1 + K3_TILE - 1) / K3_TILE;
  }
  tp[8] = acc;
}

// groups = runs of 256 tiles of one plane (two-launch rounds): gp[p] = first group of plane p, gp[8] = all groups
__device__ __forceinline__ void group_prefix(const uint32_t tp[9], uint32_t gp[9]) {
  uint32_t acc = 0;
#pragma unroll
  for (int p = 0; p < 8; ++p) { gp[p] = acc; acc += (tp[p + 1] - tp[p] + 255u) >> 8; }
  gp[8] = acc;
}

// Classify the K3_NPT nodes of one thread in PHASES so that the loads of all its nodes are in flight
// together: (1) the node triples, (2) the two rank granules of every node, (3) the third granule only for
// the nodes that code a symbol and whose split point falls in neither of the two granules already loaded.
// (Measured, round 2 SQ counters: latency-bound -- 91 VALU instructions per node, VALU 31 % busy, waves 51 % of their
// cycles in s_waitcnt; staging the tile's granule range through LDS was tried and is slower.)
// granule g of a plane: 32-bit byte offset from the plane's (uniform) base -- a plane's directory is < 2^32 bytes
// (n < 2^31 positions / 96 * 16 B), so the address is base (scalar) + offset (one shift) instead of a 64-bit multiply-add
__device__ __forceinline__ Granule gran_at(const Granule *G, uint32_t g) {
  return *reinterpret_cast<const Granule *>(reinterpret_cast<const char *>(G) + (g << 4));
}

struct TileOut {
  uint32_t has0[K3_NPT], has1[K3_NPT], hassym[K3_NPT];
  Node c0[K3_NPT], c1[K3_NPT];
  uint32_t kw[K3_NPT], ew[K3_NPT];
  uint32_t raw[K3_NPT][3];   // scan mode: kw = sym, ew = k, raw = c1, c2, cs
};

// node q of a plane's list (child0 part grows up from 0, child1 part down from cap-1): 32-bit byte offset from the
// plane's (uniform) base while a list is below 4 GB (cap * 12 < 2^32: every input up to ~7 * 10^8 bytes, and larger ones
// until a round needs more); a 64-bit index beyond that (lists of up to n/2 + 2 = 2^30 + 1 nodes, 12.9 GB)
__device__ __forceinline__ Node node_at(const Node *src, uint32_t idx) {
  return *reinterpret_cast<const Node *>(reinterpret_cast<const char *>(src) + ((idx << 3) + (idx << 2)));
}
constexpr uint32_t K3_CAP32 = 0xFFFFFFFFu / (uint32_t)sizeof(Node) - 16u;     // lists up to here use the 32-bit offsets

__device__ __forceinline__ void k3_load_nodes(const K3Args &a, uint32_t p, uint32_t tile_in_plane, uint32_t c0n, uint32_t c1n,
                                              Node (&nd)[K3_NPT]) {
  const uint32_t M = c0n + c1n;
  const Node *src = plane_nodes(a, a.par, p);
  const uint32_t cap = list_cap(a, a.par);
  if (cap <= a.cap32) {                                         // (uniform)
#pragma unroll
    for (int it = 0; it < K3_NPT; ++it) {
      const uint32_t q = tile_in_plane * K3_TILE + (uint32_t)it * K3_T + threadIdx.x;
      // out-of-range lanes re-read the tile's first node (always valid when the tile exists); their results are masked
      const uint32_t qq = q < M ? q : tile_in_plane * K3_TILE;
      nd[it] = node_at(src, qq < c0n ? qq : (cap - 1u - (qq - c0n)));
    }
  } else {
#pragma unroll
    for (int it = 0; it < K3_NPT; ++it) {
      const uint32_t q = tile_in_plane * K3_TILE + (uint32_t)it * K3_T + threadIdx.x;
      const uint32_t qq = q < M ? q : tile_in_plane * K3_TILE;
      nd[it] = src[qq < c0n ? qq : (cap - 1u - (qq - c0n))];
    }
  }
}

template <bool PACK, bool SCAN>
__device__ __forceinline__ void k3_classify(const K3Args &a, uint32_t p, uint32_t tile_in_plane, uint32_t c0n, uint32_t c1n,
                                            TileOut &t) {
  const uint32_t tid = threadIdx.x;
  const uint32_t M = c0n + c1n;
  // (prefetching the NEXT tile's triples behind this tile's granule loads was measured: no gain -- vmcnt retires in order,
  //  so the conditional third gather and, in the write kernel, the stores wait for the prefetch anyway, and the 12 extra
  //  registers cost a wave per SIMD)
  Node nd[K3_NPT];
  k3_load_nodes(a, p, tile_in_plane, c0n, c1n, nd);
  __builtin_amdgcn_sched_barrier(0);            // every node load issued before the first is consumed
  const Granule *G = a.gran + (size_t)p * a.ngran;
  const uint32_t zp = a.zeros[p];
  const PlaneCfg &cfg = a.cfg[p];
  const uint32_t pen = (a.pmask >> p) & 1u;      // is this plane's symbol stream recorded here? (see K3Args::pmask)
  uint32_t valid[K3_NPT];
#pragma unroll
  for (int it = 0; it < K3_NPT; ++it) valid[it] = tile_in_plane * K3_TILE + (uint32_t)it * K3_T + tid < M ? 1u : 0u;
  // the rank granules of every node in one phase (the split point's granule is almost always one of the other two; the
  // rest is a third, dependent load below.  Loading it unconditionally was measured: 13.9 instead of 13.4 ms)
  uint32_t ga[K3_NPT], gb[K3_NPT], gm[K3_NPT];
  Granule qa[K3_NPT], qb[K3_NPT], qm[K3_NPT];
#pragma unroll
  for (int it = 0; it < K3_NPT; ++it) {
    ga[it] = div96(nd[it].s);
    gb[it] = div96(nd[it].s + nd[it].x0 + nd[it].x1);
    gm[it] = div96(nd[it].s + nd[it].x0);
    qa[it] = gran_at(G, ga[it]);
    qb[it] = gran_at(G, gb[it]);
  }
  __builtin_amdgcn_sched_barrier(0);            // every load above is issued before the first rank below
  NodeFlat nf[K3_NPT];
#pragma unroll
  for (int it = 0; it < K3_NPT; ++it) {
    const uint32_t rs = granule_rank1(qa[it], nd[it].s - ga[it] * 96u);
    const uint32_t re = granule_rank1(qb[it], nd[it].s + nd[it].x0 + nd[it].x1 - gb[it] * 96u);
    node_flat_pre(nd[it], rs, re, nf[it]);
    qm[it] = gm[it] == ga[it] ? qa[it] : qb[it];
    if (nf[it].need_mid && gm[it] != ga[it] && gm[it] != gb[it]) qm[it] = gran_at(G, gm[it]);
  }
#pragma unroll
  for (int it = 0; it < K3_NPT; ++it) {
    const uint32_t rm = granule_rank1(qm[it], nd[it].s + nd[it].x0 - gm[it] * 96u);
    uint32_t sym, k;
    node_flat_post(nd[it], zp, nf[it], rm, t.has0[it], t.c0[it], t.has1[it], t.c1[it], sym, k);
    t.has0[it] &= valid[it];
    t.has1[it] &= valid[it];
    t.hassym[it] = nf[it].need_mid & valid[it] & pen;
    t.kw[it] = t.ew[it] = 0;
    if (SCAN) {
      t.kw[it] = sym; t.ew[it] = k;
      t.raw[it][0] = nf[it].n0x; t.raw[it][1] = nd[it].x1; t.raw[it][2] = nd[it].x0 + nd[it].x1;
    } else if (PACK && t.hassym[it]) {
      pack_symbol(cfg, p, sym, k, nf[it].n0x, nd[it].x1, nd[it].x0 + nd[it].x1, t.kw[it], t.ew[it]);
    }
  }
}

// Write one classified tile: children into plane p+1's lists of the other parity (child0 list grows up from o0,
// child1 list down from the top), symbol records from os.  lds_cnt = per (item, wave) counts of the tile.
template <bool SCAN>
__device__ __forceinline__ void k3_place(const K3Args &a, uint32_t p, const TileOut &t, const uint32_t (&pre0)[K3_NPT],
                                         const uint32_t (&pre1)[K3_NPT], const uint32_t (&pres)[K3_NPT],
                                         uint32_t (*lds_cnt)[4][3], uint32_t o0, uint32_t o1, uint64_t os) {
  const uint32_t w = threadIdx.x >> 6;
  const uint32_t pn = (p + 1u) & 7u;
  Node *dst = plane_nodes(a, a.par ^ 1u, pn);
  const uint32_t capo = list_cap(a, a.par ^ 1u);
  uint32_t run0 = 0, run1 = 0, runs_ = 0;   // counts of earlier (it, wave) groups
#pragma unroll
  for (int it = 0; it < K3_NPT; ++it) {
    uint32_t b0 = run0, b1 = run1, bs = runs_;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) {
      const uint32_t x0 = lds_cnt[it][ww][0], x1 = lds_cnt[it][ww][1], xs = lds_cnt[it][ww][2];
      if ((uint32_t)ww < w) { b0 += x0; b1 += x1; bs += xs; }
      run0 += x0; run1 += x1; runs_ += xs;
    }
    if (t.has0[it]) dst[o0 + b0 + pre0[it]] = t.c0[it];
    if (t.has1[it]) dst[capo - 1u - (o1 + b1 + pre1[it])] = t.c1[it];
    if (t.hassym[it]) {
      if (SCAN) {
        a.scanrec[os + bs + pres[it]] = scan_pack(t.kw[it], t.ew[it], t.raw[it][0], t.raw[it][1], t.raw[it][2]);
      } else {
        a.symkey[os + bs + pres[it]] = t.kw[it];
        a.symesc[os + bs + pres[it]] = t.ew[it];
      }
    }
  }
}

// Process one tile.  WRITE=false: count children/symbols.  WRITE=true: place them.
template <bool WRITE, bool SCAN>
__device__ __forceinline__ void k3_tile(const K3Args &a, uint32_t p, uint32_t tile_in_plane, uint32_t tile_global,
                                        uint32_t (*lds_cnt)[4][3], uint32_t c0n, uint32_t c1n, uint32_t group_base) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
  // the tile's output offsets do not depend on the classification: their (dependent, L2) loads go out first instead of
  // after the barrier below -- a round of 1-2 M nodes is ONE tile per block, so every round trip on the tile's path is
  // per-round fixed cost (~45 us per round in all)
  uint32_t o0 = 0, o1 = 0;
  uint64_t os = 0;
  if (WRITE) {
    o0 = a.tileoff[(size_t)tile_global * 4 + 0];
    o1 = a.tileoff[(size_t)tile_global * 4 + 1];
    os = a.ctl->symbase[p] + a.tileoff[(size_t)tile_global * 4 + 2];
    if (a.fused) {                                            // two-launch rounds: tile offsets are relative to their group
      const uint32_t *g = a.goff + (size_t)(group_base + (tile_in_plane >> 8)) * 4;
      o0 += g[0]; o1 += g[1]; os += g[2];
    }
  }
  TileOut t;
  k3_classify<WRITE, SCAN>(a, p, tile_in_plane, c0n, c1n, t);
  uint32_t pre0[K3_NPT], pre1[K3_NPT], pres[K3_NPT];
  const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int it = 0; it < K3_NPT; ++it) {
    const uint64_t b0 = __ballot(t.has0[it]), b1 = __ballot(t.has1[it]), bs = __ballot(t.hassym[it]);
    pre0[it] = (uint32_t)__popcll(b0 & lt);
    pre1[it] = (uint32_t)__popcll(b1 & lt);
    pres[it] = (uint32_t)__popcll(bs & lt);
    if (lane == 0) {
      lds_cnt[it][w][0] = (uint32_t)__popcll(b0);
      lds_cnt[it][w][1] = (uint32_t)__popcll(b1);
      lds_cnt[it][w][2] = (uint32_t)__popcll(bs);
    }
  }
  __syncthreads();
  if (!WRITE) {
    if (tid < 3) {
      uint32_t tt = 0;
      for (int it = 0; it < K3_NPT; ++it)
        for (int ww = 0; ww < 4; ++ww) tt += lds_cnt[it][ww][tid];
      a.tilecnt[(size_t)tile_global * 4 + tid] = tt;
    }
  } else {
    k3_place<SCAN>(a, p, t, pre0, pre1, pres, lds_cnt, o0, o1, os);
  }
  __syncthreads();
}

__device__ __forceinline__ uint32_t tile_plane(const uint32_t *tp, uint32_t tile) {
  uint32_t p = 0;
#pragma unroll
  for (int k = 1; k < 8; ++k) p += (tile >= tp[k]) ? 1u : 0u;
  return p;
}

template <bool WRITE, bool SCAN>
__global__ __launch_bounds__(K3_T) void k3_tiles_kernel(K3Args a) {
  __shared__ uint32_t tp[9], gp[9], cn[8][2], s_skip;
  __shared__ uint32_t lds_cnt[K3_NPT][4][3];
  if (threadIdx.x == 0) { tile_prefix(a, tp, cn, &s_skip); group_prefix(tp, gp); }
  __syncthreads();
  if (s_skip) return;
  const uint32_t T = tp[8];
  for (uint32_t tile = blockIdx.x; tile < T; tile += gridDim.x) {
    const uint32_t p = tile_plane(tp, tile);
    k3_tile<WRITE, SCAN>(a, p, tile - tp[p], tile, lds_cnt, cn[p][0], cn[p][1], gp[p]);
  }
}

// Scan: block p scans the tile counts of plane p (tile offsets are plane-local); the block that finishes
// last folds the 8 plane totals into the control block: symbol bases, flush / overflow decisions, next-round
// counts, run table.  (threadfence + atomic ticket: the classic last-block pattern.)
__global__ __launch_bounds__(1024) void k3_scan_kernel(K3Args a) {
  __shared__ uint32_t tp[9];
  __shared__ uint32_t s_last;
  EnumCtl *ctl = a.ctl;
  if (ctl->need_flush || ctl->overflow) return;
  const uint32_t tid = threadIdx.x, p = blockIdx.x;
  if (tid == 0) tile_prefix(a, tp);
  __syncthreads();
  uint32_t r0 = 0, r1 = 0, rs = 0;
  for (uint32_t base = tp[p]; base < tp[p + 1]; base += 1024) {
    const uint32_t t = base + tid;
    const bool valid = t < tp[p + 1];
    const uint32_t v0 = valid ? a.tilecnt[(size_t)t * 4 + 0] : 0u;
    const uint32_t v1 = valid ? a.tilecnt[(size_t)t * 4 + 1] : 0u;
    const uint32_t vs = valid ? a.tilecnt[(size_t)t * 4 + 2] : 0u;
    // child counts <= 1024 per tile: pack both in one u64 scan with the symbol count
    uint64_t tot;
    const uint64_t ex = block_excl_scan_sum64<1024>((uint64_t)v0 | ((uint64_t)v1 << 21) | ((uint64_t)vs << 42), &tot);
    if (valid) {
      a.tileoff[(size_t)t * 4 + 0] = r0 + (uint32_t)(ex & 0x1FFFFFu);
      a.tileoff[(size_t)t * 4 + 1] = r1 + (uint32_t)((ex >> 21) & 0x1FFFFFu);
      a.tileoff[(size_t)t * 4 + 2] = rs + (uint32_t)(ex >> 42);
    }
    r0 += (uint32_t)(tot & 0x1FFFFFu); r1 += (uint32_t)((tot >> 21) & 0x1FFFFFu); rs += (uint32_t)(tot >> 42);
  }
  if (tid == 0) {
    ctl->ptot[p][0] = r0; ctl->ptot[p][1] = r1; ctl->ptot[p][2] = rs;
    __threadfence();
    s_last = atomicAdd(&ctl->ticket, 1u) == 7u ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  // epilogue of the last block: fetch everything it reads in parallel (a serial chain of ~45 dependent loads by
  // one thread was most of this kernel's 11 us), then thread 0 decides and stores
  __shared__ uint32_t s_pt[8][3], s_cnt[8][2];
  __shared__ uint64_t s_sym[2];
  __shared__ uint32_t s_done;
  __threadfence();
  if (tid < 24) s_pt[tid / 3][tid % 3] = ((const volatile uint32_t *)&ctl->ptot[0][0])[tid];
  else if (tid < 40) s_cnt[(tid - 24) >> 1][(tid - 24) & 1] = ((const volatile uint32_t *)&ctl->cnt[a.par][0][0])[tid - 24];
  else if (tid == 40) s_sym[0] = *(const volatile uint64_t *)&ctl->sym_total;
  else if (tid == 41) s_sym[1] = *(const volatile uint64_t *)&ctl->sym_cap;
  else if (tid == 42) s_done = *(const volatile uint32_t *)&ctl->done_round;
  __syncthreads();
  if (tid != 0) return;
  ctl->ticket = 0;
  uint64_t symsum = 0, nextn = 0, curn = 0;
  uint32_t want_list = 0;
  bool ovf = false;
  const uint32_t (*pt)[3] = s_pt;
  for (int q = 0; q < 8; ++q) {
    symsum += pt[q][2];
    nextn += (uint64_t)pt[q][0] + pt[q][1];
    curn += (uint64_t)s_cnt[q][0] + s_cnt[q][1];
    if ((uint64_t)pt[q][0] + pt[q][1] > list_cap(a, a.par ^ 1u)) ovf = true;
    if ((uint64_t)pt[q][0] + pt[q][1] > want_list) want_list = (uint32_t)(pt[q][0] + pt[q][1]);
  }
  if (ovf) { ctl->overflow = 1; ctl->skip_round = a.round; ctl->want_list = want_list; return; }       // (the host grows the lists and runs the round again: k3_grow_lists)
  if (s_sym[0] + symsum > s_sym[1]) { ctl->need_flush = 1; ctl->skip_round = a.round; ctl->want_syms = symsum; return; }
  uint64_t acc = s_sym[0];
  for (uint32_t q = 0; q < 8; ++q) {
    const uint32_t qn = (q + 1u) & 7u;
    ctl->symbase[q] = acc;
    ctl->cnt[a.par ^ 1u][qn][0] = pt[q][0];
    ctl->cnt[a.par ^ 1u][qn][1] = pt[q][1];
    RunEntry e; e.start = acc; e.count = pt[q][2]; e.round = a.round;
    a.runs[(size_t)a.run_slot * 8 + q] = e;
    acc += pt[q][2];
  }
  ctl->sym_total = acc;
  if (!a.repeat) atomicAdd((unsigned long long *)&ctl->nodes_total, (unsigned long long)curn);
  ctl->next_nodes = nextn;
  if (nextn == 0 && s_done == 0xFFFFFFFFu) ctl->done_round = a.round + 1u;
}

// ------------------------------------------------------------------------------------------------------
// Wide rounds in TWO launches: the scan rides on the count kernel.  Every tile publishes its three counts in a
// round-tagged 64-bit word (a plain store: nothing on the tile's path waits for memory).  The block that handles
// the LAST tile of a group (256 consecutive tiles of a plane) waits for the group's words, scans them (one per
// thread) into group-relative tile offsets and publishes the group totals; the block of the round's last tile
// then waits for the group words of every plane, scans them and runs the epilogue k3_scan_kernel runs (flush /
// overflow decision, symbol bases, next list sizes, run table).  Words carry the round number, so nothing is
// reset and a reader that comes early spins instead of reading stale counts.  A waiter only waits for tiles
// with smaller indices, i.e. for blocks that were dispatched no later than itself and wait, if at all, for
// still smaller ones: no cycle.  (A ticket per tile -- an atomic with a return value on every tile's path --
// made the kernel 70 % slower.)  The write kernel adds the group offset to the tile offset.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t ld_word(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Wait for a round-tagged word.  The waits in this file rely on the predecessor's block running or done (in-order
// dispatch, the device gate of api.hip); if that ever failed -- new firmware, a partition mode, a foreign spinning kernel
// on the device -- the wait would be a silent GPU hang.  So it is bounded: after K3_SPIN_LIMIT polls (seconds) the
// round is declared stalled, every later kernel no-ops (overflow flag) and the host returns an error.  The word
// that comes back then carries the right tag and zero counts: offsets derived from it stay inside the lists.
constexpr uint32_t K3_SPIN_LIMIT = 1u << 22;
__device__ __forceinline__ uint64_t wait_word(const unsigned long long *p, int shift, uint64_t epoch, uint64_t w, EnumCtl *ctl) {
  uint32_t spins = 0;
  while ((w >> shift) != epoch) {
    __builtin_amdgcn_s_sleep(1);
    w = ld_word(p);
    if (++spins == K3_SPIN_LIMIT) { ctl->stalled = 1; ctl->overflow = 1; return epoch << shift; }
  }
  return w;
}
__device__ __forceinline__ void st_word(unsigned long long *p, uint64_t v) {
  __hip_atomic_store(p, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool SCAN>
__global__ __launch_bounds__(K3_T) void k3_count2_kernel(K3Args a) {
  __shared__ uint32_t tp[9], gp[9], cn[8][2], s_skip;
  __shared__ uint32_t lds_cnt[K3_NPT][4][3];
  __shared__ uint32_t s_pt[8][3];
  EnumCtl *ctl = a.ctl;
  const uint32_t tid = threadIdx.x;
  if (tid == 0) { tile_prefix(a, tp, cn, &s_skip); group_prefix(tp, gp); }
  __syncthreads();
  if (s_skip) return;
  const uint32_t T = tp[8];
  if (T == 0) {
    // a queued round after the end: what the epilogue would do with nothing -- an empty run-table row and EMPTY
    // lists for the next round (its parity still holds the lists of two rounds ago)
    if (blockIdx.x == 0 && tid < 8) { RunEntry e; e.start = ctl->sym_total; e.count = 0; e.round = a.round; a.runs[(size_t)a.run_slot * 8 + tid] = e; }
    if (blockIdx.x == 0 && tid < 16) ctl->cnt[a.par ^ 1u][tid >> 1][tid & 1] = 0;
    if (blockIdx.x == 0 && tid == 0) ctl->next_nodes = 0;
    return;
  }
  const uint64_t epoch = (uint64_t)((a.round + 1u) & 0x3FFFFFFu);   // 26 bits: the group words keep 38 for their payload
  for (uint32_t tile = blockIdx.x; tile < T; tile += gridDim.x) {
    const uint32_t p = tile_plane(tp, tile);
    const uint32_t ti = tile - tp[p], Tp = tp[p + 1] - tp[p];
    const uint32_t lane = tid & 63u, w = tid >> 6;
    {
      TileOut t;
      k3_classify<false, SCAN>(a, p, ti, cn[p][0], cn[p][1], t);
#pragma unroll
      for (int it = 0; it < K3_NPT; ++it) {
        const uint64_t b0 = __ballot(t.has0[it]), b1 = __ballot(t.has1[it]), bs = __ballot(t.hassym[it]);
        if (lane == 0) {
          lds_cnt[it][w][0] = (uint32_t)__popcll(b0);
          lds_cnt[it][w][1] = (uint32_t)__popcll(b1);
          lds_cnt[it][w][2] = (uint32_t)__popcll(bs);
        }
      }
    }
    __syncthreads();
    if (w == 0) {                                              // lanes 0..2 add up one count each, lane 0 publishes
      uint32_t tt = 0;
      if (lane < 3)
        for (int it = 0; it < K3_NPT; ++it)
          for (int ww = 0; ww < 4; ++ww) tt += lds_cnt[it][ww][lane];
      const uint64_t t1 = (uint32_t)__shfl((int)tt, 1), t2 = (uint32_t)__shfl((int)tt, 2);
      // [10:0] child0, [21:11] child1, [32:22] symbols (<= 1024 each), [58:33] round tag.  Fire and forget.
      if (lane == 0) st_word(&a.tw[tile], (epoch << 33) | (t2 << 22) | (t1 << 11) | (uint64_t)tt);
    }
    const uint32_t g = ti >> 8, gidx = gp[p] + g, gfirst = g << 8, nin = Tp - gfirst < 256u ? Tp - gfirst : 256u;
    if (ti == gfirst + nin - 1u) {
      // ---- the block of a group's LAST tile scans the group (K3_T = 256 threads, one tile each).  Its tiles are
      // in flight in blocks that started no later than this one and wait for nothing: spinning is safe. ----
      const uint32_t j = tp[p] + gfirst + tid;
      const bool valid = tid < nin;
      uint64_t wv = 0;
      if (valid) wv = wait_word(&a.tw[j], 33, epoch, ld_word(&a.tw[j]), ctl);
      const uint64_t v = valid ? ((wv & 0x7FFu) | (((wv >> 11) & 0x7FFu) << 21) | (((wv >> 22) & 0x7FFu) << 42)) : 0ull;
      uint64_t tot;
      const uint64_t ex = block_excl_scan_sum64<K3_T>(v, &tot);
      if (valid) {
        a.tileoff[(size_t)j * 4 + 0] = (uint32_t)(ex & 0x1FFFFFu);
        a.tileoff[(size_t)j * 4 + 1] = (uint32_t)((ex >> 21) & 0x1FFFFFu);
        a.tileoff[(size_t)j * 4 + 2] = (uint32_t)(ex >> 42);
      }
      if (tid == 0) {
        // group totals < 2^19 each: word a = tag | child1 << 19 | child0, word b = tag | symbols
        st_word(&a.gwa[gidx], (epoch << 38) | (((tot >> 21) & 0x1FFFFFu) << 19) | (tot & 0x1FFFFFu));
        st_word(&a.gwb[gidx], (epoch << 38) | (tot >> 42));
      }
      if (tile == T - 1u) {
        // ---- the block of the round's last tile: scan the groups of every plane, then the round's bookkeeping ----
        if (gp[8] <= (uint32_t)K3_T) {
          // all groups of all planes in ONE scan (<= 256 groups = 65 536 tiles = 67 M nodes per round): the plane-local
          // offset is the block-wide prefix minus the prefix at the plane's first group.  (Plane by plane this tail was
          // eight scans and ~6 us of every round -- a quarter of the count kernel on rounds of 1-2 M nodes.)
          __shared__ uint64_t sE01[K3_T + 1];
          __shared__ uint32_t sEs[K3_T + 1];
          const uint32_t NG = gp[8];
          const bool ok = tid < NG;
          uint64_t wa = 0, wb = 0;
          if (ok) {
            wa = ld_word(&a.gwa[tid]); wb = ld_word(&a.gwb[tid]);
            wa = wait_word(&a.gwa[tid], 38, epoch, wa, ctl);
            wb = wait_word(&a.gwb[tid], 38, epoch, wb, ctl);
          }
          const uint64_t v01 = ok ? ((wa & 0x7FFFFu) | (((wa >> 19) & 0x7FFFFu) << 32)) : 0ull;
          const uint32_t vs = ok ? (uint32_t)(wb & 0x7FFFFu) : 0u;
          uint64_t t01;
          uint32_t ts;
          const uint64_t e01 = block_excl_scan_sum64<K3_T>(v01, &t01);
          const uint32_t es = block_excl_scan_sum<K3_T>(vs, &ts);
          sE01[tid] = e01; sEs[tid] = es;
          if (tid == 0) { sE01[K3_T] = t01; sEs[K3_T] = ts; }
          __syncthreads();
          auto at01 = [&](uint32_t g) -> uint64_t { return g < NG ? sE01[g] : sE01[K3_T]; };
          auto ats = [&](uint32_t g) -> uint32_t { return g < NG ? sEs[g] : sEs[K3_T]; };
          if (ok) {
            uint32_t q = 0;
#pragma unroll
            for (int k = 1; k < 8; ++k) q += (tid >= gp[k]) ? 1u : 0u;
            const uint64_t b01 = at01(gp[q]);
            a.goff[(size_t)tid * 4 + 0] = (uint32_t)e01 - (uint32_t)b01;
            a.goff[(size_t)tid * 4 + 1] = (uint32_t)(e01 >> 32) - (uint32_t)(b01 >> 32);
            a.goff[(size_t)tid * 4 + 2] = es - ats(gp[q]);
          }
          if (tid < 8) {
            const uint64_t lo = at01(gp[tid]), hi = at01(gp[tid + 1]);
            s_pt[tid][0] = (uint32_t)hi - (uint32_t)lo;
            s_pt[tid][1] = (uint32_t)(hi >> 32) - (uint32_t)(lo >> 32);
            s_pt[tid][2] = ats(gp[tid + 1]) - ats(gp[tid]);
          }
        } else
        for (uint32_t q = 0; q < 8; ++q) {
          uint32_t c0 = 0, c1 = 0, cs = 0;
          for (uint32_t base = gp[q]; base < gp[q + 1]; base += K3_T) {
            const uint32_t i = base + tid;
            const bool ok = i < gp[q + 1];
            uint64_t wa = 0, wb = 0;
            if (ok) {
              wa = ld_word(&a.gwa[i]); wb = ld_word(&a.gwb[i]);
              wa = wait_word(&a.gwa[i], 38, epoch, wa, ctl);
              wb = wait_word(&a.gwb[i], 38, epoch, wb, ctl);
            }
            const uint64_t v01 = ok ? ((wa & 0x7FFFFu) | (((wa >> 19) & 0x7FFFFu) << 32)) : 0ull;
            const uint32_t vs = ok ? (uint32_t)(wb & 0x7FFFFu) : 0u;
            uint64_t t01;
            uint32_t ts;
            const uint64_t e01 = block_excl_scan_sum64<K3_T>(v01, &t01);
            const uint32_t es = block_excl_scan_sum<K3_T>(vs, &ts);
            if (ok) {
              a.goff[(size_t)i * 4 + 0] = c0 + (uint32_t)e01;
              a.goff[(size_t)i * 4 + 1] = c1 + (uint32_t)(e01 >> 32);
              a.goff[(size_t)i * 4 + 2] = cs + es;
            }
            c0 += (uint32_t)t01; c1 += (uint32_t)(t01 >> 32); cs += ts;
          }
          if (tid == 0) { s_pt[q][0] = c0; s_pt[q][1] = c1; s_pt[q][2] = cs; }
        }
        __syncthreads();
        if (tid == 0) {
          uint64_t symsum = 0, nextn = 0, curn = 0;
          uint32_t want_list = 0;
          bool ovf = false;
          for (int q = 0; q < 8; ++q) {
            symsum += s_pt[q][2];
            nextn += (uint64_t)s_pt[q][0] + s_pt[q][1];
            curn += (uint64_t)cn[q][0] + cn[q][1];
            if ((uint64_t)s_pt[q][0] + s_pt[q][1] > list_cap(a, a.par ^ 1u)) ovf = true;
            if (s_pt[q][0] + s_pt[q][1] > want_list) want_list = s_pt[q][0] + s_pt[q][1];
          }
          const uint64_t sym0 = ctl->sym_total;
          if (ovf) { ctl->overflow = 1; ctl->skip_round = a.round; ctl->want_list = want_list; }
          else if (sym0 + symsum > ctl->sym_cap) {
            for (int q = 0; q < 8; ++q) ctl->ptot[q][2] = s_pt[q][2];     // (per plane: a round that is too large for one flush goes plane group by plane group)
            ctl->need_flush = 1; ctl->skip_round = a.round; ctl->want_syms = symsum;
          }
          else {
            uint64_t acc = sym0;
            for (uint32_t q = 0; q < 8; ++q) {
              const uint32_t qn = (q + 1u) & 7u;
              ctl->symbase[q] = acc;
              ctl->cnt[a.par ^ 1u][qn][0] = s_pt[q][0];
              ctl->cnt[a.par ^ 1u][qn][1] = s_pt[q][1];
              RunEntry e; e.start = acc; e.count = s_pt[q][2]; e.round = a.round;
              a.runs[(size_t)a.run_slot * 8 + q] = e;
              acc += s_pt[q][2];
            }
            ctl->sym_total = acc;
            atomicAdd((unsigned long long *)&ctl->nodes_total, (unsigned long long)curn);
            ctl->next_nodes = nextn;
            if (nextn == 0 && ctl->done_round == 0xFFFFFFFFu) ctl->done_round = a.round + 1u;
          }
        }
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------
// Narrow rounds in ONE launch.  A dependent kernel costs ~8 us on this GPU whatever it does, so count / scan /
// write spend ~30 us on a round of a few thousand nodes.  Here every block takes one tile (dynamic ticket =
// look-back order), classifies it once, publishes its three counts in one 64-bit word tagged with the round
// (agent-scope store; nothing to reset), waits for the words of ALL earlier tiles (at most K3_SMALL_MAXTILES,
// one per thread per sweep -- earlier tickets belong to blocks that are running or done, so this cannot
// deadlock), sums them into its offsets and writes.  The block with the last ticket waits for every tile's word
// and folds them into the control block exactly as k3_scan_kernel's epilogue does.  Whether the symbol buffer could overflow is decided
// up front from the node count (a node codes at most one symbol), identically in every block; a round that does
// not fit the grid or the tile table sets small_bail and is run again by the wide kernels.
// ------------------------------------------------------------------------------------------------------
template <bool SCAN>
__global__ __launch_bounds__(K3_T) void k3_small_kernel(K3Args a, unsigned long long *words) {
  __shared__ uint32_t tp[9], cn[8][2];
  __shared__ uint32_t lds_cnt[K3_NPT][4][3];
  __shared__ unsigned long long s_acc[3];
  __shared__ unsigned long long s_tot[8];
  EnumCtl *ctl = a.ctl;
  if (ctl->need_flush || ctl->overflow || ctl->small_bail) return;
  const uint32_t tid = threadIdx.x;
  if (tid == 0) tile_prefix(a, tp, cn);
  if (tid < 3) s_acc[tid] = 0;
  if (tid < 8) s_tot[tid] = 0;
  __syncthreads();
  const uint32_t T = tp[8];
  // decisions every block takes identically from the state the previous launch left
  uint64_t M = 0;
  bool fits = T <= gridDim.x && T <= K3_SMALL_MAXTILES;      // (<= 2048 tiles: per-plane totals stay below 2^21)
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const uint64_t m = (uint64_t)ctl->cnt[a.par][q][0] + ctl->cnt[a.par][q][1];
    M += m;
    if (2 * m > list_cap(a, a.par ^ 1u)) fits = false;       // children could overflow a list: let the wide path decide
  }
  if (M >= (1u << 21)) fits = false;
  const uint64_t sym0 = ctl->sym_total;
  if (!fits) {
    if (blockIdx.x == 0 && tid == 0) { ctl->skip_round = a.round; ctl->small_bail = 1; }
    return;
  }
  if (sym0 + M > ctl->sym_cap) {
    if (blockIdx.x == 0 && tid == 0) { ctl->skip_round = a.round; ctl->want_syms = M; ctl->need_flush = 1; }
    return;
  }
  // tile = block index.  (A ticket drawn from one counter made the look-back order independent of the dispatch order, but
  // one address takes ~88 atomics per us: 2000 tiles = 23 us before anything else happens.  Like k3_count2_kernel this
  // now relies on workgroups being dispatched in index order per XCD: a block only ever waits for smaller indices, the
  // smallest index that is not running yet never waits for a block that is not, so a slot always frees up for it.)
  const uint32_t tile = blockIdx.x;
  const uint64_t epoch = (uint64_t)((a.round + 1u) & 0x7FFFFFFFu);
  if (tile < T) {
    uint32_t p = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) p += (tile >= tp[k]) ? 1u : 0u;
    const uint32_t lane = tid & 63u, w = tid >> 6;
    TileOut t;
    k3_classify<true, SCAN>(a, p, tile - tp[p], cn[p][0], cn[p][1], t);
    uint32_t pre0[K3_NPT], pre1[K3_NPT], pres[K3_NPT];
    const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int it = 0; it < K3_NPT; ++it) {
      const uint64_t b0 = __ballot(t.has0[it]), b1 = __ballot(t.has1[it]), bs = __ballot(t.hassym[it]);
      pre0[it] = (uint32_t)__popcll(b0 & lt);
      pre1[it] = (uint32_t)__popcll(b1 & lt);
      pres[it] = (uint32_t)__popcll(bs & lt);
      if (lane == 0) {
        lds_cnt[it][w][0] = (uint32_t)__popcll(b0);
        lds_cnt[it][w][1] = (uint32_t)__popcll(b1);
        lds_cnt[it][w][2] = (uint32_t)__popcll(bs);
      }
    }
    __syncthreads();
    if (tid == 0) {
      uint64_t tt[3] = {0, 0, 0};
      for (int it = 0; it < K3_NPT; ++it)
        for (int ww = 0; ww < 4; ++ww)
          for (int j = 0; j < 3; ++j) tt[j] += lds_cnt[it][ww][j];
      // [10:0] child0, [21:11] child1, [32:22] symbols (<= 1024 each), [63:33] round tag
      __hip_atomic_store(&words[tile], (epoch << 33) | (tt[2] << 22) | (tt[1] << 11) | tt[0], __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    // look back: all earlier tiles give the symbol offset, those of my plane the child offsets
    // (all of a thread's words -- at most K3_SMALL_MAXTILES / K3_T = 8 -- are requested together and only the ones that
    //  are not there yet are polled again: one L2 round trip for the look-back, not one per word)
    uint64_t c0 = 0, c1 = 0, cs = 0;
    {
      constexpr int LB = (int)(K3_SMALL_MAXTILES / K3_T);
      uint64_t wv[LB];
#pragma unroll
      for (int q = 0; q < LB; ++q) { const uint32_t j = tid + (uint32_t)q * K3_T; wv[q] = j < tile ? ld_word(&words[j]) : 0ull; }
#pragma unroll
      for (int q = 0; q < LB; ++q) {
        const uint32_t j = tid + (uint32_t)q * K3_T;
        if (j < tile) {
          wv[q] = wait_word(&words[j], 33, epoch, wv[q], ctl);
          cs += (wv[q] >> 22) & 0x7FFu;
          if (j >= tp[p]) { c0 += wv[q] & 0x7FFu; c1 += (wv[q] >> 11) & 0x7FFu; }
        }
      }
    }
    if (c0) atomicAdd(&s_acc[0], (unsigned long long)c0);
    if (c1) atomicAdd(&s_acc[1], (unsigned long long)c1);
    if (cs) atomicAdd(&s_acc[2], (unsigned long long)cs);
    __syncthreads();
    k3_place<SCAN>(a, p, t, pre0, pre1, pres, lds_cnt, (uint32_t)s_acc[0], (uint32_t)s_acc[1], sym0 + s_acc[2]);
  }
  // ---- the last block folds the round into the control block ----
  // (it waits for every tile's word below, so every block WITH a tile has read what it reads from the control block; the
  //  blocks without one only read fields this epilogue does not change -- flags, this parity's counts -- or ignore them)
  if (tile != gridDim.x - 1u) return;
  {
    constexpr int LB = (int)(K3_SMALL_MAXTILES / K3_T);
    uint64_t wv[LB];
#pragma unroll
    for (int q = 0; q < LB; ++q) { const uint32_t j = tid + (uint32_t)q * K3_T; wv[q] = j < T ? ld_word(&words[j]) : 0ull; }
#pragma unroll
    for (int q = 0; q < LB; ++q) {
      const uint32_t j = tid + (uint32_t)q * K3_T;
      if (j < T) {
        const uint32_t p = tile_plane(tp, j);
        wv[q] = wait_word(&words[j], 33, epoch, wv[q], ctl);
        // per-plane totals fit 21-bit fields: a plane's children / symbols number at most its nodes < 2^21 (checked: M)
        atomicAdd(&s_tot[p], (unsigned long long)((wv[q] & 0x7FFu) | (((wv[q] >> 11) & 0x7FFu) << 21) | (((wv[q] >> 22) & 0x7FFu) << 42)));
      }
    }
  }
  __syncthreads();
  if (tid != 0) return;
  uint64_t acc = sym0, nextn = 0;
  for (uint32_t q = 0; q < 8; ++q) {
    const uint32_t qn = (q + 1u) & 7u;
    const uint32_t t0 = (uint32_t)(s_tot[q] & 0x1FFFFFu), t1 = (uint32_t)((s_tot[q] >> 21) & 0x1FFFFFu), ts = (uint32_t)(s_tot[q] >> 42);
    ctl->cnt[a.par ^ 1u][qn][0] = t0;
    ctl->cnt[a.par ^ 1u][qn][1] = t1;
    RunEntry e; e.start = acc; e.count = ts; e.round = a.round;
    a.runs[(size_t)a.run_slot * 8 + q] = e;
    acc += ts;
    nextn += (uint64_t)t0 + t1;
  }
  ctl->sym_total = acc;
  atomicAdd((unsigned long long *)&ctl->nodes_total, (unsigned long long)M);
  ctl->next_nodes = nextn;
  if (nextn == 0 && ctl->done_round == 0xFFFFFFFFu) ctl->done_round = a.round + 1u;
}

// ------------------------------------------------------------------------------------------------------
// Persistent tail: ONE workgroup keeps every live node (<= K3_TAIL_CAP over all 8 planes) in LDS and loops
// over rounds on the device -- no launches, no host round trips.  Rounds = longest repeated bit string, so
// inputs with long repeats have 10^4..10^6 rounds that each hold a handful of nodes (SURVEY section 7,
// "round tail"); a launch per round would dominate everything.  Per round: classify (global rank gathers),
// one packed block scan (child0 | child1 | symbol counts in 16-bit fields of a u64), scatter children into
// the other LDS buffer, symbols and the per-(round, plane) run table to HBM.  The kernel leaves when nothing
// is left, when the node count grows past the LDS capacity, when the symbol buffer cannot take another
// round, or after max_rounds; state is handed back through the same control block the wide rounds use.
// ------------------------------------------------------------------------------------------------------
constexpr int KT_T = 256;                    // 4 waves: cheap barriers, no 128-VGPR cap (a 1024-thread version spilled)
constexpr int KT_NPT = K3_TAIL_CAP / KT_T;   // 4 consecutive nodes per thread

__device__ __forceinline__ uint64_t kt_block_excl_scan(uint64_t v, uint64_t *ws, uint64_t *total) {
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  const uint64_t inc = wave_incl_sum64(v);
  if (lane == 63) ws[wid] = inc;
  __syncthreads();
  uint64_t wbase = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < KT_T / 64; ++i) {
    const uint64_t t = ws[i];
    if ((uint32_t)i < wid) wbase += t;
    tot += t;
  }
  __syncthreads();
  *total = tot;
  return wbase + inc - v;
}

template <bool SCAN>
__global__ __launch_bounds__(KT_T) void k3_tail_kernel(K3Args a, RunEntry *truns, uint32_t max_rounds) {
  __shared__ Node buf[2][K3_TAIL_CAP];
  __shared__ uint32_t cnt[2][8][2];
  __shared__ uint32_t off[9], noff[9];
  __shared__ uint64_t ws[KT_T / 64];
  __shared__ uint64_t pstart[9];
  EnumCtl *ctl = a.ctl;
  if (ctl->need_flush || ctl->overflow) return;
  const uint32_t tid = threadIdx.x;
  uint32_t par = a.par, round = a.round, cur = 0, executed = 0;
  if (tid < 16) cnt[par][tid >> 1][tid & 1] = ctl->cnt[par][tid >> 1][tid & 1];
  __syncthreads();
  if (tid == 0) {
    uint32_t acc = 0;
    for (int p = 0; p < 8; ++p) { off[p] = acc; acc += cnt[par][p][0] + cnt[par][p][1]; }
    off[8] = acc;
  }
  __syncthreads();
  uint32_t total = off[8];
  if (total > K3_TAIL_CAP) { if (tid == 0) ctl->tail_rounds = 0; return; }
  for (uint32_t q = tid; q < total; q += KT_T) {
    uint32_t p = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) p += (q >= off[k]) ? 1u : 0u;
    const uint32_t i = q - off[p], c0 = cnt[par][p][0];
    buf[0][q] = plane_nodes(a, par, p)[i < c0 ? i : (list_cap(a, par) - 1u - (i - c0))];
  }
  uint64_t sym_total = ctl->sym_total, nodes_total = ctl->nodes_total;
  const uint64_t sym_cap = ctl->sym_cap;
  uint32_t flush = 0, done_round = ctl->done_round;
  __syncthreads();
  for (;;) {
    if (total == 0) { if (done_round == 0xFFFFFFFFu) done_round = round ? round : 1u; break; }
    if (executed >= max_rounds) break;
    if (sym_total + total > sym_cap) { flush = 1; break; }
    // ---- classify my KT_NPT consecutive nodes ----
    Node nd[KT_NPT];
    uint32_t pl[KT_NPT];
    bool valid[KT_NPT];
    uint32_t ga[KT_NPT], gb[KT_NPT], gm[KT_NPT];
    Granule qa[KT_NPT], qb[KT_NPT], qm[KT_NPT];
#pragma unroll
    for (int it = 0; it < KT_NPT; ++it) {
      const uint32_t q = tid * KT_NPT + (uint32_t)it;
      valid[it] = q < total;
      nd[it] = valid[it] ? buf[cur][q] : Node{0u, 1u, 1u};
      uint32_t p = 0;
#pragma unroll
      for (int k = 1; k < 8; ++k) p += (q >= off[k]) ? 1u : 0u;
      pl[it] = valid[it] ? p : 0u;
      const Granule *G = a.gran + (size_t)pl[it] * a.ngran;
      ga[it] = div96(nd[it].s);
      gb[it] = div96(nd[it].s + nd[it].x0 + nd[it].x1);
      gm[it] = div96(nd[it].s + nd[it].x0);
      qa[it] = G[valid[it] ? ga[it] : 0u];
      qb[it] = G[valid[it] ? gb[it] : 0u];
    }
    NodePre pr[KT_NPT];
#pragma unroll
    for (int it = 0; it < KT_NPT; ++it) {
      const uint32_t rs = granule_rank1(qa[it], nd[it].s - ga[it] * 96u);
      const uint32_t re = granule_rank1(qb[it], nd[it].s + nd[it].x0 + nd[it].x1 - gb[it] * 96u);
      node_pre(nd[it], rs, re, pr[it]);
      qm[it] = gm[it] == ga[it] ? qa[it] : qb[it];
      if (valid[it] && pr[it].kind == 3u && gm[it] != ga[it] && gm[it] != gb[it])
        qm[it] = (a.gran + (size_t)pl[it] * a.ngran)[gm[it]];
    }
    StepOut so[KT_NPT];
    uint64_t mine = 0;
#pragma unroll
    for (int it = 0; it < KT_NPT; ++it) {
      so[it].has0 = so[it].has1 = so[it].hassym = 0;
      if (valid[it]) {
        const uint32_t rm = granule_rank1(qm[it], nd[it].s + nd[it].x0 - gm[it] * 96u);
        node_post(nd[it], a.zeros[pl[it]], pr[it], rm, so[it]);
      }
      mine += (uint64_t)so[it].has0 | ((uint64_t)so[it].has1 << 16) | ((uint64_t)so[it].hassym << 32);
    }
    // ---- one packed scan: [15:0] child0, [31:16] child1, [47:32] symbols ----
    uint64_t tot;
    uint64_t ex = kt_block_excl_scan(mine, ws, &tot);
    if (tid < 9) pstart[tid] = tot;                       // default for planes with no nodes: fixed up below
    __syncthreads();
    {
      uint64_t run = ex;
#pragma unroll
      for (int it = 0; it < KT_NPT; ++it) {
        const uint32_t q = tid * KT_NPT + (uint32_t)it;
        if (valid[it] && q == off[pl[it]]) pstart[pl[it]] = run;     // exclusive prefix at the plane's first node
        run += (uint64_t)so[it].has0 | ((uint64_t)so[it].has1 << 16) | ((uint64_t)so[it].hassym << 32);
      }
    }
    __syncthreads();
    if (tid == 0) {
      for (int p = 7; p >= 0; --p) if (off[p] == off[p + 1]) pstart[p] = pstart[p + 1];
      // next-round geometry: plane pn receives the children of plane (pn + 7) % 8
      uint32_t acc = 0;
      for (int pn = 0; pn < 8; ++pn) {
        const int p = (pn + 7) & 7;
        const uint64_t d = pstart[p + 1] - pstart[p];
        const uint32_t t0 = (uint32_t)(d & 0xFFFFu), t1 = (uint32_t)((d >> 16) & 0xFFFFu);
        cnt[par ^ 1u][pn][0] = t0;
        cnt[par ^ 1u][pn][1] = t1;
        noff[pn] = acc;
        acc += t0 + t1;
      }
      noff[8] = acc;
    }
    __syncthreads();
    if (noff[8] > K3_TAIL_CAP) break;                      // the children would not fit in LDS: hand this round to the wide kernels
    if (tid < 8) {                                         // run table row of this round
      RunEntry e;
      e.start = sym_total + ((pstart[tid] >> 32) & 0xFFFFu);
      e.count = (uint32_t)(((pstart[tid + 1] - pstart[tid]) >> 32) & 0xFFFFu);
      e.round = round;
      truns[(size_t)executed * 8 + tid] = e;
    }
    // ---- scatter children (LDS) and symbols (HBM) ----
    {
      uint64_t run = ex;
      const PlaneCfg *cfgs = a.cfg;
#pragma unroll
      for (int it = 0; it < KT_NPT; ++it) {
        if (valid[it]) {
          const uint32_t p = pl[it], pn = (p + 1u) & 7u;
          const uint64_t rel = run - pstart[p];
          const uint32_t t0 = cnt[par ^ 1u][pn][0];
          if (so[it].has0) buf[cur ^ 1u][noff[pn] + (uint32_t)(rel & 0xFFFFu)] = so[it].c0;
          if (so[it].has1) buf[cur ^ 1u][noff[pn] + t0 + (uint32_t)((rel >> 16) & 0xFFFFu)] = so[it].c1;
          if (so[it].hassym) {
            const uint64_t si = sym_total + ((run >> 32) & 0xFFFFu);
            if (SCAN) {
              a.scanrec[si] = scan_pack(so[it].sym, so[it].k, so[it].ctx1, so[it].ctx2, so[it].ctxs);
            } else {
              uint32_t kw, ew;
              pack_symbol(cfgs[p], p, so[it].sym, so[it].k, so[it].ctx1, so[it].ctx2, so[it].ctxs, kw, ew);
              a.symkey[si] = kw;
              a.symesc[si] = ew;
            }
          }
        }
        run += (uint64_t)so[it].has0 | ((uint64_t)so[it].has1 << 16) | ((uint64_t)so[it].hassym << 32);
      }
    }
    sym_total += (tot >> 32) & 0xFFFFu;
    nodes_total += total;
    __syncthreads();
    if (tid < 9) off[tid] = noff[tid];
    __syncthreads();
    total = off[8];
    par ^= 1u; cur ^= 1u; ++round; ++executed;
  }
  // ---- hand the state back ----
  __syncthreads();
  for (uint32_t q = tid; q < total && total <= K3_TAIL_CAP; q += KT_T) {
    uint32_t p = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) p += (q >= off[k]) ? 1u : 0u;
    const uint32_t i = q - off[p], c0 = cnt[par][p][0];
    plane_nodes(a, par, p)[i < c0 ? i : (list_cap(a, par) - 1u - (i - c0))] = buf[cur][q];
  }
  if (tid < 16) ctl->cnt[par][tid >> 1][tid & 1] = cnt[par][tid >> 1][tid & 1];
  if (tid == 0) {
    ctl->sym_total = sym_total;
    ctl->nodes_total = nodes_total;
    ctl->next_nodes = total;
    ctl->done_round = done_round;
    ctl->tail_rounds = executed;
    if (flush) { ctl->need_flush = 1; ctl->skip_round = round; ctl->want_syms = total; }
  }
}

K3Args k3_make_args(bce_hip_ctx *c, uint32_t round, uint32_t run_slot) {
  K3Args a;
  a.ctl = c->ctl.as<EnumCtl>();
  a.nodes[0] = c->nlist[0].as<Node>(); a.nodes[1] = c->nlist[1].as<Node>();
  a.gran = c->gran.as<Granule>();
  a.cfg = c->dcfg.as<PlaneCfg>();
  a.symkey = c->skey[0].as<uint32_t>();
  a.symesc = c->sesc.as<uint32_t>();
  a.scanrec = c->scanrec.as<uint32_t>();
  a.tilecnt = c->tilecnt.as<uint32_t>();
  a.tileoff = c->tileoff.as<uint32_t>();
  a.runs = c->runs.as<RunEntry>();
  a.tw = c->k3tw.as<unsigned long long>();
  {
    // group arrays carved from one buffer: words a | words b | offsets[4]
    const size_t ng = c->k3_groups;
    uint8_t *base = c->k3grp.as<uint8_t>();
    a.gwa = reinterpret_cast<unsigned long long *>(base);
    a.gwb = reinterpret_cast<unsigned long long *>(base + ng * 8);
    a.goff = reinterpret_cast<uint32_t *>(base + ng * 16);
  }
  a.fused = 0;
  a.cap[0] = c->capL[0]; a.cap[1] = c->capL[1]; a.ngran = c->ngran; a.n = c->n;
  for (int i = 0; i < 8; ++i) a.zeros[i] = c->zeros[i];
  a.par = round & 1u; a.round = round; a.run_slot = run_slot;
  a.repeat = 0;
  {
    static const char *e = getenv("BCE_HIP_CAP32");             // (test knob: the 64-bit list indexing from this many nodes per list on)
    static const uint32_t v = e ? (uint32_t)strtoul(e, nullptr, 10) : K3_CAP32;
    a.cap32 = v < K3_CAP32 ? v : K3_CAP32;
  }
  a.pmask = (c->scan_mode || !c->coder) ? 0xFFu : (c->coder->plane_mask & 0xFFu);
  return a;
}

// The largest list an input of n bytes can need: n/2 nodes per plane-round (disjoint intervals of width >= 2; text peaks at
// 0.06-0.09 n, random bytes at ~0.3 n).  Two consecutive rounds of a plane hold at most 3n/4 nodes together (n/4 nodes of width
// 4 with n/2 children of width 2), i.e. 72 n bytes in all sixteen lists: 155 GB at n = 2^31 - 1, beside the 13 n bytes of text,
// BWT, planes, suffix array and inverse that stay -- K1's sort scratch goes back first when that is what it takes (ctx_trim).
static uint32_t full_capP(uint32_t n) { return (uint32_t)((uint64_t)n / 2 + 2); }

// What a compression STARTS with: n / 8 nodes per list (at n / 2 + 2, the worst case, the lists were 55 % of a context's 17.6 GB at
// 10^8 bytes, twenty times what text ever fills; memory the driver has to clear before it hands it out -- 27 ms per GB whenever an earlier process has used
// it), or whatever a buffer the context has already holds.  A round that does not fit is not run (the epilogue checks before
// anything is written); the host replaces the lists the round would WRITE -- the other parity's, which hold nothing -- by larger
// ones and runs it again (k3_grow_lists).  BCE_HIP_CAPP_DIV=d / test knob 12: start with n / d + 4096 nodes per list.
static uint32_t initial_capP(bce_hip_ctx *c, uint32_t n, uint32_t full, int par) {
  uint64_t div = 8;
  bool forced = false;                                         // (a test asks for small lists: also in a context that holds larger ones)
  if (c->dbg_capp_div) { div = c->dbg_capp_div; forced = true; }
  if (const char *e = getenv("BCE_HIP_CAPP_DIV")) { const uint64_t v = strtoull(e, nullptr, 10); if (v >= 1) { div = v; forced = true; } }
  uint64_t cap = (uint64_t)n / div + 4096;
  const uint64_t held = c->nlist[par].cap / (8 * sizeof(Node));
  if (held > cap && !forced) cap = held;
  return (uint32_t)(cap < full ? cap : full);
}

// everything that is sized by the lists' capacity (the larger of the two parities')
static int k3_size_tile_arrays(bce_hip_ctx *c) {
  const uint32_t capm = c->capL[0] > c->capL[1] ? c->capL[0] : c->capL[1];
  const size_t tiles = (size_t)8 * ((capm + K3_TILE - 1) / K3_TILE) + 8;
  BCE_TRY(ensure(c, c->tilecnt, tiles * 16));
  BCE_TRY(ensure(c, c->tileoff, tiles * 16));
  // two-launch rounds: a count word per tile; per group of 256 tiles: two words and four offsets (32 B)
  c->k3_groups = tiles / 256 + 16;
  BCE_TRY(ensure(c, c->k3tw, tiles * 8));
  BCE_TRY(ensure(c, c->k3grp, c->k3_groups * 32 + 64));
  BCE_HIP_TRY(c, hipMemsetAsync(c->k3tw.p, 0, tiles * 8, c->stream));               // round tags of an earlier compression (or of whoever had the memory)
  BCE_HIP_TRY(c, hipMemsetAsync(c->k3grp.p, 0, c->k3_groups * 32 + 64, c->stream));
  return BCE_HIP_OK;
}

// A round found its children would not fit the lists (ctl.overflow, ctl.skip_round = c->round, ctl.want_list = the longest list
// it would write: nothing of it was written).  The lists it writes are the OTHER parity's, and those hold nothing now -- their
// nodes were this round's parents one round ago --, so they are given back and allocated anew, no copy and never old and new
// side by side: twice what the round needs (the next rounds of a growing enumeration need more), at most the worst case, and
// exactly what it needs (+ 1/16) when the device cannot give more.  BCE_HIP_E_NOMEM only when even that does not fit.
int k3_grow_lists(bce_hip_ctx *c, const EnumCtl &ctl) {
  const uint32_t full = full_capP(c->n);
  const int out = (int)((c->round & 1u) ^ 1u);
  const uint64_t need = ctl.want_list;
  if (need <= c->capL[out] || need > full) {
    snprintf(c->err, sizeof c->err, "k3: round %u reports a list of %llu nodes (lists hold %u, an input of %u bytes needs at most %u)", c->round,
             (unsigned long long)need, c->capL[out], c->n, full);
    return BCE_HIP_E_INTERNAL;
  }
  BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (hipStream_t st : {c->k4_stream, c->copy_stream}) if (st) BCE_HIP_TRY(c, hipStreamSynchronize(st));
  release(c->nlist[out]);
  c->capL[out] = 0;
  uint64_t want = need * 2 + 4096;
  if (want > full) want = full;
  int rc = ensure(c, c->nlist[out], (size_t)8 * want * sizeof(Node));
  if (rc == BCE_HIP_E_NOMEM) {
    want = need + need / 16 + 4096;
    if (want > full) want = full;
    rc = ensure(c, c->nlist[out], (size_t)8 * want * sizeof(Node));
  }
  if (rc != BCE_HIP_OK) {
    if (rc == BCE_HIP_E_NOMEM) snprintf(c->err, sizeof c->err, "k3: no device memory for node lists of %llu nodes (round %u, n = %u)", (unsigned long long)want, c->round, c->n);
    return rc;
  }
  c->capL[out] = (uint32_t)want;
  c->stats.list_grows += 1.0;
  c->stats.list_nodes = c->capL[0] > c->capL[1] ? c->capL[0] : c->capL[1];
  if (getenv("BCE_ALLOC_TRACE")) fprintf(stderr, "k3: round %u needs lists of %llu nodes: parity %d now holds %u per list, parity %d %u (n = %u)\n", c->round, (unsigned long long)need, out, c->capL[out], out ^ 1, c->capL[out ^ 1], c->n);
  BCE_TRY(k3_size_tile_arrays(c));
  BCE_HIP_TRY(c, hipMemsetAsync(&c->ctl.as<EnumCtl>()->overflow, 0, sizeof(uint32_t), c->stream));
  return BCE_HIP_OK;
}

uint64_t k3_symbol_capacity(const bce_hip_ctx *c, uint32_t n) {
  uint64_t cap = c->sym_cap_user;
  if (!cap) {
    // default flush granularity: 16M records (pipelines with the host coders); a round that needs more
    // grows the buffer on demand (k3_grow_symbols)
    const char *env = getenv("BCE_HIP_FLUSH_RECORDS");
    const uint64_t want = (uint64_t)8 * n + 1024, soft = env ? strtoull(env, nullptr, 10) : ((uint64_t)1 << 24);
    cap = want < soft ? want : soft;
  }
  return cap;
}

int k3_begin(bce_hip_ctx *c) {
  const uint32_t n = c->n;
  for (int par = 0; par < 2; ++par) {
    // (a buffer the decoder left in nlist[0] holds both of ITS parities: as one parity's lists here it is simply larger)
    c->capL[par] = initial_capP(c, n, full_capP(n), par);
    BCE_TRY(ensure(c, c->nlist[par], (size_t)8 * c->capL[par] * sizeof(Node)));
  }
  c->stats.list_grows = 0; c->stats.list_nodes = c->capL[0] > c->capL[1] ? c->capL[0] : c->capL[1];
  BCE_TRY(k3_size_tile_arrays(c));
  BCE_TRY(ensure(c, c->ctl, sizeof(EnumCtl)));
  BCE_TRY(ensure(c, c->runs, (size_t)K3_MAXBATCH * 8 * sizeof(RunEntry)));
  if (!c->h_ctl) BCE_TRY(pin_alloc(c, &c->h_ctl, sizeof(EnumCtl)));
  if (!c->h_runs) BCE_TRY(pin_alloc(c, &c->h_runs, (size_t)K3_MAXBATCH * 8 * sizeof(RunEntry)));
  // symbol buffer capacity
  const uint64_t cap = k3_symbol_capacity(c, n);
  c->sym_cap = cap;
  BCE_TRY(ensure(c, c->skey[0], (size_t)cap * 4));
  BCE_TRY(ensure(c, c->sesc, (size_t)cap * 4));
  if (c->overlap && !c->scan_mode) {                           // the pair K4 reads while K3 fills the other (common.h: k4_stream)
    BCE_TRY(ensure(c, c->skey_alt, (size_t)cap * 4));
    BCE_TRY(ensure(c, c->sesc_alt, (size_t)cap * 4));
  }
  if (c->scan_mode) BCE_TRY(ensure(c, c->scanrec, (size_t)cap * 4));
  // roots: (0, C[i], n - C[i]) with C[i] = zeros(plane (i+7)%8), only where both are non-zero (bce.cpp:1237-1240)
  EnumCtl ctl;
  memset(&ctl, 0, sizeof ctl);
  ctl.sym_cap = cap;
  ctl.done_round = 0xFFFFFFFFu;
  for (int i = 0; i < 8; ++i) {
    const uint32_t C = c->zeros[(i + 7) & 7];
    if (C && n - C) {
      Node root = {0u, C, n - C};
      BCE_HIP_TRY(c, hipMemcpyAsync(c->nlist[0].as<Node>() + (size_t)i * c->capL[0], &root, sizeof root,
                                    hipMemcpyHostToDevice, c->stream));
      BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));   // `root` is a stack temporary
      ctl.cnt[0][i][0] = 1;
    }
  }
  BCE_HIP_TRY(c, hipMemcpyAsync(c->ctl.p, &ctl, sizeof ctl, hipMemcpyHostToDevice, c->stream));
  BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
  // round tags of an earlier compression must not look current
  BCE_TRY(ensure(c, c->smwords, (size_t)K3_SMALL_MAXTILES * 8));
  BCE_HIP_TRY(c, hipMemsetAsync(c->smwords.p, 0, (size_t)K3_SMALL_MAXTILES * 8, c->stream));
  c->round = 0;
  c->enum_active = true;
  for (int p = 0; p < 8; ++p) c->run_log[p].clear();
  c->stats.k3_ms = 0; c->stats.k3_launches = 0;
  return BCE_HIP_OK;
}

int k3_rounds(bce_hip_ctx *c, uint32_t count, uint64_t nodes_hint) {
  if (count > K3_MAXBATCH) count = K3_MAXBATCH;
  // nodes_hint = 0: unknown / growing (ramp-up: the count doubles every round) -> full grid.  Otherwise the caller
  // has seen the count stop growing, so the hint bounds the coming rounds and narrow rounds do not pay for 2048
  // idle blocks per launch (tiles are grid-strided: the size only affects speed).
  uint64_t want = nodes_hint ? (nodes_hint * 2 + K3_TILE - 1) / K3_TILE + 16 : 2048;
  const uint32_t grid = (uint32_t)(want < 2048 ? want : 2048);
  // The two-launch round's group owners spin on tiles of OTHER blocks, which is only free of stalls while every block of
  // the grid is resident.  A model flush running beside us (k4_stream) takes CU slots away: those batches use the three
  // launches (count, scan, write), which wait for nothing.
  const bool fused = !c->dbg_no_fused && !k4_in_flight(c);
  // k3_count2_kernel's group owners wait for tiles of other blocks: keep every block resident (a block that is not
  // yet running would only start when a running one EXITS, and an owner waiting for it idles until then)
  if (fused && !c->k3_count2_grid) {
    int per_cu = 0, cus = 0;
    BCE_HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k3_count2_kernel<false>, K3_T, 0));
    BCE_HIP_TRY(c, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
    c->k3_count2_grid = (uint32_t)(per_cu > 0 && cus > 0 ? per_cu * cus : 1024);
  }
  const uint32_t grid2 = fused ? (grid < c->k3_count2_grid ? grid : c->k3_count2_grid) : grid;
  for (uint32_t i = 0; i < count; ++i) {
    K3Args a = k3_make_args(c, c->round + i, i);
    a.fused = fused ? 1u : 0u;
    if (fused) {
      if (c->scan_mode) hipLaunchKernelGGL((k3_count2_kernel<true>), dim3(grid2), dim3(K3_T), 0, c->stream, a);
      else hipLaunchKernelGGL((k3_count2_kernel<false>), dim3(grid2), dim3(K3_T), 0, c->stream, a);
    } else {
      hipLaunchKernelGGL((k3_tiles_kernel<false, false>), dim3(grid), dim3(K3_T), 0, c->stream, a);
      hipLaunchKernelGGL(k3_scan_kernel, dim3(8), dim3(1024), 0, c->stream, a);
    }
    if (c->scan_mode) hipLaunchKernelGGL((k3_tiles_kernel<true, true>), dim3(grid), dim3(K3_T), 0, c->stream, a);
    else hipLaunchKernelGGL((k3_tiles_kernel<true, false>), dim3(grid), dim3(K3_T), 0, c->stream, a);
  }
  BCE_HIP_TRY(c, hipGetLastError());
  c->stats.k3_launches += (fused ? 2.0 : 3.0) * count;
  return BCE_HIP_OK;
}

// ONE round (c->round) in three launches, recording the symbols of the planes in `mask` only; `repeat`: a further pass of a round
// that has been counted already.  For rounds whose symbols do not fit one model flush (api.hip, split_round): every pass
// classifies all nodes and writes all children (identical each time), the symbol records and the run table are the pass's own.
int k3_round_masked(bce_hip_ctx *c, uint32_t mask, bool repeat) {
  K3Args a = k3_make_args(c, c->round, 0);
  a.fused = 0;
  a.pmask &= mask;
  a.repeat = repeat ? 1u : 0u;
  const uint32_t grid = 2048;
  hipLaunchKernelGGL((k3_tiles_kernel<false, false>), dim3(grid), dim3(K3_T), 0, c->stream, a);
  hipLaunchKernelGGL(k3_scan_kernel, dim3(8), dim3(1024), 0, c->stream, a);
  if (c->scan_mode) hipLaunchKernelGGL((k3_tiles_kernel<true, true>), dim3(grid), dim3(K3_T), 0, c->stream, a);
  else hipLaunchKernelGGL((k3_tiles_kernel<true, false>), dim3(grid), dim3(K3_T), 0, c->stream, a);
  BCE_HIP_TRY(c, hipGetLastError());
  c->stats.k3_launches += 3.0;
  return BCE_HIP_OK;
}
int k3_clear_need_flush(bce_hip_ctx *c) {
  BCE_HIP_TRY(c, hipMemsetAsync(&c->ctl.as<EnumCtl>()->need_flush, 0, sizeof(uint32_t), c->stream));
  return BCE_HIP_OK;
}

// Narrow rounds, one launch each (k3_small_kernel).  The grid must cover the tiles of every queued round: the node
// count at most doubles per round while it grows and is taken as <= 1.5x the last known count once it decays; a
// round that does not fit bails out (small_bail) and the caller runs it with the wide kernels.
int k3_rounds_small(bce_hip_ctx *c, uint32_t count, uint64_t cur_nodes, bool growing) {
  if (count > K3_MAXBATCH) count = K3_MAXBATCH;
  BCE_TRY(ensure(c, c->smwords, (size_t)K3_SMALL_MAXTILES * 8));
  unsigned long long *words = c->smwords.as<unsigned long long>();
  for (uint32_t i = 0; i < count; ++i) {
    uint64_t bound = growing ? (cur_nodes << (i + 1 < 20 ? i + 1 : 20)) : cur_nodes + (cur_nodes >> 1);
    uint64_t tiles = (bound + K3_TILE - 1) / K3_TILE + 8;
    const uint32_t grid = (uint32_t)(tiles < K3_SMALL_MAXTILES ? tiles : K3_SMALL_MAXTILES);
    const K3Args a = k3_make_args(c, c->round + i, i);
    if (c->scan_mode) hipLaunchKernelGGL((k3_small_kernel<true>), dim3(grid), dim3(K3_T), 0, c->stream, a, words);
    else hipLaunchKernelGGL((k3_small_kernel<false>), dim3(grid), dim3(K3_T), 0, c->stream, a, words);
  }
  BCE_HIP_TRY(c, hipGetLastError());
  c->stats.k3_launches += 1.0 * count;
  return BCE_HIP_OK;
}

int k3_clear_small_bail(bce_hip_ctx *c) {
  EnumCtl *d = c->ctl.as<EnumCtl>();
  BCE_HIP_TRY(c, hipMemsetAsync(&d->small_bail, 0, sizeof(uint32_t), c->stream));
  return BCE_HIP_OK;
}

int k3_tail(bce_hip_ctx *c, uint32_t max_rounds) {
  if (max_rounds == 0 || max_rounds > K3_TAIL_MAXROUNDS) max_rounds = K3_TAIL_MAXROUNDS;
  BCE_TRY(ensure(c, c->truns, (size_t)K3_TAIL_MAXROUNDS * 8 * sizeof(RunEntry)));
  if (!c->h_truns) BCE_TRY(pin_alloc(c, &c->h_truns, (size_t)K3_TAIL_MAXROUNDS * 8 * sizeof(RunEntry)));
  const K3Args a = k3_make_args(c, c->round, 0);
  if (c->scan_mode) hipLaunchKernelGGL(k3_tail_kernel<true>, dim3(1), dim3(KT_T), 0, c->stream, a, c->truns.as<RunEntry>(), max_rounds);
  else hipLaunchKernelGGL(k3_tail_kernel<false>, dim3(1), dim3(KT_T), 0, c->stream, a, c->truns.as<RunEntry>(), max_rounds);
  BCE_HIP_TRY(c, hipGetLastError());
  c->stats.k3_launches += 1.0;
  return BCE_HIP_OK;
}

int k3_fetch_tail_runs(bce_hip_ctx *c, uint32_t rounds) {
  if (!rounds) return BCE_HIP_OK;
  BCE_HIP_TRY(c, hipMemcpyAsync(c->h_truns, c->truns.p, (size_t)rounds * 8 * sizeof(RunEntry), hipMemcpyDeviceToHost, c->stream));
  BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
  const RunEntry *r = reinterpret_cast<const RunEntry *>(c->h_truns);
  for (uint32_t i = 0; i < rounds; ++i)
    for (int p = 0; p < 8; ++p)
      if (r[(size_t)i * 8 + p].count) c->run_log[p].push_back(r[(size_t)i * 8 + p]);
  return BCE_HIP_OK;
}

int k3_sync_ctl(bce_hip_ctx *c, EnumCtl *out) {
  BCE_HIP_TRY(c, hipMemcpyAsync(c->h_ctl, c->ctl.p, sizeof(EnumCtl), hipMemcpyDeviceToHost, c->stream));
  BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
  memcpy(out, c->h_ctl, sizeof(EnumCtl));
  return BCE_HIP_OK;
}

int k3_fetch_runs(bce_hip_ctx *c, uint32_t first_round, uint32_t count) {
  (void)first_round;
  if (!count) return BCE_HIP_OK;
  BCE_HIP_TRY(c, hipMemcpyAsync(c->h_runs, c->runs.p, (size_t)count * 8 * sizeof(RunEntry), hipMemcpyDeviceToHost,
                                c->stream));
  BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
  const RunEntry *r = reinterpret_cast<const RunEntry *>(c->h_runs);
  for (uint32_t i = 0; i < count; ++i)
    for (int p = 0; p < 8; ++p)
      if (r[(size_t)i * 8 + p].count) c->run_log[p].push_back(r[(size_t)i * 8 + p]);
  return BCE_HIP_OK;
}

int k3_reset_symbols(bce_hip_ctx *c) {
  // sym_total = 0, need_flush = 0 (skip_round is informational)
  EnumCtl *d = c->ctl.as<EnumCtl>();
  BCE_HIP_TRY(c, hipMemsetAsync(&d->sym_total, 0, sizeof(uint64_t), c->stream));
  BCE_HIP_TRY(c, hipMemsetAsync(&d->need_flush, 0, sizeof(uint32_t), c->stream));
  for (int p = 0; p < 8; ++p) c->run_log[p].clear();
  return BCE_HIP_OK;
}

int k3_grow_symbols(bce_hip_ctx *c, uint64_t cap) {
  if (cap >= (1ull << 31)) return BCE_HIP_E_OVERFLOW;
  // the buffer is empty (sym_total == 0): nothing to preserve
  BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->k4_stream) BCE_HIP_TRY(c, hipStreamSynchronize(c->k4_stream));    // a flush may still be reading the other pair
  BCE_TRY(ensure(c, c->skey[0], (size_t)cap * 4));
  BCE_TRY(ensure(c, c->sesc, (size_t)cap * 4));
  if (c->overlap && !c->scan_mode) {
    BCE_TRY(ensure(c, c->skey_alt, (size_t)cap * 4));
    BCE_TRY(ensure(c, c->sesc_alt, (size_t)cap * 4));
  }
  if (c->scan_mode) BCE_TRY(ensure(c, c->scanrec, (size_t)cap * 4));
  c->sym_cap = cap;
  EnumCtl *d = c->ctl.as<EnumCtl>();
  BCE_HIP_TRY(c, hipMemcpyAsync(&d->sym_cap, &c->sym_cap, sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
  BCE_HIP_TRY(c, hipMemsetAsync(&d->need_flush, 0, sizeof(uint32_t), c->stream));
  BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
  return BCE_HIP_OK;
}

int k3_get_nodes(bce_hip_ctx *c, int plane, uint32_t *out, uint32_t cap, uint32_t *count) {
  EnumCtl ctl;
  BCE_TRY(k3_sync_ctl(c, &ctl));
  const uint32_t par = c->round & 1u;
  const uint32_t c0 = ctl.cnt[par][plane][0], c1 = ctl.cnt[par][plane][1];
  *count = c0 + c1;
  if (c0 + c1 > cap) return BCE_HIP_E_OVERFLOW;
  const uint32_t capl = c->capL[par];
  const Node *src = c->nlist[par].as<Node>() + (size_t)plane * capl;
  if (c0) BCE_HIP_TRY(c, hipMemcpy(out, src, (size_t)c0 * sizeof(Node), hipMemcpyDeviceToHost));
  if (c1) {
    std::vector<Node> tmp(c1);
    BCE_HIP_TRY(c, hipMemcpy(tmp.data(), src + (capl - c1), (size_t)c1 * sizeof(Node), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < c1; ++i) {   // stored downwards from cap-1
      const Node &nd = tmp[c1 - 1 - i];
      out[3 * (size_t)(c0 + i) + 0] = nd.s; out[3 * (size_t)(c0 + i) + 1] = nd.x0; out[3 * (size_t)(c0 + i) + 2] = nd.x1;
    }
  }
  return BCE_HIP_OK;
}

}  // namespace bce
