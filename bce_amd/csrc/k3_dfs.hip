// k3_dfs.hip -- K3, depth-first tail.  Finishes BCE::code (bce.cpp:1236-1374) once few nodes are alive.
//
// Why: the number of rounds is the length in bits of the longest repeated string; a 20 KB repeat keeps one
// two-row interval alive for 160 000 rounds, each of which is a pass-through that emits nothing.  The round
// structure only matters for the ORDER of the symbols inside each coder's stream (round, then s).  So when the
// live set is small and almost all of the 8n-8 nodes are done, every live node is handed to one thread that walks
// its whole subtree depth-first with a private stack -- no barriers, no launches -- and tags each symbol with
// (plane, round, s); the tagged symbols are sorted into stream order afterwards and go through K4 as usual.
//
// A walker at plane 0 (= BWT row order) whose x rows all have the same preceding bytes can SKIP the chain: the
// next k bytes are 8k pass-through rounds, the rows stay adjacent (LF-mapping keeps the order of rows with equal
// bytes), and they end at row ISA[SA[s] - k], 8k rounds later, with (x0, x1) unchanged.  k is found by comparing
// the text backwards from the x suffix-array positions (exact, no hashing).  SA / ISA are K1's arrays; the skip
// is disabled when K1 did not end with all rotations distinct (periodic inputs) or the BWT was injected.
#include <stdlib.h>

#include "common.h"
#include "k3_args.h"

namespace bce {

constexpr int KD_T = 64;                       // one wave per block: walkers spread over the CUs
constexpr uint32_t KD_STACK = 192;             // pending siblings per walker
constexpr uint32_t KD_MAXX = 8;                // rows compared for a chain skip
constexpr uint32_t K3_DFS_ENTER = 8192;        // live nodes at which the depth-first tail may start
constexpr uint32_t K3_DFS_BUDGET = 1u << 18;   // nodes one walker may classify before the attempt is abandoned
constexpr uint32_t K3_DFS_SYMCAP = 4u << 20;   // tagged symbols

struct DfsCtl {
  uint32_t nsym;         // tagged symbols emitted
  uint32_t err;          // 1 stack overflow, 2 symbol capacity, 3 a walker exceeded its budget (bushy subtree)
  uint64_t nodes;        // nodes visited (skipped pass-through nodes included)
  uint64_t maxround;
  uint32_t cntp[8];      // symbols per plane
};

struct DNode { uint32_t s, x0, x1, plane; uint64_t round; };

struct DfsArgs {
  K3Args k;
  const uint8_t *text;
  const uint32_t *sa, *isa;
  uint32_t skip_ok;
  DfsCtl *dctl;
  uint32_t *tkey, *tesc, *ts, *trlo, *trhi;
  DNode *stacks;
  uint64_t round0;
  uint32_t symcap;
  uint32_t budget;
};

// Number of bytes on which the rotations starting at p and q agree going BACKWARDS (cyclic), capped at lim.
// Executed by the WHOLE wave with uniform arguments: while neither side wraps, each of the 64 lanes compares 16
// bytes, i.e. 1 KB per step (a multi-megabyte repeat is a few thousand steps); the rest goes byte by byte.
__device__ __forceinline__ uint32_t ld32u(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

__device__ __forceinline__ uint32_t lce_back_wave(const uint8_t *__restrict__ T, uint32_t n, uint32_t p, uint32_t q,
                                                  uint32_t lim, uint32_t lane) {
  uint32_t t = 0;
  uint32_t cp = p, cq = q;                                   // the next bytes to compare are T[cp-1], T[cq-1] (cyclic)
  while (t < lim) {
    if (cp == 0) cp = n;
    if (cq == 0) cq = n;
    uint32_t run = cp < cq ? cp : cq;                        // bytes before either side wraps
    if (run > lim - t) run = lim - t;
    if (run >= 1024) {
      // lane L owns the bytes at distances [16 L, 16 L + 16) from the current position
      const uint8_t *a = T + (cp - 16u * lane - 16u), *b = T + (cq - 16u * lane - 16u);
      uint32_t match = 16;                                   // bytes matching from the NEAREST (highest address) end
#pragma unroll
      for (int w = 3; w >= 0; --w) {
        const uint32_t d = ld32u(a + 4 * w) ^ ld32u(b + 4 * w);
        if (d && match == 16) match = (uint32_t)(3 - w) * 4u + ((uint32_t)__clz((int)d) >> 3);
      }
      const uint64_t mm = __ballot(match < 16);
      if (mm) {
        const uint32_t L = (uint32_t)__ffsll((long long)mm) - 1u;     // nearest lane with a mismatch
        return t + 16u * L + (uint32_t)__builtin_amdgcn_readlane((int)match, (int)L);
      }
      t += 1024; cp -= 1024; cq -= 1024;
    } else {                                                 // uniform scalar stretch (< 1 KB, up to the next wrap)
      uint32_t i = 0;
      while (i < run && T[cp - 1 - i] == T[cq - 1 - i]) ++i;
      t += i; cp -= i; cq -= i;
      if (i < run) return t;
    }
  }
  return t;
}

__global__ __launch_bounds__(KD_T) void k3_dfs_kernel(DfsArgs a) {
  const K3Args &k = a.k;
  const EnumCtl *ctl = k.ctl;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t gid = blockIdx.x * KD_T + threadIdx.x;
  // my start node: flattened index over the planes' lists
  uint32_t acc = 0, p0 = 8, idx = 0;
#pragma unroll
  for (uint32_t p = 0; p < 8; ++p) {
    const uint32_t m = ctl->cnt[k.par][p][0] + ctl->cnt[k.par][p][1];
    if (p0 == 8 && gid < acc + m) { p0 = p; idx = gid - acc; }
    acc += m;
  }
  bool alive = p0 != 8;
  DNode cur{0u, 1u, 1u, 1u, 0ull};
  if (alive) {
    const uint32_t c0 = ctl->cnt[k.par][p0][0];
    const Node nd = plane_nodes(k, k.par, p0)[idx < c0 ? idx : (k.capP - 1u - (idx - c0))];
    cur.s = nd.s; cur.x0 = nd.x0; cur.x1 = nd.x1; cur.plane = p0; cur.round = a.round0;
  }
  DNode *stack = a.stacks + (size_t)gid * KD_STACK;
  uint32_t sp = 0;
  uint64_t nodes = 0, maxround = 0;
  uint32_t visited = 0;
  const uint32_t n = k.n;
  // All 64 lanes stay in the loop until every walker of the wave is finished: finished lanes help with the
  // cooperative text comparisons.
  while (__any(alive)) {
    if (alive && (++visited > a.budget || a.dctl->err)) { if (!a.dctl->err) a.dctl->err = 3; alive = false; }
    // ---- chain skip: lanes at plane 0 with few rows, served one after the other by the whole wave ----
    const uint32_t x = cur.x0 + cur.x1;
    uint64_t want = __ballot(alive && a.skip_ok && cur.plane == 0 && x <= KD_MAXX);
    uint32_t mykk = 0;
    while (want) {
      const int L = __ffsll((long long)want) - 1;
      want &= want - 1;
      const uint32_t sL = (uint32_t)__builtin_amdgcn_readlane((int)cur.s, L);
      const uint32_t xL = (uint32_t)__builtin_amdgcn_readlane((int)x, L);
      const uint32_t pa = a.sa[sL];
      uint32_t kk = n - 1;
      const uint32_t ra = a.isa[pa];
      for (uint32_t i = 1; i < xL && kk; ++i) {
        const uint32_t pb = a.sa[sL + i];
        if (a.isa[pb] == ra) continue;                       // identical rotations (periodic input): agree for ever
        kk = lce_back_wave(a.text, n, pa, pb, kk, lane);
      }
      if (kk == n - 1) kk = 0;                               // every row identical: cannot happen for a live node
      if ((int)lane == L) mykk = kk;
    }
    if (alive) {
      if (mykk) {                                    // mykk whole bytes of pass-through: 8*mykk rounds, no symbols
        const uint32_t pa = a.sa[cur.s];
        cur.s = a.isa[pa >= mykk ? pa - mykk : pa + n - mykk];
        cur.round += 8ull * mykk;
        nodes += 8ull * mykk;
      }
      const uint32_t p = cur.plane;
      const Granule *G = k.gran + (size_t)p * k.ngran;
      const Node nd{cur.s, cur.x0, cur.x1};
      const uint32_t ga = div96(nd.s), gb = div96(nd.s + nd.x0 + nd.x1), gm = div96(nd.s + nd.x0);
      const Granule qa = G[ga], qb = G[gb];
      NodeFlat nf;
      node_flat_pre(nd, granule_rank1(qa, nd.s - ga * 96u), granule_rank1(qb, nd.s + nd.x0 + nd.x1 - gb * 96u), nf);
      Granule qm = gm == ga ? qa : qb;
      if (nf.need_mid && gm != ga && gm != gb) qm = G[gm];
      uint32_t has0, has1, sym, kq;
      Node c0, c1;
      node_flat_post(nd, k.zeros[p], nf, granule_rank1(qm, nd.s + nd.x0 - gm * 96u), has0, c0, has1, c1, sym, kq);
      ++nodes;
      maxround = cur.round > maxround ? cur.round : maxround;
      if (nf.need_mid) {
        const uint32_t i = atomicAdd(&a.dctl->nsym, 1u);
        if (i >= a.symcap) { a.dctl->err = 2; alive = false; }
        else {
          uint32_t kw, ew;
          pack_symbol(k.cfg[p], p, sym, kq, nf.n0x, nd.x1, nd.x0 + nd.x1, kw, ew);
          a.tkey[i] = kw; a.tesc[i] = ew; a.ts[i] = nd.s;
          a.trlo[i] = (uint32_t)cur.round;
          a.trhi[i] = (uint32_t)(cur.round >> 32) | (p << 8);      // round < 2^40
          atomicAdd(&a.dctl->cntp[p], 1u);
        }
      }
      const uint32_t pn = (p + 1u) & 7u;
      if (has0 && has1) {
        if (sp >= KD_STACK) { a.dctl->err = 1; alive = false; }
        else stack[sp++] = DNode{c1.s, c1.x0, c1.x1, pn, cur.round + 1};
      }
      if (has0) { cur = DNode{c0.s, c0.x0, c0.x1, pn, cur.round + 1}; }
      else if (has1) { cur = DNode{c1.s, c1.x0, c1.x1, pn, cur.round + 1}; }
      else if (sp) { cur = stack[--sp]; }
      else alive = false;
    }
  }
  atomicAdd((unsigned long long *)&a.dctl->nodes, (unsigned long long)nodes);
  atomicMax((unsigned long long *)&a.dctl->maxround, (unsigned long long)maxround);
}

// keys[i] = src[perm[i]]
__global__ void kd_gather_kernel(const uint32_t *__restrict__ src, const uint32_t *__restrict__ perm, uint32_t m,
                                 uint32_t *__restrict__ keys) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) keys[i] = src[perm[i]];
}
__global__ void kd_iota_kernel(const uint32_t *__restrict__ src, uint32_t m, uint32_t *__restrict__ keys,
                               uint32_t *__restrict__ vals) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) { keys[i] = src[i]; vals[i] = i; }
}
__global__ void kd_place_kernel(const uint32_t *__restrict__ tkey, const uint32_t *__restrict__ tesc,
                                const uint32_t *__restrict__ perm, uint32_t m, uint32_t *__restrict__ symkey,
                                uint32_t *__restrict__ symesc) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
    symkey[i] = tkey[perm[i]];
    symesc[i] = tesc[perm[i]];
  }
}

// Host side.  The walkers start when the node count has stopped growing, at most K3_DFS_ENTER nodes are alive and
// an eighth of all nodes has been visited (i.e. not in the ramp-up); a walker that meets a bushy subtree instead
// of chains gives up after K3_DFS_BUDGET nodes and the attempt is abandoned.
// Preconditions (checked by the caller): the symbol buffer is empty (everything emitted so far has been flushed),
// `ctl` is current.  On success the symbol buffer holds the tail's symbols in stream order,
// run_log holds one run per plane, and *done = true.  On a walker error nothing has been changed and the caller
// continues with the round-based kernels.
int k3_dfs_tail(bce_hip_ctx *c, const EnumCtl &ctl, bool *done) {
  *done = false;
  if (c->scan_mode || c->dbg_no_dfs) return BCE_HIP_OK;
  const uint32_t n = c->n;
  const uint32_t live = ctl.next_nodes;
  const uint64_t all = 8ull * (n - 1);
  if (live == 0 || live > K3_DFS_ENTER || ctl.nodes_total < all / 8) return BCE_HIP_OK;
  const uint32_t cap = K3_DFS_SYMCAP;
  // carve: tkey tesc ts trlo trhi | sort keys x2 vals x2 | DfsCtl | stacks
  const size_t o_sort = (size_t)cap * 4 * 5, o_ctl = o_sort + (size_t)cap * 4 * 4, o_stack = o_ctl + 256;
  BCE_TRY(ensure(c, c->dfs, o_stack + (size_t)K3_DFS_ENTER * KD_STACK * sizeof(DNode)));
  uint8_t *base = c->dfs.as<uint8_t>();
  DfsArgs a;
  a.k = k3_make_args(c, c->round, 0);
  a.text = c->text.as<uint8_t>();
  a.sa = c->sa[c->sa_res].as<uint32_t>();
  a.isa = c->rank.as<uint32_t>();
  a.skip_ok = (!c->dbg_no_skip && c->k1_valid && c->text.p && c->rank.p) ? 1u : 0u;
  uint32_t *w = reinterpret_cast<uint32_t *>(base);
  a.tkey = w; a.tesc = w + cap; a.ts = w + 2 * (size_t)cap; a.trlo = w + 3 * (size_t)cap; a.trhi = w + 4 * (size_t)cap;
  uint32_t *sk[2] = {reinterpret_cast<uint32_t *>(base + o_sort), reinterpret_cast<uint32_t *>(base + o_sort) + cap};
  uint32_t *sv[2] = {reinterpret_cast<uint32_t *>(base + o_sort) + 2 * (size_t)cap, reinterpret_cast<uint32_t *>(base + o_sort) + 3 * (size_t)cap};
  a.dctl = reinterpret_cast<DfsCtl *>(base + o_ctl);
  a.stacks = reinterpret_cast<DNode *>(base + o_stack);
  a.round0 = c->round;
  a.symcap = cap;
  a.budget = c->dbg_dfs_budget ? c->dbg_dfs_budget : K3_DFS_BUDGET;
  BCE_HIP_TRY(c, hipMemsetAsync(a.dctl, 0, sizeof(DfsCtl), c->stream));
  hipLaunchKernelGGL(k3_dfs_kernel, dim3((live + KD_T - 1) / KD_T), dim3(KD_T), 0, c->stream, a);
  DfsCtl h;
  BCE_HIP_TRY(c, hipMemcpyAsync(&h, a.dctl, sizeof h, hipMemcpyDeviceToHost, c->stream));
  BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
  BCE_HIP_TRY(c, hipGetLastError());
  c->stats.k3_launches += 1.0;
  if (h.err) return BCE_HIP_OK;                     // fall back to the rounds; nothing was modified
  const uint32_t m = h.nsym;
  if (m > c->sym_cap) BCE_TRY(k3_grow_symbols(c, (uint64_t)m + 1024));
  if (m) {
    // stream order inside a coder = (round, s); planes are separated: LSD sort by s, round low, (plane, round high)
    const uint32_t g = (m + 255) / 256 < 2048 ? (m + 255) / 256 : 2048;
    int r = 0;
    hipLaunchKernelGGL(kd_iota_kernel, dim3(g), dim3(256), 0, c->stream, a.ts, m, sk[0], sv[0]);
    BCE_TRY(radix_sort_pairs(c, sk, sv, m, 0, 31, &r));
    uint32_t *k2[2] = {sk[r ^ 1], sk[r]}, *v2[2] = {sv[r], sv[r ^ 1]};
    hipLaunchKernelGGL(kd_gather_kernel, dim3(g), dim3(256), 0, c->stream, a.trlo, v2[0], m, k2[0]);
    BCE_TRY(radix_sort_pairs(c, k2, v2, m, 0, 32, &r));
    uint32_t *k3[2] = {k2[r ^ 1], k2[r]}, *v3[2] = {v2[r], v2[r ^ 1]};
    hipLaunchKernelGGL(kd_gather_kernel, dim3(g), dim3(256), 0, c->stream, a.trhi, v3[0], m, k3[0]);
    BCE_TRY(radix_sort_pairs(c, k3, v3, m, 0, 11, &r));
    hipLaunchKernelGGL(kd_place_kernel, dim3(g), dim3(256), 0, c->stream, a.tkey, a.tesc, v3[r], m,
                       c->skey[0].as<uint32_t>(), c->sesc.as<uint32_t>());
  }
  // control block: the enumeration is finished
  EnumCtl *d = c->ctl.as<EnumCtl>();
  const uint64_t symtot = m, nodes = ctl.nodes_total + h.nodes;
  const uint32_t done_round = (uint32_t)(h.maxround + 1 > 0xFFFFFFFEull ? 0xFFFFFFFEull : h.maxround + 1);
  BCE_HIP_TRY(c, hipMemcpyAsync(&d->sym_total, &symtot, 8, hipMemcpyHostToDevice, c->stream));
  BCE_HIP_TRY(c, hipMemcpyAsync(&d->nodes_total, &nodes, 8, hipMemcpyHostToDevice, c->stream));
  BCE_HIP_TRY(c, hipMemcpyAsync(&d->done_round, &done_round, 4, hipMemcpyHostToDevice, c->stream));
  BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (int p = 0; p < 8; ++p) c->run_log[p].clear();
  uint64_t start = 0;
  for (int p = 0; p < 8; ++p) {
    if (h.cntp[p]) c->run_log[p].push_back(RunEntry{start, h.cntp[p], c->round});
    start += h.cntp[p];
  }
  *done = true;
  return BCE_HIP_OK;
}

}  // namespace bce
