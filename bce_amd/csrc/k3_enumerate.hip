// k3_enumerate.hip -- K3: the interval split/count enumeration ("interval-count kernel").
// Replaces the round loop of BCE::code, mode 1 (bce.cpp:1236-1374; node body :1261-1351).
//
// One round = one pass over every plane's node list (all 8 planes in the same launches).  A node is an
// absolute (s, x0, x1) triple; lists are sorted by s, children come out sorted, so every round is a
// streaming read of triples + 2-3 rank gathers per node + two stable compactions (child0 / child1
// lists of the next plane) + one compaction of symbol records.  The gamma-coded pArray queues
// (bce.cpp:226-356) do not exist here.
//
// Node lists of a plane live in one buffer of capP nodes: the child0 list (positions < zeros, the
// reference's Q[i][0]) grows up from index 0, the child1 list (Q[i][1]) grows DOWN from capP-1, so the
// two lists share the capacity whatever their split is.
//
// Round structure (v1): count kernel -> single-block scan -> write kernel, all parameterised by the
// round parity and driven by a device-resident control block, so the host can queue many rounds
// without reading anything back.  If the symbol buffer could overflow the scan kernel marks the round
// as skipped (need_flush) and every later queued round becomes a no-op; the host flushes the model
// (K4) and resumes from that round.
#include <stdlib.h>

#include "common.h"
#include "scan_util.h"

namespace bce {

constexpr int K3_T = 256;
constexpr int K3_NPT = 4;                       // nodes per thread
constexpr uint32_t K3_TILE = K3_T * K3_NPT;     // 1024 nodes per tile
constexpr uint32_t K3_MAXBATCH = 256;           // rounds per run-table batch

struct K3Args {
  EnumCtl *ctl;
  Node *nodes;            // [2][8][capP]
  const Granule *gran;    // [8][ngran]
  const PlaneCfg *cfg;    // [8]
  uint32_t *symkey;       // symbol records: key words (K4 sort input)
  uint32_t *symesc;       // symbol records: escape words
  uint32_t *tilecnt;      // [tiles][4]
  uint32_t *tileoff;      // [tiles][4]
  RunEntry *runs;         // [K3_MAXBATCH][8]
  uint32_t capP, ngran, n;
  uint32_t zeros[8];
  uint32_t par, round, run_slot;
};

__device__ __forceinline__ Node *plane_nodes(const K3Args &a, uint32_t par, uint32_t p) {
  return a.nodes + ((size_t)(par * 8u + p)) * a.capP;
}

// tile -> plane lookup table: tp[p] = first tile of plane p, tp[8] = total tiles
__device__ __forceinline__ void tile_prefix(const K3Args &a, uint32_t tp[9]) {
  uint32_t acc = 0;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    tp[p] = acc;
    const uint32_t m = a.ctl->cnt[a.par][p][0] + a.ctl->cnt[a.par][p][1];
    acc += (m + K3_TILE - 1) / K3_TILE;
  }
  tp[8] = acc;
}

// Process one tile.  WRITE=false: count children/symbols.  WRITE=true: place them.
template <bool WRITE>
__device__ __forceinline__ void k3_tile(const K3Args &a, uint32_t p, uint32_t tile_in_plane, uint32_t tile_global,
                                        uint32_t (*lds_cnt)[4][3]) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
  const uint32_t c0n = a.ctl->cnt[a.par][p][0], c1n = a.ctl->cnt[a.par][p][1];
  const uint32_t M = c0n + c1n;
  const Node *src = plane_nodes(a, a.par, p);
  const Granule *G = a.gran + (size_t)p * a.ngran;
  const uint32_t zp = a.zeros[p];
  auto rank1 = [&](uint32_t s) -> uint32_t {
    const uint32_t g = div96(s);
    const Granule q = G[g];
    return granule_rank1(q, s - g * 96u);
  };
  StepOut so[K3_NPT];
  uint32_t pre0[K3_NPT], pre1[K3_NPT], pres[K3_NPT];
  const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int it = 0; it < K3_NPT; ++it) {
    const uint32_t q = tile_in_plane * K3_TILE + (uint32_t)it * K3_T + tid;
    const bool valid = q < M;
    so[it].has0 = so[it].has1 = so[it].hassym = 0;
    if (valid) {
      const Node nd = src[q < c0n ? q : (a.capP - 1u - (q - c0n))];
      node_step(nd, zp, rank1, so[it]);
    }
    const uint64_t b0 = __ballot(so[it].has0), b1 = __ballot(so[it].has1), bs = __ballot(so[it].hassym);
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
      uint32_t t = 0;
      for (int it = 0; it < K3_NPT; ++it)
        for (int ww = 0; ww < 4; ++ww) t += lds_cnt[it][ww][tid];
      a.tilecnt[(size_t)tile_global * 4 + tid] = t;
    }
  } else {
    const uint32_t o0 = a.tileoff[(size_t)tile_global * 4 + 0];
    const uint32_t o1 = a.tileoff[(size_t)tile_global * 4 + 1];
    const uint64_t os = (uint64_t)a.tileoff[(size_t)tile_global * 4 + 2] |
                        ((uint64_t)a.tileoff[(size_t)tile_global * 4 + 3] << 32);
    const uint32_t pn = (p + 1u) & 7u;
    Node *dst = plane_nodes(a, a.par ^ 1u, pn);
    const PlaneCfg &cfg = a.cfg[p];
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
      if (so[it].has0) dst[o0 + b0 + pre0[it]] = so[it].c0;
      if (so[it].has1) dst[a.capP - 1u - (o1 + b1 + pre1[it])] = so[it].c1;
      if (so[it].hassym) {
        uint32_t kw, ew;
        pack_symbol(cfg, p, so[it].sym, so[it].k, so[it].ctx1, so[it].ctx2, so[it].ctxs, kw, ew);
        a.symkey[os + bs + pres[it]] = kw;
        a.symesc[os + bs + pres[it]] = ew;
      }
    }
  }
  __syncthreads();
}

template <bool WRITE>
__global__ __launch_bounds__(K3_T) void k3_tiles_kernel(K3Args a) {
  __shared__ uint32_t tp[9];
  __shared__ uint32_t lds_cnt[K3_NPT][4][3];
  if (a.ctl->need_flush || a.ctl->overflow) return;
  if (threadIdx.x == 0) tile_prefix(a, tp);
  __syncthreads();
  const uint32_t T = tp[8];
  for (uint32_t tile = blockIdx.x; tile < T; tile += gridDim.x) {
    uint32_t p = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) p += (tile >= tp[k]) ? 1u : 0u;
    k3_tile<WRITE>(a, p, tile - tp[p], tile, lds_cnt);
  }
}

// Single block: exclusive scan of the tile counts per plane, symbol bases, next-round counts, run table.
__global__ __launch_bounds__(1024) void k3_scan_kernel(K3Args a) {
  __shared__ uint32_t tp[9];
  __shared__ uint32_t tot[8][3];
  EnumCtl *ctl = a.ctl;
  if (ctl->need_flush || ctl->overflow) return;
  const uint32_t tid = threadIdx.x;
  if (tid == 0) tile_prefix(a, tp);
  __syncthreads();
  for (uint32_t p = 0; p < 8; ++p) {
    uint32_t r0 = 0, r1 = 0, rs = 0;
    for (uint32_t base = tp[p]; base < tp[p + 1]; base += 1024) {
      const uint32_t t = base + tid;
      const bool valid = t < tp[p + 1];
      const uint32_t v0 = valid ? a.tilecnt[(size_t)t * 4 + 0] : 0u;
      const uint32_t v1 = valid ? a.tilecnt[(size_t)t * 4 + 1] : 0u;
      const uint32_t vs = valid ? a.tilecnt[(size_t)t * 4 + 2] : 0u;
      uint32_t t0, t1, ts;
      const uint32_t e0 = block_excl_scan_sum<1024>(v0, &t0);
      const uint32_t e1 = block_excl_scan_sum<1024>(v1, &t1);
      const uint32_t es = block_excl_scan_sum<1024>(vs, &ts);
      if (valid) {
        a.tileoff[(size_t)t * 4 + 0] = r0 + e0;
        a.tileoff[(size_t)t * 4 + 1] = r1 + e1;
        a.tileoff[(size_t)t * 4 + 2] = rs + es;   // plane-local for now; the symbol base is added below
      }
      r0 += t0; r1 += t1; rs += ts;
    }
    if (tid == 0) { tot[p][0] = r0; tot[p][1] = r1; tot[p][2] = rs; }
  }
  __syncthreads();
  // symbol bases and the flush / overflow decisions (uniform: every thread computes the same values)
  uint64_t symsum = 0, sb[8];
  uint32_t nextn = 0, curn = 0;
  bool ovf = false;
  for (int p = 0; p < 8; ++p) {
    sb[p] = ctl->sym_total + symsum;
    symsum += tot[p][2];
    nextn += tot[p][0] + tot[p][1];
    curn += ctl->cnt[a.par][p][0] + ctl->cnt[a.par][p][1];
    if ((uint64_t)tot[p][0] + tot[p][1] > a.capP) ovf = true;
  }
  const bool flush = ctl->sym_total + symsum > ctl->sym_cap;
  __syncthreads();   // everyone has read ctl before thread 0 changes it
  if (ovf) { if (tid == 0) ctl->overflow = 1; return; }
  if (flush) { if (tid == 0) { ctl->need_flush = 1; ctl->skip_round = a.round; ctl->want_syms = symsum; } return; }
  // fold the symbol bases into the tile offsets (64-bit: low in [2], high in [3])
  for (uint32_t p = 0; p < 8; ++p)
    for (uint32_t t = tp[p] + tid; t < tp[p + 1]; t += 1024) {
      const uint64_t v = sb[p] + a.tileoff[(size_t)t * 4 + 2];
      a.tileoff[(size_t)t * 4 + 2] = (uint32_t)v;
      a.tileoff[(size_t)t * 4 + 3] = (uint32_t)(v >> 32);
    }
  if (tid < 8) {
    const uint32_t p = tid, pn = (p + 1u) & 7u;
    ctl->cnt[a.par ^ 1u][pn][0] = tot[p][0];
    ctl->cnt[a.par ^ 1u][pn][1] = tot[p][1];
    RunEntry e; e.start = sb[p]; e.count = tot[p][2]; e.round = a.round;
    a.runs[(size_t)a.run_slot * 8 + p] = e;
  }
  if (tid == 0) {
    ctl->sym_total += symsum;
    ctl->nodes_total += curn;
    ctl->next_nodes = nextn;
    if (nextn == 0 && ctl->done_round == 0xFFFFFFFFu) ctl->done_round = a.round + 1u;
  }
}

static K3Args make_args(bce_hip_ctx *c, uint32_t round, uint32_t run_slot) {
  K3Args a;
  a.ctl = c->ctl.as<EnumCtl>();
  a.nodes = c->nodes.as<Node>();
  a.gran = c->gran.as<Granule>();
  a.cfg = c->dcfg.as<PlaneCfg>();
  a.symkey = c->skey[0].as<uint32_t>();
  a.symesc = c->sesc.as<uint32_t>();
  a.tilecnt = c->tilecnt.as<uint32_t>();
  a.tileoff = c->tileoff.as<uint32_t>();
  a.runs = c->runs.as<RunEntry>();
  a.capP = c->capP; a.ngran = c->ngran; a.n = c->n;
  for (int i = 0; i < 8; ++i) a.zeros[i] = c->zeros[i];
  a.par = round & 1u; a.round = round; a.run_slot = run_slot;
  return a;
}

static uint32_t default_capP(uint32_t n) {
  // worst case is n/2 nodes per plane-round (disjoint intervals of width >= 2)
  const uint64_t worst = (uint64_t)n / 2 + 2;
  const uint64_t soft = (uint64_t)192 << 20;   // 192M nodes per plane buffer = 36.9 GB for all 16 buffers
  return (uint32_t)(worst < soft ? worst : soft);
}

int k3_begin(bce_hip_ctx *c) {
  const uint32_t n = c->n;
  c->capP = default_capP(n);
  BCE_TRY(ensure(c, c->nodes, (size_t)16 * c->capP * sizeof(Node)));
  const size_t tiles = (size_t)8 * ((c->capP + K3_TILE - 1) / K3_TILE) + 8;
  BCE_TRY(ensure(c, c->tilecnt, tiles * 16));
  BCE_TRY(ensure(c, c->tileoff, tiles * 16));
  BCE_TRY(ensure(c, c->ctl, sizeof(EnumCtl)));
  BCE_TRY(ensure(c, c->runs, (size_t)K3_MAXBATCH * 8 * sizeof(RunEntry)));
  if (!c->h_ctl) BCE_HIP_TRY(c, hipHostMalloc(&c->h_ctl, sizeof(EnumCtl), hipHostMallocDefault));
  if (!c->h_runs) BCE_HIP_TRY(c, hipHostMalloc(&c->h_runs, (size_t)K3_MAXBATCH * 8 * sizeof(RunEntry), hipHostMallocDefault));
  // symbol buffer capacity
  uint64_t cap = c->sym_cap_user;
  if (!cap) {
    // default flush granularity: 16M records (pipelines with the host coders); a round that needs more
    // grows the buffer on demand (k3_grow_symbols)
    const char *env = getenv("BCE_HIP_FLUSH_RECORDS");
    const uint64_t want = (uint64_t)8 * n + 1024, soft = env ? strtoull(env, nullptr, 10) : ((uint64_t)1 << 24);
    cap = want < soft ? want : soft;
  }
  c->sym_cap = cap;
  BCE_TRY(ensure(c, c->skey[0], (size_t)cap * 4));
  BCE_TRY(ensure(c, c->sesc, (size_t)cap * 4));
  // roots: (0, C[i], n - C[i]) with C[i] = zeros(plane (i+7)%8), only where both are non-zero (bce.cpp:1237-1240)
  EnumCtl ctl;
  memset(&ctl, 0, sizeof ctl);
  ctl.sym_cap = cap;
  ctl.done_round = 0xFFFFFFFFu;
  for (int i = 0; i < 8; ++i) {
    const uint32_t C = c->zeros[(i + 7) & 7];
    if (C && n - C) {
      Node root = {0u, C, n - C};
      BCE_HIP_TRY(c, hipMemcpyAsync(c->nodes.as<Node>() + (size_t)i * c->capP, &root, sizeof root,
                                    hipMemcpyHostToDevice, c->stream));
      BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));   // `root` is a stack temporary
      ctl.cnt[0][i][0] = 1;
    }
  }
  BCE_HIP_TRY(c, hipMemcpyAsync(c->ctl.p, &ctl, sizeof ctl, hipMemcpyHostToDevice, c->stream));
  BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->round = 0;
  c->enum_active = true;
  for (int p = 0; p < 8; ++p) c->run_log[p].clear();
  c->stats.k3_ms = 0; c->stats.k3_launches = 0;
  return BCE_HIP_OK;
}

int k3_rounds(bce_hip_ctx *c, uint32_t count) {
  if (count > K3_MAXBATCH) count = K3_MAXBATCH;
  const uint32_t grid = 2048;
  for (uint32_t i = 0; i < count; ++i) {
    const K3Args a = make_args(c, c->round + i, i);
    hipLaunchKernelGGL(k3_tiles_kernel<false>, dim3(grid), dim3(K3_T), 0, c->stream, a);
    hipLaunchKernelGGL(k3_scan_kernel, dim3(1), dim3(1024), 0, c->stream, a);
    hipLaunchKernelGGL(k3_tiles_kernel<true>, dim3(grid), dim3(K3_T), 0, c->stream, a);
  }
  BCE_HIP_TRY(c, hipGetLastError());
  c->stats.k3_launches += 3.0 * count;
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
  BCE_TRY(ensure(c, c->skey[0], (size_t)cap * 4));
  BCE_TRY(ensure(c, c->sesc, (size_t)cap * 4));
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
  const Node *src = c->nodes.as<Node>() + (size_t)(par * 8u + (uint32_t)plane) * c->capP;
  if (c0) BCE_HIP_TRY(c, hipMemcpy(out, src, (size_t)c0 * sizeof(Node), hipMemcpyDeviceToHost));
  if (c1) {
    std::vector<Node> tmp(c1);
    BCE_HIP_TRY(c, hipMemcpy(tmp.data(), src + (c->capP - c1), (size_t)c1 * sizeof(Node), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < c1; ++i) {   // stored downwards from capP-1
      const Node &nd = tmp[c1 - 1 - i];
      out[3 * (size_t)(c0 + i) + 0] = nd.s; out[3 * (size_t)(c0 + i) + 1] = nd.x0; out[3 * (size_t)(c0 + i) + 2] = nd.x1;
    }
  }
  return BCE_HIP_OK;
}

}  // namespace bce
