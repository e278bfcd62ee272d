// core_emul.cpp -- TEST-ONLY harness: drives the per-element functions of bce_amd/csrc/bce_core.h and
// the host coder (host_coder.cpp) sequentially on the CPU, so their arithmetic can be checked against
// the oracle without a GPU.  It is NOT a product path and is never loaded by bce_amd/.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../bce_amd/csrc/bce_core.h"
#include "../bce_amd/csrc/host_coder.h"

using namespace bce;

namespace {
// wavelet-matrix level order + granules, the layout k2_planes.hip produces
void build(const uint8_t *bwt, uint32_t n, std::vector<Granule> G[8], uint32_t zeros[8]) {
  std::vector<uint8_t> cur(bwt, bwt + n), nxt(n);
  const uint32_t ng = n / 96 + 1;
  for (int j = 0; j < 8; ++j) {
    G[j].assign(ng, Granule{0, 0, 0, 0});
    uint32_t ones = 0, z = 0;
    for (uint32_t p = 0; p <= n; ++p) {
      if (p % 96 == 0) G[j][p / 96].cum = ones;
      if (p == n) break;
      const uint32_t bit = (cur[p] >> j) & 1u;
      if (bit) {
        uint32_t o = p % 96;
        uint32_t *w = o < 32 ? &G[j][p / 96].w0 : (o < 64 ? &G[j][p / 96].w1 : &G[j][p / 96].w2);
        *w |= 1u << (o & 31);
        ++ones;
      } else ++z;
    }
    zeros[j] = z;
    uint32_t zi = 0, oi = z;
    for (uint32_t p = 0; p < n; ++p) { if ((cur[p] >> j) & 1u) nxt[oi++] = cur[p]; else nxt[zi++] = cur[p]; }
    cur.swap(nxt);
  }
}
}  // namespace

extern "C" int emul_rank1(const uint8_t *bwt, uint32_t n, int plane, const uint32_t *idx, uint32_t cnt, uint32_t *out) {
  std::vector<Granule> G[8];
  uint32_t zeros[8];
  build(bwt, n, G, zeros);
  for (uint32_t i = 0; i < cnt; ++i) { const uint32_t g = div96(idx[i]); out[i] = granule_rank1(G[plane][g], idx[i] - g * 96); }
  return 0;
}

// BWT -> archive through bce_core.h + host_coder.cpp; returns malloc'd bytes
extern "C" int emul_encode_from_bwt(const uint8_t *bwt, uint32_t n, uint32_t offset, const uint8_t *config,
                                    uint8_t **out, size_t *out_len, uint64_t *nodes_out, uint64_t *syms_out) {
  uint8_t cfgb[9][32];
  memcpy(cfgb, config ? config : &kDefaultConfig[0][0], 288);
  std::vector<Granule> G[8];
  uint32_t zeros[8];
  build(bwt, n, G, zeros);
  PlaneCfg cfg[8];
  std::vector<uint8_t> stat[8];
  for (int p = 0; p < 8; ++p) { plane_cfg_init(cfg[p], cfgb[p]); stat[p].assign(cfg[p].stat_bytes + 1, 0); }
  uint32_t C[8];
  for (int i = 0; i < 8; ++i) C[i] = zeros[(i + 7) & 7];
  HostCoder hc;
  hc.begin(cfgb, C, n);
  std::vector<Node> cur[8][2], nxt[8][2];
  for (int i = 0; i < 8; ++i)
    if (C[i] && n - C[i]) cur[i][0].push_back(Node{0, C[i], n - C[i]});
  uint64_t nodes = 0, syms = 0;
  for (;;) {
    bool any = false;
    for (int p = 0; p < 8; ++p) {
      std::vector<uint64_t> outrec;
      auto rank1 = [&](uint32_t s) { const uint32_t g = div96(s); return granule_rank1(G[p][g], s - g * 96); };
      for (int j = 0; j < 2; ++j)
        for (const Node &nd : cur[p][j]) {
          StepOut so;
          node_step(nd, zeros[p], rank1, so);
          {   // the branch-free form used by the kernels must agree with the reference-shaped one
            NodeFlat nf; uint32_t h0, h1, sy, kk; Node a0, a1;
            node_flat_pre(nd, rank1(nd.s), rank1(nd.s + nd.x0 + nd.x1), nf);
            node_flat_post(nd, zeros[p], nf, nf.need_mid ? rank1(nd.s + nd.x0) : 0u, h0, a0, h1, a1, sy, kk);
            bool ok = h0 == so.has0 && h1 == so.has1 && nf.need_mid == so.hassym;
            if (h0) ok = ok && a0.s == so.c0.s && a0.x0 == so.c0.x0 && a0.x1 == so.c0.x1;
            if (h1) ok = ok && a1.s == so.c1.s && a1.x0 == so.c1.x0 && a1.x1 == so.c1.x1;
            if (nf.need_mid) ok = ok && sy == so.sym && kk == so.k;
            if (!ok) return -7;
          }
          ++nodes;
          if (so.has0) nxt[(p + 1) & 7][0].push_back(so.c0);
          if (so.has1) nxt[(p + 1) & 7][1].push_back(so.c1);
          if (so.hassym) {
            uint32_t kw, ew;
            pack_symbol(cfg[p], (uint32_t)p, so.sym, so.k, so.ctx1, so.ctx2, so.ctxs, kw, ew);
            const uint32_t k = key_k(kw);
            uint8_t *ctr = stat[p].data() + cfg[p].off[k] + (key_slot(kw) - cfg[p].ctxoff[k]) * k;
            outrec.push_back(model_step(ctr, k, key_sym(kw), ew));
            ++syms;
          }
        }
      SymRun run{0, (uint32_t)outrec.size(), 0};
      hc.consume(p, &run, 1, outrec.data());
    }
    for (int p = 0; p < 8; ++p)
      for (int j = 0; j < 2; ++j) { cur[p][j].swap(nxt[p][j]); nxt[p][j].clear(); if (!cur[p][j].empty()) any = true; }
    if (!any) break;
  }
  std::vector<uint16_t> arch;
  hc.finish(cfgb, n, offset, arch);
  *out_len = arch.size() * 2;
  *out = (uint8_t *)malloc(*out_len ? *out_len : 1);
  memcpy(*out, arch.data(), *out_len);
  if (nodes_out) *nodes_out = nodes;
  if (syms_out) *syms_out = syms;
  return 0;
}
extern "C" void emul_free(void *p) { free(p); }

// bce_core.h's context_index / pack_symbol on arrays of (k, c1, c2, cs) with the plane's config row: out[2*i] = context,
// out[2*i+1] = slot (pack_symbol's key word).  Same float-reciprocal arithmetic as the kernels (host == device, bce_core.h).
extern "C" void emul_context(const uint8_t *row32, const uint32_t *kccs, uint32_t cnt, uint32_t *out) {
  PlaneCfg cfg;
  plane_cfg_init(cfg, row32);
  for (uint32_t i = 0; i < cnt; ++i) {
    const uint32_t k = kccs[4 * i], c1 = kccs[4 * i + 1], c2 = kccs[4 * i + 2], cs = kccs[4 * i + 3];
    out[2 * i] = context_index(cfg.bits[k], c1, c2, cs);
    uint32_t kw, ew;
    pack_symbol(cfg, 0, 0, k, c1, c2, cs, kw, ew);
    out[2 * i + 1] = key_slot(kw);
  }
}
extern "C" uint32_t emul_small_quotient(uint32_t a, uint32_t b) { return small_quotient(a, b); }

// exhaustive-ish check of the reciprocal division used by the host coder: returns the number of mismatches
extern "C" uint64_t emul_check_recip(uint64_t seed, uint64_t samples_per_divisor) {
  uint64_t bad = 0, st = seed ? seed : 1;
  auto next = [&]() { st ^= st >> 12; st ^= st << 25; st ^= st >> 27; return st * 0x2545F4914F6CDD1DULL; };
  for (uint32_t d = 2; d < 8192; ++d) {
    const uint64_t edge[] = {0, 1, d - 1ull, d, d + 1ull, ~0ull, ~0ull - 1, ~0ull - d, (~0ull / d) * d, (~0ull / d) * d - 1,
                             1ull << 63, (1ull << 63) - 1, (1ull << 48), (1ull << 48) - 1};
    for (uint64_t x : edge) bad += bce::bce_test_div_recip(x, d) != x / d;
    for (uint64_t i = 0; i < samples_per_divisor; ++i) {
      uint64_t x = next();
      if (i & 1) x >>= (next() & 63);
      bad += bce::bce_test_div_recip(x, d) != x / d;
    }
  }
  return bad;
}

// `bce -s` host part (scan_coder.cpp): symbol tuples (plane, s, k, c1, c2, cs) in the oracle's call order -> the 288-byte
// table + the nine "Result size" values.  threads = 0: the sequential ScanCoder::set / flush; else ScanSet on that many
// threads, fed in `chunks` consume() calls (as the flushes of bce_hip_scan feed it).
#include "../bce_amd/csrc/scan_coder.h"
#include <stdio.h>
#include <chrono>
static double emul_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
extern "C" int emul_scan(const uint32_t *syms6, size_t nsym, unsigned threads, unsigned chunks, uint8_t *config288, double *result9) {
  uint8_t init[9][32];
  memset(init, 0, sizeof init);
  if (threads == 0) {
    std::vector<ScanCoder> coders;
    for (int i = 0; i < 8; ++i) coders.emplace_back(i);
    for (size_t t = 0; t < nsym; ++t) { const uint32_t *r = syms6 + 6 * t; coders[r[0]].set(r[1], r[2], r[3], r[4], r[5]); }
    const double t1 = emul_now();
    for (int i = 0; i < 8; ++i) result9[i] = coders[i].flush(init);
    if (getenv("EMUL_TIMING")) fprintf(stderr, "emul_scan: sequential flush %.3f s\n", emul_now() - t1);
    ScanCoder mainc(-1);
    result9[8] = mainc.flush(init);
  } else {
    ScanSet set(threads);
    double t_cons = 0;
    if (chunks == 0) chunks = 1;
    for (unsigned ch = 0; ch < chunks; ++ch) {
      const size_t lo = nsym * ch / chunks, hi = nsym * (ch + 1) / chunks;
      std::vector<uint32_t> rec(hi - lo);
      std::vector<ScanSpan> spans[8];
      // lay the chunk out plane by plane (any layout will do: the spans say where a plane's records are, in stream order)
      size_t w = 0;
      for (int p = 0; p < 8; ++p) {
        const size_t w0 = w;
        for (size_t t = lo; t < hi; ++t) { const uint32_t *r = syms6 + 6 * t; if ((int)r[0] == p) { rec[w] = scan_pack(r[1], r[2], r[3], r[4], r[5]); ++w; } }
        if (w > w0) { const size_t mid = w0 + (w - w0) / 2; spans[p].push_back(ScanSpan{w0, mid - w0}); spans[p].push_back(ScanSpan{mid, w - mid}); }
      }
      const double t0 = emul_now();
      set.consume(rec.data(), spans);
      t_cons += emul_now() - t0;
    }
    const double t1 = emul_now();
    set.flush(init, result9);
    if (getenv("EMUL_TIMING")) fprintf(stderr, "emul_scan: consume %.3f s, flush %.3f s\n", t_cons, emul_now() - t1);
  }
  memcpy(config288, init, 288);
  return 0;
}

// scan_add_repeated against the loop it replaces
extern "C" int emul_add_repeated(double z, double c, uint64_t m, double *fast, double *slow) {
  *fast = bce::scan_add_repeated(z, c, m);
  double t = z;
  for (uint64_t i = 0; i < m; ++i) t += c;
  *slow = t;
  return *fast == *slow;
}
