// decoder_core.h -- the sequential half of the decoder, shared by the host decoder (decoder.cpp) and the
// GPU-assisted one (kd_decode.hip): AdaptiveCoder<31>'s decode side -- range decoder (bce.cpp:592-608, shift_in
// :663-669, ctor :495-504), adaptive model get (bce.cpp:555-590), VCoder::getv (bce.cpp:372-377), init(0, i)
// (bce.cpp:692-705).  Plain host C++.
#pragma once
#include <stdint.h>

#include <vector>

#include "bce_core.h"
#include "host_coder.h"

namespace bce {

constexpr uint32_t kUnknown = 0xFFFFFFFFu;

// AdaptiveCoder<31>, decode side.  One per plane, each driven by its own thread in the GPU-assisted decoder: own cache
// lines (l, h, m and the read position change with every symbol).
struct alignas(128) Decoder {
  uint64_t l = 0, h = ~0ull, m = 0;
  const uint16_t *data = nullptr;
  size_t size = 0, o = 0;
  PlaneCfg cfg;
  uint8_t bits[32];
  std::vector<uint8_t> stat;
  bool overrun = false;

  uint16_t next() { const uint16_t v = o < size ? data[o] : 0; ++o; return v; }   // reads past the end give 0 (:568)
  void open(const uint16_t *d, size_t n) {                       // ctor :495-504: the first 4 words, missing ones as 0
    data = d; size = n; o = 0; l = 0; h = ~0ull; m = 0;
    for (int i = 0; i < 4; ++i) m = (m << 16) + next();
  }
  void shift_in() {                                              // :663-669
    while (!((h ^ l) >> 48)) {
      m = (m << 16) + next();
      l = (l << 16) + 0x0000;
      h = (h << 16) + 0xFFFF;
    }
  }
  uint32_t get(uint32_t k) {                                     // :592-608
    if (h - l < k) { for (int i = 0; i < 4; ++i) m = (m << 16) + next(); l = 0; h = ~0ull; }
    const uint64_t step = (h - l) / k;
    const uint32_t s = (uint32_t)((m - l) / step);
    l += step * s;
    h = step + l - 1;
    shift_in();
    return s;
  }
  uint32_t getv() {                                              // :372-377
    uint32_t s = 0;
    int i = 0;
    for (uint32_t j = get(3); i < 31 && j != 2; ++i, j = get(3)) s |= j << i;
    return s;
  }
  void init() {                                                  // init(0, i) :692-705
    uint32_t last = 0;
    for (int b = 0; b < 32; ++b) { const uint32_t bit = get(2) ? get(6) : last; bits[b] = (uint8_t)bit; last = bit; }
    plane_cfg_init(cfg, bits);
    stat.assign(cfg.stat_bytes + 1, 0);
  }
  uint32_t get_adaptive(uint32_t k, uint32_t c1, uint32_t c2, uint32_t cs) {   // :555-590
    if (k > (uint32_t)kMaxK) {
      const uint32_t s = get(2);
      return (get_adaptive((k + (~s & 1u)) >> 1, c1, c2, cs) << 1) | s;
    }
    const uint32_t b = cfg.bits[k];
    const uint32_t ctxv = (((uint32_t)(c1 << b) / cs) << b) | ((uint32_t)(c2 << b) / cs);   // :671-677
    return get_slot(k, ctxv);
  }
  // the same with the context already resolved (k <= 31): the GPU-assisted decoder computes ctxv on the device
  uint32_t get_slot(uint32_t k, uint32_t ctxv) {
    uint8_t *ctx = stat.data() + cfg.off[k] + ctxv * k;
    if (k == 2) {
      // binary contexts (most symbols): the same arithmetic without the data-dependent loop branch
      const uint32_t c0 = ctx[0], c1 = ctx[1], tot = c0 + c1 + 2u;
      if (h - l < tot) { for (int i = 0; i < 4; ++i) m = (m << 16) + next(); l = 0; h = ~0ull; }
      const uint64_t step = div_small(h - l, tot, recip);
      const uint64_t h0 = l + step * ((uint64_t)c0 + 1) - 1;       // last value of symbol 0
      const uint32_t s = h0 < m ? 1u : 0u;
      l = s ? h0 + 1 : l;
      h = s ? h0 + step * ((uint64_t)c1 + 1) : h0;
      if (++ctx[s] == 0xFF) { ctx[0] >>= 1; ctx[1] >>= 1; }
      shift_in();
      return s;
    }
    uint32_t tot = k;
    for (uint32_t i = 0; i < k; ++i) tot += ctx[i];
    if (h - l < tot) { for (int i = 0; i < 4; ++i) m = (m << 16) + next(); l = 0; h = ~0ull; }
    const uint64_t step = div_small(h - l, tot, recip);           // tot < 8192: exact single-multiply division
    h = l - 1;
    uint32_t s = ~0u;
    do {
      ++s;
      l = h + 1;
      h += step * ((uint64_t)ctx[s] + 1);
    } while (h < m && s + 1 < k);
    if (++ctx[s] == 0xFF) for (uint32_t i = 0; i < k; ++i) ctx[i] >>= 1;
    shift_in();
    return s;
  }
  void prefetch_slot(uint32_t k, uint32_t ctxv) const { __builtin_prefetch(stat.data() + cfg.off[k] + ctxv * k); }
  const Recip *recip = recip_table();
};

// The framing of BCE::decode (bce.cpp:1177-1211): header coder (n, offset, stream sizes), the eight plane decoders
// opened and initialised, and C[i] = zeros of plane (i+7)%8.  Returns 0, or a negative bce_hip_status value.
struct ArchiveHead {
  uint32_t n = 0, offset = 0;
  uint32_t C[8] = {0};
  std::vector<Decoder> dec;
};
inline int parse_archive(const uint8_t *archive, size_t len, ArchiveHead &hd, bool header_only) {
  if (!archive || len < 4 || (len & 1)) return -1;
  const uint16_t *w = reinterpret_cast<const uint16_t *>(archive);
  const size_t nw = len / 2;
  const uint32_t header_size = w[0];                              // :1178
  if ((size_t)header_size + 1 > nw) return -1;
  Decoder mainc;
  mainc.open(w + 1, header_size);
  mainc.init();
  hd.n = mainc.getv();                                            // :1181-1183
  if (hd.n == 0) return -1;
  hd.offset = mainc.get(hd.n + 1);
  uint32_t size = mainc.getv();
  if (header_only) return 0;
  size_t coff[9];
  coff[0] = (size_t)header_size + 1;
  for (int i = 0; i < 7; ++i) {                                   // :1187-1190
    const uint32_t li = mainc.get(size + 1u);
    coff[i + 1] = coff[i] + li;
    size -= li;
  }
  coff[8] = nw;
  for (int i = 0; i < 9; ++i) if (coff[i] > nw) return -1;
  hd.dec.assign(8, Decoder());
  for (int i = 0; i < 8; ++i) {                                   // :1193-1202
    if (coff[i + 1] < coff[i]) return -1;
    hd.dec[i].open(w + coff[i], coff[i + 1] - coff[i]);
    hd.dec[i].init();
  }
  for (int i = 0; i < 8; ++i) {                                   // :1207-1211
    hd.C[i] = hd.dec[i].get(hd.n + 1);
    if (hd.C[i] > hd.n) return -1;
  }
  return 0;
}

}  // namespace bce
