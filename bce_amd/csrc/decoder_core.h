// decoder_core.h -- the sequential half of the decoder, shared by the host decoder (decoder.cpp) and the
// GPU-assisted one (kd_decode.hip): AdaptiveCoder<31>'s decode side -- range decoder (bce.cpp:592-608, shift_in
// :663-669, ctor :495-504), adaptive model get (bce.cpp:555-590), VCoder::getv (bce.cpp:372-377), init(0, i)
// (bce.cpp:692-705).  Plain host C++.
#pragma once
#include <stdint.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

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
    stat.assign(cfg.stat_bytes + 1 + 32, 0);                     // + slack: get_slot_avx2 loads 32 bytes at a slot
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
#if defined(__x86_64__)
    if (have_avx2) return get_slot_avx2(ctx, k);
#endif
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
#if defined(__x86_64__)
  // The same step for 3 <= k <= 31 without the two data-dependent loops: cum[i] = sum_{j<=i} (ctx[j] + 1) for all
  // 32 lanes at once (16-bit prefix sums), then the symbol is the number of i < k with cum[i] < t, where
  // t = floor((m - l) / step) + 1 (h_i = l - 1 + step cum[i] >= m  <=>  cum[i] >= t), capped at k - 1 as the loop is.
  __attribute__((target("avx2"))) uint32_t get_slot_avx2(uint8_t *ctx, uint32_t k) {
    alignas(32) static const int16_t iota[32] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15,
                                                 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31};
    const __m256i kk = _mm256_set1_epi16((short)k);
    const __m256i m_lo = _mm256_cmpgt_epi16(kk, _mm256_load_si256((const __m256i *)iota));         // lanes i < k
    const __m256i m_hi = _mm256_cmpgt_epi16(kk, _mm256_load_si256((const __m256i *)(iota + 16)));
    const __m256i raw = _mm256_loadu_si256((const __m256i *)ctx);                                   // stat has 32 bytes of slack
    const __m256i one = _mm256_set1_epi16(1);
    __m256i lo = _mm256_and_si256(_mm256_add_epi16(_mm256_cvtepu8_epi16(_mm256_castsi256_si128(raw)), one), m_lo);
    __m256i hi = _mm256_and_si256(_mm256_add_epi16(_mm256_cvtepu8_epi16(_mm256_extracti128_si256(raw, 1)), one), m_hi);
    // inclusive prefix sums of 16 x u16 per register: inside the 128-bit halves, then across them
    lo = _mm256_add_epi16(lo, _mm256_slli_si256(lo, 2)); hi = _mm256_add_epi16(hi, _mm256_slli_si256(hi, 2));
    lo = _mm256_add_epi16(lo, _mm256_slli_si256(lo, 4)); hi = _mm256_add_epi16(hi, _mm256_slli_si256(hi, 4));
    lo = _mm256_add_epi16(lo, _mm256_slli_si256(lo, 8)); hi = _mm256_add_epi16(hi, _mm256_slli_si256(hi, 8));
    const __m256i bsel = _mm256_set1_epi16(0x0F0E);                                                 // bytes 14,15 = element 7 of each half
    __m256i carry = _mm256_shuffle_epi8(lo, bsel);                                                  // per half: its own total
    lo = _mm256_add_epi16(lo, _mm256_permute2x128_si256(carry, carry, 0x08));                       // upper half += total of lower half
    carry = _mm256_shuffle_epi8(hi, bsel);
    hi = _mm256_add_epi16(hi, _mm256_permute2x128_si256(carry, carry, 0x08));
    const __m256i lo_tot = _mm256_permute2x128_si256(_mm256_shuffle_epi8(lo, bsel), _mm256_setzero_si256(), 0x11);   // element 15 of lo, everywhere
    hi = _mm256_add_epi16(hi, lo_tot);
    alignas(32) uint16_t cum[32];
    _mm256_store_si256((__m256i *)cum, lo);
    _mm256_store_si256((__m256i *)(cum + 16), hi);
    const uint32_t tot = cum[31];                                  // lanes >= k added nothing
    if (h - l < tot) { for (int i = 0; i < 4; ++i) m = (m << 16) + next(); l = 0; h = ~0ull; }
    const uint64_t step = div_small(h - l, tot, recip);
    const uint64_t tq = (m - l) / step + 1;
    const __m256i t16 = _mm256_set1_epi16((short)(tq > 16384 ? 16384 : tq));                        // cum <= 8192
    const uint32_t lt_lo = (uint32_t)_mm256_movemask_epi8(_mm256_and_si256(_mm256_cmpgt_epi16(t16, lo), m_lo));
    const uint32_t lt_hi = (uint32_t)_mm256_movemask_epi8(_mm256_and_si256(_mm256_cmpgt_epi16(t16, hi), m_hi));
    uint32_t s = (uint32_t)(__builtin_popcount(lt_lo) + __builtin_popcount(lt_hi)) >> 1;             // two mask bits per lane
    if (s > k - 1) s = k - 1;
    const uint64_t l0 = l;
    l = l0 + step * (s ? cum[s - 1] : 0u);
    h = l0 - 1 + step * cum[s];
    if (++ctx[s] == 0xFF) for (uint32_t i = 0; i < k; ++i) ctx[i] >>= 1;
    shift_in();
    return s;
  }
  bool have_avx2 = __builtin_cpu_supports("avx2") != 0;
#endif
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
