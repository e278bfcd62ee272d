// decoder_core.h -- the sequential half of the decoder, shared by the host decoder (decoder.cpp) and the
// GPU-assisted one (kd_decode.hip): AdaptiveCoder<31>'s decode side -- range decoder (bce.cpp:592-608, shift_in
// :663-669, ctor :495-504), adaptive model get (bce.cpp:555-590), VCoder::getv (bce.cpp:372-377), init(0, i)
// (bce.cpp:692-705).  Plain host C++.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <vector>

#include "bce_core.h"
#include "host_coder.h"

namespace bce {

constexpr uint32_t kUnknown = 0xFFFFFFFFu;

// AdaptiveCoder<31>, decode side.  One per plane, each driven by its own thread in the GPU-assisted decoder: own cache
// lines (the state and the read position change with every symbol).
//
// State: the reference keeps (l, h, m) -- range bounds and the 64 archive bits under the cursor (bce.cpp:495-504).  Here
// it is (l, R = h - l + 1, D = m - l): the same numbers, but a symbol's dependency chain is then R -> step -> R'
// and D -> D' (one single-multiply division, one multiply, one compare each), with l updated off the chain --
// the form the encoder's host coder uses.  R == 0 stands for the full range (h - l = 2^64 - 1).  shift_in
// (:663-669) becomes l <<= 16, R <<= 16, D = (D << 16) + next word.  Every quotient and product below is the one
// the reference computes.
struct alignas(128) Decoder {
  uint64_t l = 0, R = 0, D = 0;
  const uint16_t *data = nullptr;
  size_t size = 0, o = 0;
  PlaneCfg cfg;
  uint8_t bits[32];
  std::vector<uint8_t> stat;
  bool overrun = false;

  uint16_t next() { const uint16_t v = o < size ? data[o] : 0; ++o; return v; }   // reads past the end give 0 (:568)
  void open(const uint16_t *d, size_t n) {                       // ctor :495-504: the first 4 words, missing ones as 0
    data = d; size = n; o = 0; l = 0; R = 0; D = 0;
    for (int i = 0; i < 4; ++i) D = (D << 16) + next();
  }
  void reset() { for (int i = 0; i < 4; ++i) D = (D << 16) + next(); l = 0; R = 0; }   // :565-570: m = 4 fresh words, l = 0, h = ~0
  void shift_in() {                                              // :663-669
    while (!(((l + R - 1) ^ l) >> 48)) {
      D = (D << 16) + next();
      l <<= 16;
      R <<= 16;
    }
  }
  uint32_t get(uint32_t k) {                                     // :592-608
    if (k == 0) return 0;                                        // (only a crafted header asks this: no division by zero)
    if (R - 1 < k) reset();                                      // h - l < k (never with the full range: R - 1 wraps to 2^64 - 1)
    const uint64_t step = (R - 1) / k;
    const uint32_t s = (uint32_t)(D / step);
    l += step * s;
    D -= step * s;
    R = step;                                                    // h = step + l - 1
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
  // the same with the context already resolved (k <= 31): the GPU-assisted decoder computes ctxv on the device.
  // The reference walks s = 0, 1, ... with h_s = l - 1 + step cum_s until h_s >= m or s = k - 1 (:577-583);
  // h_s >= m  <=>  step cum_s > D.
  // The hot state as a plain local struct: a loop that keeps it in registers (answer_batch) runs the chain without a
  // store-to-load round trip through the Decoder object on every link.
  struct St { uint64_t l, R, D; size_t o; };
  St save() const { return St{l, R, D, o}; }
  void restore(const St &t) { l = t.l; R = t.R; D = t.D; o = t.o; }
  static __attribute__((always_inline)) inline uint64_t word_at(const uint16_t *data, size_t size, St &t) {
    const uint64_t v = t.o < size ? data[t.o] : 0; ++t.o; return v;
  }
  static __attribute__((always_inline)) inline void st_reset(const uint16_t *data, size_t size, St &t) {
    for (int i = 0; i < 4; ++i) t.D = (t.D << 16) + word_at(data, size, t);
    t.l = 0; t.R = 0;
  }
  static __attribute__((always_inline)) inline void st_shift_in(const uint16_t *data, size_t size, St &t) {
    while (__builtin_expect(!(((t.l + t.R - 1) ^ t.l) >> 48), 0)) {
      t.D = (t.D << 16) + word_at(data, size, t);
      t.l <<= 16;
      t.R <<= 16;
    }
  }
  // one uniform bit, get(2) of bce.cpp:592-608, on a local state: step = (R - 1) / 2 is a shift, and s = D / step is 0 or 1
  // for every stream an encoder wrote (a corrupt one can ask for more: the division then)
  static __attribute__((always_inline)) inline uint32_t st_bit(const uint16_t *data, size_t size, St &t) {
    if (__builtin_expect(t.R - 1 < 2u, 0)) st_reset(data, size, t);
    const uint64_t step = (t.R - 1) >> 1;
    uint64_t s = t.D >= step ? 1u : 0u;
    if (__builtin_expect(t.D - step >= step && s, 0)) s = t.D / step;
    t.l += step * s;
    t.D -= step * s;
    t.R = step;
    st_shift_in(data, size, t);
    return (uint32_t)s;
  }
  // one adaptive symbol with k <= 8 or without AVX2 (scalar), on a local state
  static __attribute__((always_inline)) inline uint32_t st_slot(const uint16_t *data, size_t size, const Recip *recip, St &t,
                                                              uint8_t *ctx, uint32_t k) {
    if (k == 2) {
      // binary contexts (most symbols).  Measured step by step (tools/decoder_step_bisect.cpp, 40 M coin flips): the symbol
      // taken by cmp + cmov instead of setcc / neg / and / xor masks: 25.3 -> 24.0 cycles; and, the big one, the counter
      // written back as the slot's two bytes at the slot's OWN address instead of one byte at ctx + s: 24.0 -> 17.4 --
      // a store whose address hangs on the decoded symbol keeps the next symbols' counter loads waiting until it is known
      // (they might alias), which strings load -> total -> reciprocal -> multiply behind every symbol.
      uint16_t w;
      __builtin_memcpy(&w, ctx, 2);
      const uint32_t c0 = w & 0xFFu, c1 = w >> 8, tot = c0 + c1 + 2u;
      if (__builtin_expect(t.R - 1 < tot, 0)) st_reset(data, size, t);
      const uint64_t step = div_small(t.R - 1, tot, recip);       // tot < 8192: exact single-multiply division
      const uint64_t x0 = step * ((uint64_t)c0 + 1), x1 = step * ((uint64_t)c1 + 1);
      uint64_t lo = 0, R = x0;
      uint32_t s;
#if defined(__x86_64__)
      __asm__("xorl %k[s], %k[s]\n\tcmpq %[x0], %[D]\n\tcmovaeq %[x0], %[lo]\n\tcmovaeq %[x1], %[R]\n\tsetae %b[s]"
              : [lo] "+r"(lo), [R] "+r"(R), [s] "=&q"(s) : [D] "r"(t.D), [x0] "r"(x0), [x1] "r"(x1) : "cc");
#else
      s = t.D >= x0 ? 1u : 0u;
      { const uint64_t mask = 0ull - (uint64_t)s; lo = x0 & mask; R = x0 ^ ((x0 ^ x1) & mask); }
#endif
      t.l += lo;
      t.D -= lo;
      t.R = R;
      w = (uint16_t)(w + (1u << (8u * s)));
      __builtin_memcpy(ctx, &w, 2);
      if (__builtin_expect(((w >> (8u * s)) & 0xFFu) == 0xFFu, 0)) { ctx[0] >>= 1; ctx[1] >>= 1; }
      st_shift_in(data, size, t);
      return s;
    }
    if (k <= 8) {
      // small alphabets, one straight-line path for k = 3..8.  Everything that hangs on the counters alone -- the eight
      // bytes of the slot in one load (the neighbours' bytes masked off), counter + 1 per byte, their prefix sums a_1..a_8,
      // the total -- is off the chain R -> step -> R' (the core works it out while the symbol before is still in flight).
      // On the chain: step, then SEVEN INDEPENDENT products step * a_i (the first version walked s = 0, 1, ... with a
      // multiply-add and a masked compare per step: 7 x 7 dependent cycles, 26 ns per such symbol against 9 for a binary
      // one), the symbol = the number of boundaries at or below D (those beyond k - 1 are compared against all ones: the
      // walk of bce.cpp:577-583 stops at s = k - 1 whatever D is), and two more products for the new range.
      uint64_t xraw;
      __builtin_memcpy(&xraw, ctx, 8);
      const uint64_t km = ~0ull >> (64 - 8 * k);
      const uint64_t y = (xraw & km) + (0x0101010101010101ull & km);          // byte i = b_i = counter i + 1 (<= 255) for i < k, else 0
      // prefix sums a_i = b_0 + .. + b_(i-1) in 16-bit lanes (a_8 <= 2040): pairs, their running sums by one multiplication,
      // and the odd ones from those -- a dozen operations instead of a shift, a mask and an add per byte
      const uint64_t ev = y & 0x00FF00FF00FF00FFull, od = (y >> 8) & 0x00FF00FF00FF00FFull;
      const uint64_t pc = (ev + od) * 0x0001000100010001ull;                  // lanes: a_2, a_4, a_6, a_8
      const uint64_t po = (pc << 16) + ev;                                    // lanes: a_1, a_3, a_5, a_7
      alignas(8) uint16_t A[12];                                              // A[i] = a_i (A[0] = 0)
      __builtin_memcpy(&A[4], &pc, 8);                                        // (interleaved below)
      const uint32_t a1 = (uint32_t)po & 0xFFFFu, a2 = (uint32_t)pc & 0xFFFFu, a3 = (uint32_t)(po >> 16) & 0xFFFFu, a4 = (uint32_t)(pc >> 16) & 0xFFFFu,
                     a5 = (uint32_t)(po >> 32) & 0xFFFFu, a6 = (uint32_t)(pc >> 32) & 0xFFFFu, a7 = (uint32_t)(po >> 48), tot = (uint32_t)(pc >> 48);
      A[0] = 0; A[1] = (uint16_t)a1; A[2] = (uint16_t)a2; A[3] = (uint16_t)a3; A[4] = (uint16_t)a4; A[5] = (uint16_t)a5; A[6] = (uint16_t)a6; A[7] = (uint16_t)a7; A[8] = (uint16_t)tot;
      if (__builtin_expect(t.R - 1 < tot, 0)) st_reset(data, size, t);
      const uint64_t step = div_small(t.R - 1, tot, recip);
      const uint64_t D = t.D;
      // the sums saturate at the total behind the last counter, so a boundary beyond k - 1 is step * total: D reaches it
      // only from the unused top of the range, where every real boundary counts as well -- the cap at k - 1 does the rest
      // (measured on the EPYC 9575F of the GPU boxes and dropped: the eight boundaries in ONE AVX-512 multiply and the symbol
      //  from one compare mask -- ~30 instructions instead of ~100, 8.70 against 8.82 ns per symbol on a real stream: nothing)
      const uint32_t cnt = (uint32_t)(D >= step * a1) + (uint32_t)(D >= step * a2) + (uint32_t)(D >= step * a3) + (uint32_t)(D >= step * a4) +
                           (uint32_t)(D >= step * a5) + (uint32_t)(D >= step * a6) + (uint32_t)(D >= step * a7);
      const uint32_t s = cnt < k - 1u ? cnt : k - 1u;
      const uint64_t lo = step * A[s];
      t.l += lo;
      t.D = D - lo;
      t.R = step * (uint64_t)(A[s + 1] - A[s]);
      // (the eight bytes go back where they came from, the neighbours' as they were: see the binary case; this decoder's
      //  thread is the only one that touches its counters, and the array has 32 bytes of slack behind the last slot)
      const uint64_t up = xraw + (1ull << (8 * s));
      __builtin_memcpy(ctx, &up, 8);
      if (__builtin_expect(((up >> (8 * s)) & 0xFFu) == 0xFFu, 0)) for (uint32_t i = 0; i < k; ++i) ctx[i] >>= 1;
      st_shift_in(data, size, t);
      return s;
    }
    uint32_t tot = k;
    for (uint32_t i = 0; i < k; ++i) tot += ctx[i];
    if (__builtin_expect(t.R - 1 < tot, 0)) st_reset(data, size, t);
    const uint64_t step = div_small(t.R - 1, tot, recip);
    uint64_t lo = 0, hi = step * ((uint64_t)ctx[0] + 1);
    uint32_t s = 0;
    while (hi <= t.D && s + 1 < k) {
      ++s;
      lo = hi;
      hi += step * ((uint64_t)ctx[s] + 1);
    }
    t.l += lo;
    t.D -= lo;
    t.R = hi - lo;
    if (__builtin_expect(++ctx[s] == 0xFF, 0)) for (uint32_t i = 0; i < k; ++i) ctx[i] >>= 1;
    st_shift_in(data, size, t);
    return s;
  }
  uint32_t get_slot(uint32_t k, uint32_t ctxv) {
    uint8_t *ctx = stat.data() + cfg.off[k] + ctxv * k;
#if defined(__x86_64__)
    if (have_avx2 && k > 8) return get_slot_avx2(ctx, k);
#endif
    St t = save();
    const uint32_t s = st_slot(data, size, recip, t, ctx, k);
    restore(t);
    return s;
  }
  // A run of queries of one plane in stream order (the GPU-assisted decoder's rounds): q[i] = k | ctx << 5, or
  // kEscapeQuery with the next record of e = (k, c1, c2, cs).  The state stays in registers across the run.
  static constexpr uint32_t kEscapeQuery = 0x80000000u;
  struct Esc { uint32_t k, c1, c2, cs; };
  void answer_batch(const uint32_t *q, const Esc *e, uint32_t *r, uint32_t cnt) {
    St t = save();
    uint8_t *const sbase = stat.data();
    const uint16_t *const dat = data;
    const size_t sz = size;
    const Recip *const rc = recip;
    const uint32_t off2 = cfg.off[2];
    for (uint32_t i = 0; i < cnt; ++i) {
      if (i + 8 < cnt) {                                         // the counters of a query soon to come
        const uint32_t qn = q[i + 8];
        if (!(qn & kEscapeQuery)) __builtin_prefetch(sbase + cfg.off[qn & 31u] + (qn >> 5) * (qn & 31u));
      }
      const uint32_t qi = q[i];
      const uint32_t k = qi & 31u;
      if (__builtin_expect((qi & (kEscapeQuery | 31u)) == 2u, 1)) {        // binary slots: six symbols in ten
        r[i] = st_slot(dat, sz, rc, t, sbase + off2 + (qi >> 5) * 2u, 2u);
        continue;
      }
#if defined(__x86_64__)
      const bool wide = have_avx2 && k > 8;
#else
      const bool wide = false;
#endif
      if (__builtin_expect((qi & kEscapeQuery) != 0, 0)) {
        // k > 31 (bce.cpp:557-560): uniform bits, LSB first, each halving the range, then the slot of the k that is left
        uint32_t kk = e->k, bits = 0, nb = 0;
        while (kk > (uint32_t)kMaxK) {
          const uint32_t sb = st_bit(dat, sz, t);
          bits |= sb << nb; ++nb;
          kk = (kk + (~sb & 1u)) >> 1;
        }
        const uint32_t b = cfg.bits[kk];
        const uint32_t ctxv = context_index(b, e->c1, e->c2, e->cs);   // :671-677 (bce_core.h: each quotient by one float reciprocal, exact)
        uint8_t *ctx = sbase + cfg.off[kk] + ctxv * kk;
        uint32_t top;
        if (have_wide_path() && kk > 8) { restore(t); top = get_slot(kk, ctxv); t = save(); }
        else top = st_slot(dat, sz, rc, t, ctx, kk);
        r[i] = (top << nb) | bits;
        ++e;
        continue;
      }
      if (__builtin_expect(wide, 0)) {                            // k = 9..31 with AVX2: on the object
        restore(t);
        r[i] = get_slot(k, qi >> 5);
        t = save();
        continue;
      }
      r[i] = st_slot(dat, sz, rc, t, sbase + cfg.off[k] + (qi >> 5) * k, k);
    }
    restore(t);
  }
#if defined(__x86_64__)
  // The same step for larger k without the two data-dependent loops: cum[i] = sum_{j<=i} (ctx[j] + 1) for all 32
  // lanes at once (16-bit prefix sums), then the symbol is the number of i < k with cum[i] < t, where
  // t = floor(D / step) + 1 (step cum[i] > D  <=>  cum[i] >= t), capped at k - 1 as the loop is.
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
    if (R - 1 < tot) reset();
    const uint64_t step = div_small(R - 1, tot, recip);
    const uint64_t tq = D / step + 1;
    const __m256i t16 = _mm256_set1_epi16((short)(tq > 16384 ? 16384 : tq));                        // cum <= 8192
    const uint32_t lt_lo = (uint32_t)_mm256_movemask_epi8(_mm256_and_si256(_mm256_cmpgt_epi16(t16, lo), m_lo));
    const uint32_t lt_hi = (uint32_t)_mm256_movemask_epi8(_mm256_and_si256(_mm256_cmpgt_epi16(t16, hi), m_hi));
    uint32_t s = (uint32_t)(__builtin_popcount(lt_lo) + __builtin_popcount(lt_hi)) >> 1;             // two mask bits per lane
    if (s > k - 1) s = k - 1;
    const uint64_t below = step * (uint64_t)(cum[(s + 31u) & 31u] & (0u - (uint32_t)(s != 0)));   // cum[s - 1], 0 for s = 0
    l += below;
    D -= below;
    R = step * ((uint64_t)ctx[s] + 1);
    if (++ctx[s] == 0xFF) for (uint32_t i = 0; i < k; ++i) ctx[i] >>= 1;
    shift_in();
    return s;
  }
  bool have_avx2 = __builtin_cpu_supports("avx2") != 0;
  bool have_wide_path() const { return have_avx2; }
#else
  bool have_wide_path() const { return false; }
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
  // n < 2^31 is what the encoder accepts (saidx_t) and what every size computed from n assumes; a crafted header can
  // decode to more (getv reads up to 31 bits and a slack symbol), and n + 1 would wrap to 0 in the next get()
  if (hd.n == 0 || hd.n >= 0x80000000u) return -1;
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
