// bisect the cost of the binary decode step: variants by -DV=...
#include <chrono>
#include <cstdio>
#include <random>
#include <vector>
#include "decoder_core.h"
using namespace bce;
#ifndef V
#define V 0
#endif
int main() {
  const size_t N = 40000000;
  std::vector<uint16_t> words(N / 4 + 100);
  std::mt19937_64 rng(1);
  for (auto &w : words) w = (uint16_t)rng();
  std::vector<uint32_t> q(N), r(N);
  for (size_t i = 0; i < N; ++i) q[i] = (uint32_t)(rng() % 256);
  std::vector<uint8_t> stat(4096, 0);
  const Recip *rt = recip_table();
  const uint16_t *data = words.data(); const size_t size = words.size();
  Decoder::St t{0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) t.D = (t.D << 16) + Decoder::word_at(data, size, t);
  auto t0 = std::chrono::steady_clock::now();
  for (size_t i = 0; i < N; ++i) {
    uint8_t *ctx = stat.data() + q[i] * 2;
    const uint32_t c0 = ctx[0], c1 = ctx[1], tot = c0 + c1 + 2u;
    if (__builtin_expect(t.R - 1 < tot, 0)) Decoder::st_reset(data, size, t);
#if V == 2
    const uint64_t step = (t.R - 1) >> 9;
#else
    const uint64_t step = div_small(t.R - 1, tot, rt);
#endif
    const uint64_t x0 = step * ((uint64_t)c0 + 1), x1 = step * ((uint64_t)c1 + 1);
#if V == 3 || V == 4 || V == 5
    uint64_t lo = 0, R = x0; uint32_t s;
    __asm__("xorl %k[s], %k[s]\n\tcmpq %[x0], %[D]\n\tcmovaeq %[x0], %[lo]\n\tcmovaeq %[x1], %[R]\n\tsetae %b[s]" : [lo] "+r"(lo), [R] "+r"(R), [s] "=&q"(s) : [D] "r"(t.D), [x0] "r"(x0), [x1] "r"(x1) : "cc");
    t.l += lo; t.D -= lo; t.R = R;
#else
    const uint32_t s = t.D >= x0 ? 1u : 0u;
    const uint64_t mask = 0ull - (uint64_t)s;
    const uint64_t lo = x0 & mask;
    t.l += lo; t.D -= lo;
    t.R = x0 ^ ((x0 ^ x1) & mask);
#endif
#if V == 1
    (void)ctx;
#elif V == 4 || V == 5
    { uint16_t w; __builtin_memcpy(&w, ctx, 2); w = (uint16_t)(w + (1u << (8 * s))); __builtin_memcpy(ctx, &w, 2);
      if (__builtin_expect(((w >> (8 * s)) & 0xFF) == 0xFF, 0)) { ctx[0] >>= 1; ctx[1] >>= 1; } }
#else
    if (__builtin_expect(++ctx[s] == 0xFF, 0)) { ctx[0] >>= 1; ctx[1] >>= 1; }
#endif
#if V == 5
    if (__builtin_expect(!(((t.l + t.R - 1) ^ t.l) >> 48), 0)) { t.D = (t.D << 16) + Decoder::word_at(data, size, t); t.l <<= 16; t.R <<= 16;
      while (__builtin_expect(!(((t.l + t.R - 1) ^ t.l) >> 48), 0)) { t.D = (t.D << 16) + Decoder::word_at(data, size, t); t.l <<= 16; t.R <<= 16; } }
#else
    Decoder::st_shift_in(data, size, t);
#endif
    r[i] = s;
  }
  double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  uint64_t sum = 0; for (auto v : r) sum += v;
  printf("V=%d: %.2f ns/symbol (%.1f cycles at 2.73 GHz), words %zu checksum %llu\n", V, dt / N * 1e9, dt / N * 2.73e9, t.o, (unsigned long long)sum);
}
