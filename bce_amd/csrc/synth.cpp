// synth.cpp -- synthetic inputs of SURVEY.md section 8c (xorshift64*, synth-rand v1, synth-text v1).
// Host-side helpers for benchmarks and tests; exported through the C ABI (include/bce_hip.h).
#include <stdint.h>
#include <stdlib.h>

#include <vector>

#include "../../include/bce_hip.h"

namespace {
inline uint64_t xs_next(uint64_t &st) {
  uint64_t x = st;
  x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
  st = x;
  return x * 0x2545F4914F6CDD1DULL;
}
}  // namespace

extern "C" void bce_hip_synth_rand(uint64_t seed, uint8_t *out, size_t n) {
  uint64_t st = seed;
  size_t k = 0;
  while (k < n) {
    const uint64_t v = xs_next(st);
    for (int b = 0; b < 8 && k < n; ++b) out[k++] = (uint8_t)(v >> (8 * b));
  }
}

extern "C" void bce_hip_synth_text(uint64_t seed, uint8_t *out, size_t n) {
  static const char letters[] = "etaoinshrdlcumwfgypbvkjxqz";
  const int V = 4096;
  uint64_t st = seed;
  std::vector<uint8_t> word((size_t)V * 10), wl(V);
  for (int w = 0; w < V; ++w) {
    const int L = 2 + (int)(xs_next(st) % 8);
    wl[w] = (uint8_t)L;
    for (int c = 0; c < L; ++c) {
      const uint64_t a = xs_next(st) % 26, b = xs_next(st) % 26;
      word[(size_t)w * 10 + c] = (uint8_t)letters[a < b ? a : b];
    }
  }
  size_t k = 0;
  uint64_t cnt = 0;
  while (k < n) {
    const uint64_t r = xs_next(st);
    const uint64_t a = (r >> 32) % V, b = (r & 0xffffffffULL) % V;
    const uint64_t w = (a * b) >> 12;
    for (int c = 0; c < wl[w] && k < n; ++c) out[k++] = word[w * 10 + c];
    ++cnt;
    if (cnt % 13 == 0) { if (k < n) out[k++] = '.'; if (k < n) out[k++] = ' '; }
    else if (k < n) out[k++] = ' ';
  }
}
