// Microbenchmark of the host range decoder + adaptive model step (decoder_core.h) on synthetic queries:
//   g++ -O3 -std=c++17 -I bce_amd/csrc tools/decoder_microbench.cpp bce_amd/csrc/host_coder.cpp -o /tmp/decbench -lpthread && /tmp/decbench
// mode 0: binary contexts only, 1: 75 % binary + k in 3..8, 2: k in 3..12, 3: k in 3..8, 4: k in 9..30, 5: k in 3..4.
#include <chrono>
#include <cstdio>
#include <random>
#include <vector>
#include "../bce_amd/csrc/decoder_core.h"
using namespace bce;
int main() {
  const size_t N = 40000000;
  std::vector<uint16_t> words(N / 4 + 100);
  std::mt19937_64 rng(1);
  for (auto &w : words) w = (uint16_t)rng();
  std::vector<uint32_t> q(N);
  uint8_t row[32]; for (int i = 0; i < 32; ++i) row[i] = i >= 2 && i < 31 ? 4 : 0;
  for (int mode = 0; mode < 6; ++mode) {
    Decoder d; d.open(words.data(), words.size()); d.have_avx2 = getenv("NOAVX") == nullptr;
    for (int b = 0; b < 32; ++b) d.bits[b] = row[b];
    plane_cfg_init(d.cfg, d.bits); d.stat.assign(d.cfg.stat_bytes + 33, 0);
    for (size_t i = 0; i < N; ++i) {
      uint32_t k = mode == 0 ? 2 : (mode == 1 ? (rng() % 4 ? 2 : 3 + rng() % 6) : mode == 2 ? 3 + rng() % 10 : mode == 3 ? 3 + rng() % 6 : mode == 4 ? 9 + rng() % 22 : 3 + rng() % 2);
      uint32_t ctx = rng() % 256;
      q[i] = k | (ctx << 5);
    }
    std::vector<uint32_t> r(N);
    auto t0 = std::chrono::steady_clock::now();
    d.answer_batch(q.data(), nullptr, r.data(), (uint32_t)N);
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    uint64_t sum = 0; for (auto v : r) sum += v;
    printf("mode %d: %.2f ns/symbol (words used %zu, checksum %llu)\n", mode, dt / N * 1e9, d.o, (unsigned long long)sum);
  }
}
