// Microbenchmark of the host range coder's bulk loop (RangeCoder::encode_run, host_coder.cpp) on synthetic model records:
//   g++ -O3 -std=c++17 -I bce_amd/csrc tools/encoder_microbench.cpp bce_amd/csrc/host_coder.cpp -o /tmp/encbench -lpthread && /tmp/encbench
// Records come from a simulated adaptive binary / small-alphabet model (skewed symbols: ~1 bit per symbol, like plane 0 of text).
#include <chrono>
#include <cstdio>
#include <random>
#include <vector>
#include "../bce_amd/csrc/bce_core.h"
#include "../bce_amd/csrc/host_coder.h"
using namespace bce;
int main() {
  const size_t N = 40000000;
  std::mt19937_64 rng(7);
  for (int mode = 0; mode < 3; ++mode) {
    std::vector<uint64_t> rec(N);
    const uint32_t k = mode == 0 ? 2 : (mode == 1 ? 4 : 12);
    const int nctx = 4096;
    std::vector<uint8_t> ctr((size_t)nctx * 32, 0);
    for (size_t i = 0; i < N; ++i) {
      const uint32_t c = (uint32_t)(rng() % nctx);
      // skewed symbol: symbol 0 with p = 0.85, the rest geometric
      uint32_t s = 0;
      while (s + 1 < k && (rng() & 1023) >= 870) ++s;
      rec[i] = model_step(&ctr[(size_t)c * 32], k, s, 0u);
    }
    RangeCoder rc;
    auto t0 = std::chrono::steady_clock::now();
    rc.encode_run(rec.data(), 0, N);
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    uint64_t sum = 0; for (auto v : rc.data()) sum = sum * 31 + v;
    printf("k = %2u: %.2f ns/symbol, %.3f bits/symbol, checksum %llu\n", k, dt / N * 1e9, rc.data().size() * 16.0 / N, (unsigned long long)sum);
  }
}
