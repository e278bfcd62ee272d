// scan_coder.cpp -- see scan_coder.h.  Plain C++ (no HIP).
#include "scan_coder.h"

#include <cmath>

namespace bce {

void ScanCoder::set(uint32_t s, uint32_t k, uint32_t c1, uint32_t c2, uint32_t cs) {
  // k > 31: one uniform bit is charged and the range halves -- with ScanCoder's OWN formula
  // (k >> 1) + (~s & 1), which differs from AdaptiveCoder's (k + (~s & 1)) >> 1 for even k (quirk Q2)
  while (k > 31u) {
    z_ += std::log(2);
    const uint32_t s0 = s;
    s = s0 >> 1;
    k = (k >> 1) + ((~s0) & 1u);
  }
  const uint32_t key = (((uint32_t)(c2 << 8) / cs) << 16) | ((uint32_t)(c1 << 8) / cs);   // :743 (uint32 wrap kept)
  stat_[k][key].push_back((uint8_t)s);
}

double ScanCoder::flush(uint8_t init[9][32]) {
  std::vector<uint16_t> ctr;
  for (uint32_t k = 2; k < 31u; ++k) {                       // k = 31 is never optimised (:754)
    auto &groups = stat_[k];
    double z_min = 0;
    for (auto &g : groups) z_min += std::log(k) * g.second.size();            // cost with no model at all
    for (uint32_t j = 0; j <= 5; ++j) {                      // candidate context bits
      ctr.assign((size_t)k << (2 * j), 0);
      double z = 0;
      for (auto &g : groups) {
        uint16_t q1 = (uint16_t)(g.first >> 0), q2 = (uint16_t)(g.first >> 16);   // 8-bit quantised c1, c2
        q1 >>= 8 - j;
        q2 >>= 8 - j;
        uint16_t *ctx = &ctr[(size_t)((q1 << j) | q2) * k];
        for (uint8_t sym : g.second) {
          uint32_t l = k;
          for (uint32_t i = 0; i < k; ++i) l += ctx[i];
          z += std::log(static_cast<double>(l) / (1 + ctx[sym]));
          if (++ctx[sym] == 0xFF)
            for (uint32_t i = 0; i < k; ++i) ctx[i] >>= 1;
        }
      }
      if (z < z_min) { z_min = z; init[i_][k] = (uint8_t)j; }
    }
    z_ += z_min;
  }
  return z_ / std::log(256);
}

}  // namespace bce
