// host_coder.cpp -- see host_coder.h.  Plain C++ (no HIP).
#include "host_coder.h"

#include <thread>

namespace bce {

// Default context-bit tables, AdaptiveCoder<31>::init_ (bce.cpp:713-724): format constants that are
// also serialised into every stream's preamble (bce.cpp:682-691).
const uint8_t kDefaultConfig[9][32] = {
    {0, 0, 5, 5, 5, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 0},
    {0, 0, 5, 5, 5, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 0},
    {0, 0, 5, 5, 5, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 3, 3, 3, 3, 0},
    {0, 0, 5, 5, 5, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 3, 3, 3, 3, 3, 3, 3, 3, 3, 0},
    {0, 0, 5, 5, 4, 4, 4, 4, 4, 4, 4, 4, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 0},
    {0, 0, 5, 5, 4, 4, 4, 4, 4, 4, 4, 4, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 0},
    {0, 0, 5, 4, 4, 4, 4, 4, 4, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 0},
    {0, 0, 4, 4, 4, 4, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 2, 2, 2, 2, 2, 2, 0},
    {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}};

void RangeCoder::setv(uint32_t s) {
  while (s) { uniform(s & 1, 3); s >>= 1; }
  uniform(2, 3);
}

void RangeCoder::preamble(const uint8_t row[32]) {
  uint32_t last = 0;
  for (int b = 0; b < 32; ++b) {
    const uint32_t bit = row[b];
    uniform(bit != last, 2);
    if (bit != last) uniform(bit, 6);
    last = bit;
  }
}

void RangeCoder::flush() {
  while (!((h_ ^ l_) >> 48)) {                        // shift_out() :611 (a no-op after encode())
    data_.push_back((uint16_t)(h_ >> 48));
    l_ = (l_ << 16) + 0x0000;
    h_ = (h_ << 16) + 0xFFFF;
  }
  const uint32_t bits = (uint32_t)__builtin_clzll(l_ ^ h_) + 1;
  data_.push_back((uint16_t)((h_ >> (64 - bits)) << (16 - bits)));
}

void HostCoder::begin(const uint8_t config[9][32], const uint32_t C[8], uint32_t n) {
  for (int i = 0; i < 8; ++i) {
    plane[i] = RangeCoder();
    plane[i].preamble(config[i]);
    plane[i].uniform(C[i], n + 1);
  }
}

void HostCoder::consume(int p, const SymRun *runs, size_t nruns, const uint64_t *out, const uint32_t *esc) {
  RangeCoder &rc = plane[p];
  for (size_t r = 0; r < nruns; ++r) {
    const uint64_t b = runs[r].start, e = b + runs[r].count;
    for (uint64_t i = b; i < e; ++i) {
      const uint64_t o = out[i];
      uint32_t bits = esc[i];                           // [26:0] escape bits, [31:27] their count
      if (__builtin_expect(bits != 0, 0)) {             // k > 31 escape, bce.cpp:507-510
        for (uint32_t nesc = bits >> 27; nesc; --nesc, bits >>= 1) rc.uniform(bits & 1, 2);
      }
      rc.encode((uint32_t)(o & 0xFFFF), (uint32_t)((o >> 16) & 0xFFFF), (uint32_t)((o >> 32) & 0xFFFF));
    }
  }
}

void HostCoder::consume_all(const std::vector<SymRun> runs[8], const uint64_t *out, const uint32_t *esc, int threads) {
  if (threads <= 1) {
    for (int p = 0; p < 8; ++p) consume(p, runs[p].data(), runs[p].size(), out, esc);
    return;
  }
  std::thread th[8];
  for (int p = 0; p < 8; ++p)
    th[p] = std::thread([this, p, runs, out, esc]() { consume(p, runs[p].data(), runs[p].size(), out, esc); });
  for (int p = 0; p < 8; ++p) th[p].join();
}

void HostCoder::finish(const uint8_t config[9][32], uint32_t n, uint32_t offset, std::vector<uint16_t> &archive) {
  unsigned size = 0u;                                   // :1134-1138
  for (int i = 0; i < 8; ++i) { plane[i].flush(); size += (unsigned)plane[i].data().size(); }
  RangeCoder mainc;                                     // coder_type main(-1) :1141 -> config row 8
  mainc.preamble(config[8]);
  mainc.setv(n);
  mainc.uniform(offset, n + 1);
  mainc.setv(size);
  int s = (int)size;
  for (int i = 0; i < 7; ++i) {
    mainc.uniform((uint32_t)plane[i].data().size(), (uint32_t)s + 1);
    s -= (int)plane[i].data().size();
  }
  mainc.flush();
  archive.clear();                                      // :1152-1157
  archive.push_back((uint16_t)mainc.data().size());
  archive.insert(archive.end(), mainc.data().begin(), mainc.data().end());
  for (int i = 0; i < 8; ++i) archive.insert(archive.end(), plane[i].data().begin(), plane[i].data().end());
}

}  // namespace bce
