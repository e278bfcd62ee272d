// scan_coder.h -- `bce -s`: ScanCoder<31> (bce.cpp:726-834), the policy that records every coded symbol and
// picks, per plane and per range size k, the number of context bits that minimises the simulated adaptive cost.
// Host C++ (SURVEY section 8f "next #2"): the enumeration that feeds it runs on the GPU (K1-K3 in scan mode),
// the optimisation itself is a small host computation.  The result must equal the reference's byte for byte,
// which pins two implementation details: the symbols are kept in a
// std::unordered_map<uint32_t, std::vector<uint8_t>> filled in stream order (its iteration order decides the
// order of the double additions) and the cost is accumulated in double with log().
#pragma once
#include <stdint.h>

#include <array>
#include <unordered_map>
#include <vector>

namespace bce {

class ScanCoder {
 public:
  explicit ScanCoder(int i) : z_(0), i_(i < 0 || i > 7 ? 8 : i) {}             // :733
  void set(uint32_t s, uint32_t k, uint32_t c1, uint32_t c2, uint32_t cs);    // :737-744
  // :751-800; writes row i_ of `init` (entries never improved keep their value), returns the "Result size" in bytes
  double flush(uint8_t init[9][32]);

 private:
  std::array<std::unordered_map<uint32_t, std::vector<uint8_t>>, 32> stat_;
  double z_;
  int i_;
};

}  // namespace bce
