// scan_coder.h -- `bce -s`: ScanCoder<31> (bce.cpp:726-834), the policy that records every coded symbol and
// picks, per plane and per range size k, the number of context bits that minimises the simulated adaptive cost.
// Host C++ (SURVEY section 8f "next #2"): the enumeration that feeds it runs on the GPU (K1-K3 in scan mode),
// the optimisation itself is a host computation.  The result must equal the reference's byte for byte,
// which pins two implementation details: the symbols are kept in a
// std::unordered_map<uint32_t, std::vector<uint8_t>> filled in stream order (its iteration order decides the
// order of the double additions) and the cost is accumulated in double with log().
//
// What is NOT pinned is who does the work.  In the reference's flush every (k, j) pair owns its accumulator z and
// its counter table, and only `z_ += z_min` is ordered over k (bce.cpp:754-796); stat_[k] of different k (and of
// different coders) never meet.  So recording is split by (coder, class of k) and the optimisation by (coder, k, j)
// over a pool of host threads (ScanSet below), every map keeping its insertion sequence and every sum its order:
// the same bytes as the sequential reference, ~10 s -> ~1 s per 10^8 input bytes.
#pragma once
#include <stdint.h>

#include <array>
#include <unordered_map>
#include <vector>

namespace bce {

double scan_add_repeated(double z, double c, uint64_t m);  // `z += c` m times, the loop's own double (scan_coder.cpp)

class ScanCoder {
 public:
  explicit ScanCoder(int i) : nesc_(0), z_(0), i_(i < 0 || i > 7 ? 8 : i) {}    // :733
  void set(uint32_t s, uint32_t k, uint32_t c1, uint32_t c2, uint32_t cs);     // :737-744 (sequential form)
  // :751-800; writes row i_ of `init` (entries never improved keep their value), returns the "Result size" in bytes
  double flush(uint8_t init[9][32]);

  // ---- the pieces ScanSet runs side by side ----
  static constexpr int kClasses = 7;                       // k = 2 | 3 | 4 | 5 | 6..7 | 8..15 | 16..31 (after the escape loop)
  static int class_of(uint32_t k) { return k <= 5 ? (int)k - 2 : k < 8 ? 4 : k < 16 ? 5 : 6; }
  // set() for ONE class of k on a scan_pack word (bce_core.h: the escape loop and the map key were worked out by the kernel
  // that emitted it): the record is kept only if its k is of class `cls`; the thread of class 0 also adds up the escapes
  // (z_ += log(2) each, :739)
  void set_packed(uint32_t word, int cls);
  void prepare_k(uint32_t k) const { prepare(k); }         // group the recorded stream of k (needed before the costs)
  double base_cost(uint32_t k) const;                      // z_min before any j (:757)
  double trial_cost(uint32_t k, uint32_t j) const;         // z of context bits j (:759-785)
  void trial_costs(uint32_t k, double out[6]) const;       // ... all six j in one pass over the symbols (the same doubles)
  uint64_t symbols(uint32_t k) const;                      // records kept for k (task weights)
  // the ordered part of flush (:786-797) from the costs computed above
  double finish(uint8_t init[9][32], const double base[32], const double trial[32][6]);

 private:
  // What stat_[k] of the reference holds (unordered_map<uint32_t, vector<uint8_t>>, :731), kept as the two things it is
  // used for: `order` is that very map with the vector replaced by a group number -- same keys inserted in the same
  // sequence, hence the same iteration order (Q11: it decides the order of the double additions) -- and the symbols go to
  // one append-only stream per k with their group number beside them.  prepare() sorts a stream by group (a stable
  // counting sort: stream order within a group, as push_back gave) in the map's iteration order, so that base and trial
  // costs walk contiguous bytes instead of 10^5 heap vectors per k -- whose filling, walking and freeing was most of the time.
  struct PerK {
    std::unordered_map<uint32_t, uint32_t> order;          // key (:743) -> group number
    std::vector<uint32_t> group_of;                        // [q2 << 8 | q1] -> group number + 1 (0: not seen); sized on first use
    std::vector<uint8_t> sym;                              // the stream
    std::vector<uint16_t> grp;                             // ... and each symbol's group (at most 2^16 keys: both quotients < 256)
    std::vector<uint32_t> gkey;                            // prepare(): keys in iteration order,
    std::vector<uint64_t> gend;                            //   where each group ends in `sorted`,
    std::vector<uint8_t> sorted;                           //   the symbols group by group
    bool ready = false;
  };
  void put(uint32_t k, uint32_t q1, uint32_t q2, uint32_t sym);
  mutable std::array<PerK, 32> stat_;
  void prepare(uint32_t k) const;                          // idempotent; one thread per k at a time
  uint64_t nesc_;          // escapes seen by set_packed (each is one `z_ += log(2)`, added in order by finish)
  double z_;
  int i_;
  friend class ScanSet;
};

// The nine coders of one `bce -s` run (planes 0-7 and the header coder, whose set(s, k) is a no-op: bce.cpp:745-749).
struct ScanSpan { uint64_t start, count; };                // records [start, start + count) of the flush buffer, in stream order
class ScanSet {
 public:
  explicit ScanSet(unsigned threads = 0);                  // 0 = the CPUs this process may run on, at most 64
  // records = one scan_pack word each; spans[p] = plane p's runs of this buffer in stream order
  void consume(const uint32_t *records, const std::vector<ScanSpan> spans[8]);
  void flush(uint8_t init[9][32], double result_bytes[9]);
  unsigned threads() const { return threads_; }
  void release();                                          // drop everything recorded (on the pool: the streams are ~4 B per symbol)

 private:
  std::vector<ScanCoder> coders_;
  unsigned threads_;
};

}  // namespace bce
