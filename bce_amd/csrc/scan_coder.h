// scan_coder.h -- `bce -s`: ScanCoder<31> (bce.cpp:726-834), the policy that records every coded symbol and
// picks, per plane and per range size k, the number of context bits that minimises the simulated adaptive cost.
// Host C++ (SURVEY section 8f "next #2"): the enumeration that feeds it runs on the GPU (K1-K3 in scan mode),
// the optimisation itself is a host computation.  The result must equal the reference's byte for byte,
// which pins two implementation details: the reference keeps the symbols in a
// std::unordered_map<uint32_t, std::vector<uint8_t>> filled in stream order -- the map's ITERATION order decides the
// order of the double additions -- and accumulates the cost in double with log().
//
// What is NOT pinned is how the symbols are stored and who does the work.  In the reference's flush every (k, j) pair
// owns its accumulator z and its counter table, and only `z_ += z_min` is ordered over k (bce.cpp:754-796); stat_[k]
// of different k (and of different coders) never meet.  Here the map is kept without its vectors (same keys, same
// insertion sequence, hence the same iteration order) and the symbols go to one append-only stream per (coder, k);
// recording is split over ranges of a plane's records, the optimisation over (coder, k), on a pool of host threads
// (ScanSet below); every map keeps its insertion sequence and every sum its order: the same bytes as the sequential
// reference, ~10 s -> 0.3-0.5 s per 10^8 input bytes.
#pragma once
#include <stdint.h>

#include <array>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace bce {

// Where the recorded symbols live: anonymous mappings handed out by a bump pointer (huge pages asked for), given back
// as a whole.  A std::vector per stream paid for its growth in page faults and for its end in munmap calls, 16 threads
// at a time on one address space: a third of `bce -s` at 10^8 bytes.
class ScanArena {
 public:
  ScanArena() = default;
  ScanArena(const ScanArena &) = delete;
  ScanArena &operator=(const ScanArena &) = delete;
  ~ScanArena() { release(); }
  uint8_t *take(size_t bytes);                             // zero-filled, 64-byte aligned; any thread; throws std::bad_alloc
  void release();                                          // everything taken so far becomes invalid
 private:
  struct Region { uint8_t *p; size_t cap, used; };
  std::mutex mu_;
  std::vector<Region> regions_;
};

// Small allocations (the nodes and bucket arrays of the order maps) out of the arena: a private 64 KB block at a time, no
// lock on the way; nothing is given back singly.
struct ScanBump {
  ScanArena *arena;
  uint8_t *p = nullptr;
  size_t left = 0;
  void *take(size_t bytes) {
    bytes = (bytes + 15) & ~(size_t)15;
    if (bytes > (size_t)16384) return arena->take(bytes);
    if (bytes > left) { p = arena->take(65536); left = 65536; }
    void *out = p;
    p += bytes;
    left -= bytes;
    return out;
  }
};
template <class T> struct ScanBumpAlloc {
  using value_type = T;
  ScanBump *bump;
  explicit ScanBumpAlloc(ScanBump *b) : bump(b) {}
  template <class U> ScanBumpAlloc(const ScanBumpAlloc<U> &o) : bump(o.bump) {}
  T *allocate(size_t n) { return static_cast<T *>(bump->take(n * sizeof(T))); }
  void deallocate(T *, size_t) {}
  template <class U> bool operator==(const ScanBumpAlloc<U> &o) const { return bump == o.bump; }
  template <class U> bool operator!=(const ScanBumpAlloc<U> &o) const { return bump != o.bump; }
};

double scan_add_repeated(double z, double c, uint64_t m);  // `z += c` m times, the loop's own double (scan_coder.cpp)

class ScanCoder {
 public:
  explicit ScanCoder(int i, std::shared_ptr<ScanArena> arena = nullptr)                 // :733
      : arena_(arena ? std::move(arena) : std::make_shared<ScanArena>()), nesc_(0), z_(0), i_(i < 0 || i > 7 ? 8 : i) {}
  void set(uint32_t s, uint32_t k, uint32_t c1, uint32_t c2, uint32_t cs);     // :737-744 (sequential form)
  // :751-800; writes row i_ of `init` (entries never improved keep their value), returns the "Result size" in bytes
  double flush(uint8_t init[9][32]);

  // ---- the pieces ScanSet runs side by side ----
  void prepare_k(uint32_t k) const { prepare(k); }         // group the recorded stream of k (needed before the costs)
  double base_cost(uint32_t k) const;                      // z_min before any j (:757)
  double trial_cost(uint32_t k, uint32_t j) const;         // z of context bits j (:759-785)
  void trial_costs(uint32_t k, double out[6]) const;       // ... all six j in one pass over the symbols (the same doubles)
  uint64_t symbols(uint32_t k) const;                      // records kept for k (task weights)
  // the ordered part of flush (:786-797) from the costs computed above
  double finish(uint8_t init[9][32], const double base[32], const double trial[32][6]);

 private:
  // What stat_[k] of the reference holds (unordered_map<uint32_t, vector<uint8_t>>, :731), kept as the two things it is
  // used for: `order` is that very map without the vectors -- same keys inserted in the same sequence, hence the same
  // iteration order (Q11: it decides the order of the double additions) -- and the symbols go to one append-only stream
  // per k, each with the 16 bits of its key beside it.  prepare() sorts a stream by key (a stable counting sort: stream
  // order within a key, as push_back gave) in the map's iteration order, so that base and trial costs walk contiguous
  // bytes instead of 10^5 heap vectors per k -- whose filling, walking and freeing was most of the time.
  struct Seg { uint8_t *chunk; uint32_t off, count; };     // symbols [off, off + count) of a chunk
  struct Open { uint8_t *chunk = nullptr; uint32_t fill = kChunk; };   // a chunk being filled (kChunk: none / full)
  struct PerK {
    // The map, its nodes and its buckets live in the arena and go with it (never destroyed one by one: freeing ~10^7
    // nodes cost more than the optimisation); an allocator does not enter the iteration order.  (The value is unused.)
    using Order = std::unordered_map<uint32_t, uint32_t, std::hash<uint32_t>, std::equal_to<uint32_t>,
                                     ScanBumpAlloc<std::pair<const uint32_t, uint32_t>>>;
    Order *order = nullptr;
    uint8_t *seen = nullptr;                               // [q2 << 8 | q1]: key is in `order` (arena, on first use)
    // the stream: pieces of arena chunks in order.  A chunk holds kChunk symbols: uint16_t key16[kChunk] (q2 << 8 | q1:
    // both quotients < 256), then uint8_t sym[kChunk].  Pieces rather than whole chunks because ScanSet::consume lets
    // several threads record disjoint ranges of a plane's records at once, each into chunks of its own.
    std::vector<Seg> segs;
    Open own;                                              // put()'s chunk
    uint64_t n = 0;
    std::vector<uint32_t> gkey;                            // prepare(): keys in iteration order,
    std::vector<uint64_t> gend;                            //   where each key's symbols end in `sorted`,
    uint8_t *sorted = nullptr;                             //   the symbols key by key (arena)
    bool ready = false;
  };
  void learn(PerK &t, uint32_t key16);                     // the key's first symbol: stat_[k][key] creates the entry now
  static constexpr uint32_t kChunk = 8192;
  std::shared_ptr<ScanArena> arena_;
  void put(uint32_t k, uint32_t q1, uint32_t q2, uint32_t sym);
  mutable std::array<PerK, 32> stat_;
  void prepare(uint32_t k) const;                          // idempotent; one thread per k at a time
  uint64_t nesc_;          // escapes seen by ScanSet::consume (each is one `z_ += log(2)`, added in order by finish)
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
  void release();                                          // drop everything recorded

 private:
  // One recording thread's place in a plane: consume() cuts a plane's records into up to kSlots ranges; the thread of
  // range s appends to the chunks of slot s (kept open from one buffer to the next) and notes which keys IT had not seen.
  static constexpr unsigned kSlots = 16;
  struct Slot {
    ScanCoder::Open open[32];
    uint8_t *seen[32] = {};                                // bitmap of the keys this slot has reported (arena, 8 KB)
    std::vector<uint16_t> fresh[32];                       // ... reported in this buffer, in order
    std::vector<ScanCoder::Seg> segs[32];                  // pieces written in this buffer, in order
    uint64_t nesc = 0;
  };
  std::shared_ptr<ScanArena> arena_;
  std::vector<ScanCoder> coders_;
  std::vector<std::unique_ptr<Slot>> slots_;               // [plane][slot]
  unsigned threads_;
  uint64_t min_range_ = 32768;                             // a range shorter than this is not worth a task
};

}  // namespace bce
