// host_coder.h -- the sequential half of the encoder that stays on the host (north star): the 64-bit
// carry-less range coder of AdaptiveCoder (bce.cpp:520-529,538-553,610-615,655-661), the stream
// preambles (bce.cpp:682-691,1126-1130), VCoder::setv (bce.cpp:364-370) and the archive framing of
// BCE::encode (bce.cpp:1134-1157).  It consumes (cum, freq, total) records computed by the GPU model
// kernel (K4); it never touches the adaptive counters itself.  No HIP dependency.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace bce {

// Each coder is driven by its own thread: keep every instance on its own cache lines (the l_/h_ state is
// rewritten on every symbol; eight 40-byte coders side by side false-share and run ~6x slower).
class alignas(128) RangeCoder {
 public:
  RangeCoder() : l_(0), h_(~0ull) {}
  // one coding step: the shared tail of set(s,k) (cum=s, freq=1, total=k) and set(s,k,c1,c2,cs)
  inline void encode(uint32_t cum, uint32_t freq, uint32_t total) {
    if (__builtin_expect(h_ - l_ < total, 0)) {        // :520-525 / :541-546
      for (int i = 0; i < 4; ++i) data_.push_back((uint16_t)(l_ >> (48 - 16 * i)));
      l_ = 0; h_ = ~0ull;
    }
    const uint64_t step = (h_ - l_) / total;            // :527 / :548
    l_ += step * cum;                                   // :528 / :549
    h_ = l_ + step * freq - 1;                          // :529 / :550
    while (!((h_ ^ l_) >> 48)) {                        // shift_out :655-661
      data_.push_back((uint16_t)(h_ >> 48));
      l_ = (l_ << 16) + 0x0000;
      h_ = (h_ << 16) + 0xFFFF;
    }
  }
  inline void uniform(uint32_t s, uint32_t k) { encode(s, 1, k); }
  void setv(uint32_t s);                                // VCoder::setv :364-370
  void preamble(const uint8_t row[32]);                 // init(1, i) :682-691
  void flush();                                         // :610-615
  const std::vector<uint16_t> &data() const { return data_; }
  std::vector<uint16_t> &data() { return data_; }
  // bulk form of encode() for the GPU model records of one run (state kept in registers)
  void encode_run(const uint64_t *out, uint64_t begin, uint64_t end);

 private:
  uint64_t l_, h_;
  std::vector<uint16_t> data_;
};

// One GPU model record per symbol: bce_core.h pack_model_out (cum, freq, total and the k > 31 escape bits).
struct SymRun { uint64_t start; uint32_t count; uint32_t round; };

// One model flush handed to the coder threads: the GPU outputs of its records (pinned host memory
// owned by the caller) and, per plane, the (round-ordered) runs of record indices that belong to it.
struct CoderBatch {
  const uint64_t *out = nullptr;
  std::function<void()> wait_ready;  // optional: returns once `out` is filled (the device-to-host copy is asynchronous)
  std::vector<SymRun> runs[8];
  std::atomic<int> pending{0};      // planes that have not finished this batch yet
};

struct HostCoder {
  RangeCoder plane[8];
  // Planes whose records this coder codes (default: all eight).  One archive from several GPUs (SURVEY section 8e-2's aim, by
  // the split that costs least here): every rank enumerates and models everything -- the GPU's part is a fifth of a step --
  // but runs the sequential range coders, which ARE the step, only for its own planes; the finished streams of the other
  // planes come from their owners (set_stream) before the header is coded.
  uint32_t plane_mask = 0xFFu;
  void set_stream(int p, const uint16_t *words, size_t n) { plane[p].data().assign(words, words + n); }
  void rebuild_header(const uint8_t config[9][32], uint32_t n, uint32_t offset);   // the header part of finish() (:1141-1150), again
  HostCoder();
  ~HostCoder();
  // BCE::encode :1124-1130: construct the 8 coders (preamble from config rows 0..7) and code
  // C[i] = zeros(plane (i+7)%8) with range n+1.
  void begin(const uint8_t config[9][32], const uint32_t C[8], uint32_t n);
  // Feed plane p's records of one flush, run by run (coder_[i].set(...), bce.cpp:1302, coder half).
  void consume(int p, const SymRun *runs, size_t nruns, const uint64_t *out);
  // Asynchronous form: 8 persistent threads, one per plane (the reference's OpenMP loop :1250-1252 has
  // the same 8-way split); batches are coded in submission order while the GPU produces the next one.
  void submit(CoderBatch *b);
  void wait(CoderBatch *b);          // until every plane has finished batch b
  void drain();                      // until every submitted batch is finished
  // BCE::encode :1134-1157: flush, header coder main(-1) (config row 8), concatenate.
  void finish(const uint8_t config[9][32], uint32_t n, uint32_t offset);      // flush + header; archive_words / assemble lay it out
  size_t archive_words() const;
  void assemble(uint16_t *dst) const;
  void finish(const uint8_t config[9][32], uint32_t n, uint32_t offset, std::vector<uint16_t> &archive);   // all in one (tests)
  double busy_seconds();             // max over planes of the time spent coding since begin()
  bool failed() const { return failed_.load(); }   // a coder thread ran out of memory since begin()

 private:
  struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<CoderBatch *> q;
    double busy = 0;
    uint64_t nsym = 0;
    bool stop = false;
  };
  Worker w_[8];
  std::mutex done_mu_;
  std::condition_variable done_cv_;
  std::vector<uint16_t> header_;             // the coded header of the last finish()
  uint64_t submitted_ = 0, completed_ = 0;   // in units of (batch, plane); guarded by done_mu_
  std::atomic<bool> failed_{false};
  void run(int p);
  void stop_threads();
};

// Exact floor(x / d) for d < kRecipMax by one multiplication (host_coder.cpp explains the two magic variants).
struct Recip { uint64_t m; int32_t am1; uint32_t sh; };   // am1 = add - 1: the multiplicand is (x + 1) + am1
constexpr uint32_t kRecipMax = 8192;
const Recip *recip_table();
inline uint64_t div_recip1(uint64_t x1, const Recip &r) {   // floor((x1 - 1) / d) for x1 = x + 1 != 0
  return (uint64_t)(((unsigned __int128)r.m * (x1 + (uint64_t)(int64_t)r.am1)) >> 64) >> r.sh;
}
inline uint64_t div_small(uint64_t x, uint32_t d, const Recip *rt) {   // any x; table for d < kRecipMax
  return (x == ~0ull || d >= kRecipMax) ? x / d : div_recip1(x + 1, rt[d]);
}

extern const uint8_t kDefaultConfig[9][32];
uint64_t bce_test_div_recip(uint64_t x, uint32_t d);   // exposed for tests/core_emul.cpp             // AdaptiveCoder<31>::init_ :713-724

}  // namespace bce
