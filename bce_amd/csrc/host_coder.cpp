// host_coder.cpp -- see host_coder.h.  Plain C++ (no HIP).
#include "host_coder.h"

#include "bce_core.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
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
  drain();
  failed_.store(false);
  for (int i = 0; i < 8; ++i) {
    w_[i].busy = 0; w_[i].nsym = 0;
    plane[i] = RangeCoder();
    plane[i].preamble(config[i]);
    plane[i].uniform(C[i], n + 1);
  }
}

// Exact floor(x / d) for 64-bit x by ONE multiplication (divisor-invariant division, N = 64).
// With L = floor(log2 d), s = 64 + L, m_dn = floor(2^s / d), e' = 2^s - m_dn d, e = d - e':
//   * e  <= 2^L : q = mulhi(m_dn + 1, x)     >> L   is exact for every x < 2^64            (round-up magic)
//   * e' <= 2^L : q = mulhi(m_dn,     x + 1) >> L   is exact for every x < 2^64 - 1        (round-down magic)
// e + e' = d <= 2^(L+1), so one of the two always applies.  (Error terms: m (x+a) / 2^s = (x+a)/d (1 +- e/2^s);
// at x = kd + d - 1 resp. x = kd the deviation stays below 1/d.)  Powers of two use m = 2^(64-L), shift 0.
// The model totals are < 8192 (k <= 31 counters <= 254, + k), so a table indexed by the divisor replaces the
// 64-bit divide on the coder's dependency chain (step = (h - l) / total, bce.cpp:527); the quotient is
// bit-identical.  x = 2^64 - 1 (right after a range reset) and larger divisors take the real divide.
const Recip *recip_table() {
  static const std::vector<Recip> tab = [] {
    std::vector<Recip> t(kRecipMax);
    t[0] = t[1] = Recip{0, -1, 0};
    for (uint32_t d = 2; d < kRecipMax; ++d) {
      uint32_t L = 0;
      while ((2ull << L) <= d) ++L;                       // floor(log2 d)
      if ((d & (d - 1)) == 0) { t[d] = Recip{1ull << (64 - L), -1, 0}; continue; }
      const unsigned __int128 two_s = (unsigned __int128)1 << (64 + L);
      const uint64_t m_dn = (uint64_t)(two_s / d);
      const uint64_t e_dn = (uint64_t)(two_s - (unsigned __int128)m_dn * d);   // e'
      const uint64_t e_up = d - e_dn;                                           // e
      if (e_up <= (1ull << L)) t[d] = Recip{m_dn + 1, -1, L};
      else t[d] = Recip{m_dn, 0, L};
    }
    return t;
  }();
  return tab.data();
}

uint64_t bce_test_div_recip(uint64_t x, uint32_t d) {
  return x == ~0ull ? x / d : div_recip1(x + 1, recip_table()[d]);
}

// Same arithmetic as encode() (bce.cpp:520-529 + shift_out :655-661) on the pair (l, R = h - l + 1): the next
// R' = step * freq depends on R alone (one add, one high multiply, one shift, one multiply), l accumulates off
// the critical chain, and shift_out is l <<= 16, R <<= 16.  R == 0 stands for the full range (h - l = 2^64 - 1).
void RangeCoder::encode_run(const uint64_t *out, uint64_t begin, uint64_t end) {
  const Recip *rt = recip_table();
  uint64_t l = l_, R = h_ - l_ + 1;
  uint16_t stage[64];                                   // output words, appended to data_ 64 at a time
  uint32_t ns = 0;
#define BCE_EMIT(v)                                                                                       \
  do {                                                                                                    \
    stage[ns++] = (uint16_t)(v);                                                                          \
    if (__builtin_expect(ns == 64, 0)) { data_.insert(data_.end(), stage, stage + 64); ns = 0; }          \
  } while (0)
#define BCE_STEP(cum, freq, total)                                                                        \
  do {                                                                                                    \
    const uint32_t t_ = (total);                                                                          \
    if (__builtin_expect(R - 1 < t_, 0)) {              /* :520-525 */                                    \
      for (int i_ = 0; i_ < 4; ++i_) BCE_EMIT(l >> (48 - 16 * i_));                                       \
      l = 0; R = 0;                                                                                       \
    }                                                                                                     \
    const uint64_t step_ =                                                                                \
        (__builtin_expect(R == 0, 0) || t_ >= kRecipMax) ? (R - 1) / t_ : div_recip1(R, rt[t_]);          \
    l += step_ * (cum);                                 /* :528 */                                        \
    R = step_ * (freq);                                 /* h = l + step*freq - 1  (:529) */               \
    while (__builtin_expect(!(((l + R - 1) ^ l) >> 48), 0)) {   /* shift_out :655-661 */                  \
      BCE_EMIT((l + R - 1) >> 48);                                                                        \
      l <<= 16;                                                                                           \
      R <<= 16;                                         /* ((r << 16) | 0xFFFF) + 1 */                    \
    }                                                                                                     \
  } while (0)
  // one uniform bit, set(s & 1, 2) of the k > 31 escape (bce.cpp:507-510): BCE_STEP(bit, 1, 2) with the division by two as a
  // shift and no multiplication at all -- step = (R - 1) >> 1 (R = 0, the full range: 2^63 - 1, as (2^64 - 1) / 2), l += step
  // or nothing, R = step.  3.7 % of text's symbols carry ~3 such bits each: a tenth of the coder's chain steps.
#define BCE_BIT(bit)                                                                                      \
  do {                                                                                                    \
    if (__builtin_expect(R - 1 < 2u, 0)) {              /* :541-546 */                                    \
      for (int i_ = 0; i_ < 4; ++i_) BCE_EMIT(l >> (48 - 16 * i_));                                       \
      l = 0; R = 0;                                                                                       \
    }                                                                                                     \
    const uint64_t step_ = (R - 1) >> 1;                                                                  \
    l += step_ & (0ull - (uint64_t)(bit));                                                                \
    R = step_;                                                                                            \
    while (__builtin_expect(!(((l + R - 1) ^ l) >> 48), 0)) {                                             \
      BCE_EMIT((l + R - 1) >> 48);                                                                        \
      l <<= 16;                                                                                           \
      R <<= 16;                                                                                           \
    }                                                                                                     \
  } while (0)
  for (uint64_t i = begin; i < end; ++i) {
    const uint64_t o = out[i];
    uint32_t es = out_esc_sentinel(o);                  // escape bits below a sentinel bit; 1 = none
    if (__builtin_expect(es != 1u, 0))                  // k > 31 escape, bce.cpp:507-510: uniform bits, LSB first
      for (; es > 1u; es >>= 1) BCE_BIT(es & 1u);
    BCE_STEP(out_cum(o), out_freq(o), out_total(o));
  }
#undef BCE_BIT
#undef BCE_STEP
#undef BCE_EMIT
  data_.insert(data_.end(), stage, stage + ns);
  l_ = l; h_ = l + R - 1;
}

void HostCoder::consume(int p, const SymRun *runs, size_t nruns, const uint64_t *out) {
  RangeCoder &rc = plane[p];
  for (size_t r = 0; r < nruns; ++r) rc.encode_run(out, runs[r].start, runs[r].start + runs[r].count);
}

HostCoder::HostCoder() {
  // (std::thread may throw std::system_error: the threads started so far are stopped and joined before it goes on to
  //  the caller, bce_hip_create, which turns it into a status)
  try {
    for (int p = 0; p < 8; ++p) w_[p].th = std::thread([this, p]() { run(p); });
  } catch (...) {
    stop_threads();
    throw;
  }
}

void HostCoder::stop_threads() {
  for (int p = 0; p < 8; ++p) {
    { std::lock_guard<std::mutex> g(w_[p].mu); w_[p].stop = true; }
    w_[p].cv.notify_all();
  }
  for (int p = 0; p < 8; ++p) if (w_[p].th.joinable()) w_[p].th.join();
}

HostCoder::~HostCoder() { stop_threads(); }

void HostCoder::run(int p) {
  Worker &w = w_[p];
  for (;;) {
    CoderBatch *b = nullptr;
    {
      std::unique_lock<std::mutex> lk(w.mu);
      w.cv.wait(lk, [&] { return w.stop || !w.q.empty(); });
      if (w.q.empty()) return;          // stop requested and nothing left
      b = w.q.front();
      w.q.pop_front();
    }
    const auto t0 = std::chrono::steady_clock::now();
    // No exception leaves this thread (it would end in std::terminate, across the C ABI): a failed allocation of the
    // output vector marks the coder as failed; the batch still counts as done so that wait / drain never hang, and
    // the encode entry point reports BCE_HIP_E_NOMEM (failed()).
    try {
      if (b->wait_ready) b->wait_ready();
      if (!failed_.load(std::memory_order_relaxed) && ((plane_mask >> p) & 1u)) {
        consume(p, b->runs[p].data(), b->runs[p].size(), b->out);
        for (const SymRun &r : b->runs[p]) w.nsym += r.count;
      }
    } catch (...) {
      failed_.store(true);
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    {
      std::lock_guard<std::mutex> g(done_mu_);
      w.busy += dt;
      b->pending.fetch_sub(1);
      ++completed_;
    }
    done_cv_.notify_all();
  }
}

void HostCoder::submit(CoderBatch *b) {
  b->pending.store(8);
  { std::lock_guard<std::mutex> g(done_mu_); submitted_ += 8; }
  for (int p = 0; p < 8; ++p) {
    { std::lock_guard<std::mutex> g(w_[p].mu); w_[p].q.push_back(b); }
    w_[p].cv.notify_one();
  }
}

void HostCoder::wait(CoderBatch *b) {
  std::unique_lock<std::mutex> lk(done_mu_);
  done_cv_.wait(lk, [&] { return b->pending.load() == 0; });
}

void HostCoder::drain() {
  std::unique_lock<std::mutex> lk(done_mu_);
  done_cv_.wait(lk, [&] { return completed_ == submitted_; });
}

double HostCoder::busy_seconds() {
  std::lock_guard<std::mutex> g(done_mu_);
  double m = 0;
  for (int p = 0; p < 8; ++p) m = w_[p].busy > m ? w_[p].busy : m;
  return m;
}

// BCE::encode :1134-1150: flush the plane coders and code the header (n, offset, stream sizes).  The archive itself
// -- header length, header, the eight streams (:1152-1157) -- is only laid out when somebody asks for it (assemble):
// 23 MB at 10^8 B of text are then copied once, straight into the caller's buffer.
void HostCoder::finish(const uint8_t config[9][32], uint32_t n, uint32_t offset) {
  if (getenv("BCE_HIP_CODER_DEBUG"))
    for (int p = 0; p < 8; ++p) fprintf(stderr, "coder %d: busy %.1f ms, %llu symbols, %zu words\n", p, w_[p].busy * 1e3, (unsigned long long)w_[p].nsym, plane[p].data().size());
  for (int i = 0; i < 8; ++i) plane[i].flush();         // :1134-1138
  rebuild_header(config, n, offset);
}

void HostCoder::rebuild_header(const uint8_t config[9][32], uint32_t n, uint32_t offset) {
  unsigned size = 0u;
  for (int i = 0; i < 8; ++i) size += (unsigned)plane[i].data().size();
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
  header_ = mainc.data();
}

size_t HostCoder::archive_words() const {
  size_t w = 1 + header_.size();
  for (int i = 0; i < 8; ++i) w += plane[i].data().size();
  return w;
}

void HostCoder::assemble(uint16_t *dst) const {          // :1152-1157
  *dst++ = (uint16_t)header_.size();
  memcpy(dst, header_.data(), header_.size() * 2);
  dst += header_.size();
  // the streams are tens of MB: copy them with a few threads
  uint16_t *at[8];
  for (int i = 0; i < 8; ++i) { at[i] = dst; dst += plane[i].data().size(); }
  size_t total = 0;
  for (int i = 0; i < 8; ++i) total += plane[i].data().size();
  if (total < (1u << 20)) {
    for (int i = 0; i < 8; ++i) memcpy(at[i], plane[i].data().data(), plane[i].data().size() * 2);
    return;
  }
  // (a thread that cannot be started: the threads running are joined and this one copies the rest -- never a
  //  joinable std::thread destroyed, never an exception out of here)
  std::thread th[8];
  int started = 0;
  try {
    for (; started < 8; ++started) {
      const int i = started;
      th[i] = std::thread([this, i, &at] { memcpy(at[i], plane[i].data().data(), plane[i].data().size() * 2); });
    }
  } catch (...) {
  }
  for (int i = started; i < 8; ++i) memcpy(at[i], plane[i].data().data(), plane[i].data().size() * 2);
  for (int i = 0; i < started; ++i) th[i].join();
}

void HostCoder::finish(const uint8_t config[9][32], uint32_t n, uint32_t offset, std::vector<uint16_t> &archive) {
  finish(config, n, offset);
  archive.resize(archive_words());
  assemble(archive.data());
}

}  // namespace bce
