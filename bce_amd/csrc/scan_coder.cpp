// scan_coder.cpp -- see scan_coder.h.  Plain C++ (no HIP).
#include "scan_coder.h"

#include <sched.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <thread>

namespace bce {

namespace {
// key of stat_[k]: the 8-bit quantised c2 | c1 (:743, uint32 wrap kept)
inline uint32_t scan_key(uint32_t c1, uint32_t c2, uint32_t cs) { return (((uint32_t)(c2 << 8) / cs) << 16) | ((uint32_t)(c1 << 8) / cs); }

template <class F> void run_tasks(unsigned threads, size_t ntasks, F &&fn) {
  std::atomic<size_t> next{0};
  auto work = [&] { for (size_t t; (t = next.fetch_add(1)) < ntasks;) fn(t); };
  std::vector<std::thread> th;
  const unsigned extra = (unsigned)std::min<size_t>(threads > 0 ? threads - 1 : 0, ntasks > 0 ? ntasks - 1 : 0);
  for (unsigned i = 0; i < extra; ++i) th.emplace_back(work);
  work();
  for (auto &t : th) t.join();
}
}  // namespace

void ScanCoder::set(uint32_t s, uint32_t k, uint32_t c1, uint32_t c2, uint32_t cs) {
  // k > 31: one uniform bit is charged and the range halves -- with ScanCoder's OWN formula
  // (k >> 1) + (~s & 1), which differs from AdaptiveCoder's (k + (~s & 1)) >> 1 for even k (quirk Q2)
  while (k > 31u) {
    z_ += std::log(2);
    const uint32_t s0 = s;
    s = s0 >> 1;
    k = (k >> 1) + ((~s0) & 1u);
  }
  stat_[k][scan_key(c1, c2, cs)].push_back((uint8_t)s);
}

void ScanCoder::set_packed(uint32_t word, int cls) {
  const uint32_t k = (word >> 5) & 31u;
  if (cls == 0) nesc_ += word >> 26;
  if (class_of(k) != cls) return;
  // the map decides the ORDER (of its iteration, later); finding a key's vector again goes through a flat table of the
  // 2^16 possible keys (both quotients are < 256)
  const uint32_t q1 = (word >> 10) & 0xFFu, q2 = (word >> 18) & 0xFFu;
  std::vector<std::vector<uint8_t> *> &fast = fast_[k];
  if (fast.empty()) fast.assign(65536, nullptr);
  std::vector<uint8_t> *&v = fast[(q2 << 8) | q1];
  if (!v) v = &stat_[k][(q2 << 16) | q1];                     // the reference's key: (c2 << 8) / cs << 16 | (c1 << 8) / cs  (:743)
  v->push_back((uint8_t)(word & 31u));
}

uint64_t ScanCoder::symbols(uint32_t k) const {
  uint64_t t = 0;
  for (auto &g : stat_[k]) t += g.second.size();
  return t;
}

double ScanCoder::base_cost(uint32_t k) const {
  double z_min = 0;
  for (auto &g : stat_[k]) z_min += std::log(k) * g.second.size();            // cost with no model at all
  return z_min;
}

double ScanCoder::trial_cost(uint32_t k, uint32_t j) const {
  std::vector<uint16_t> ctr((size_t)k << (2 * j), 0);
  // log(l / c) has few distinct arguments: l = k + sum of k counters <= 255 k, c = 1 + counter <= 255.  For small k (where
  // nearly all symbols are) the values are kept in a table filled on first use with the very expression the reference
  // evaluates -- the same doubles, a load instead of a division and a libm call.
  const bool tabled = k <= 4;
  std::vector<double> tab;
  if (tabled) tab.assign((size_t)(255 * k + 1) * 256, -1.0);     // (a cost is never negative: l >= c)
  // larger k: a direct-mapped cache of the same values (the full table would be 16 MB at k = 31, a handful of (l, c)
  // pairs are hot); an entry holds the pair it was computed for
  struct Ent { uint32_t key; double v; };
  std::vector<Ent> cache;
  if (!tabled) cache.assign(1u << 15, Ent{0u, 0.0});
  double z = 0;
  for (auto &g : stat_[k]) {
    uint16_t q1 = (uint16_t)(g.first >> 0), q2 = (uint16_t)(g.first >> 16);   // 8-bit quantised c1, c2
    q1 >>= 8 - j;
    q2 >>= 8 - j;
    uint16_t *ctx = &ctr[(size_t)((q1 << j) | q2) * k];
    for (uint8_t sym : g.second) {
      uint32_t l = k;
      for (uint32_t i = 0; i < k; ++i) l += ctx[i];
      if (tabled) {
        double &t = tab[(size_t)l * 256 + (1u + ctx[sym])];
        if (t < 0) t = std::log(static_cast<double>(l) / (1 + ctx[sym]));
        z += t;
      } else {
        const uint32_t cc = 1u + ctx[sym], key = (l << 8) | cc;      // (l <= 255 k < 2^13, c <= 255: never 0)
        Ent &e = cache[(key * 2654435761u) >> 17];
        if (e.key != key) { e.key = key; e.v = std::log(static_cast<double>(l) / (1 + ctx[sym])); }
        z += e.v;
      }
      if (++ctx[sym] == 0xFF)
        for (uint32_t i = 0; i < k; ++i) ctx[i] >>= 1;
    }
  }
  return z;
}

double ScanCoder::finish(uint8_t init[9][32], const double base[32], const double trial[32][6]) {
  for (uint64_t e = 0; e < nesc_; ++e) z_ += std::log(2);    // the escapes of set_class, added one by one as set() does
  nesc_ = 0;
  for (uint32_t k = 2; k < 31u; ++k) {                       // k = 31 is never optimised (:754)
    double z_min = base[k];
    for (uint32_t j = 0; j <= 5; ++j)                        // candidate context bits
      if (trial[k][j] < z_min) { z_min = trial[k][j]; init[i_][k] = (uint8_t)j; }
    z_ += z_min;
  }
  return z_ / std::log(256);
}

double ScanCoder::flush(uint8_t init[9][32]) {
  double base[32] = {0}, trial[32][6] = {{0}};
  for (uint32_t k = 2; k < 31u; ++k) {
    base[k] = base_cost(k);
    for (uint32_t j = 0; j <= 5; ++j) trial[k][j] = trial_cost(k, j);
  }
  return finish(init, base, trial);
}

ScanSet::ScanSet(unsigned threads) : threads_(threads) {
  for (int i = 0; i < 8; ++i) coders_.emplace_back(i);
  coders_.emplace_back(-1);
  if (const char *e = getenv("BCE_HIP_SCAN_THREADS")) threads_ = (unsigned)atoi(e);
  if (threads_ == 0) {
    cpu_set_t set;
    CPU_ZERO(&set);
    threads_ = sched_getaffinity(0, sizeof set, &set) == 0 ? (unsigned)CPU_COUNT(&set) : std::thread::hardware_concurrency();
    threads_ = std::max(1u, std::min(threads_, 64u));
  }
}

void ScanSet::consume(const uint32_t *records, const std::vector<ScanSpan> spans[8]) {
  // task = (plane, class of k): every map stat_[k] is filled by exactly one thread, in stream order
  run_tasks(threads_, 8 * ScanCoder::kClasses, [&](size_t t) {
    const int p = (int)(t / ScanCoder::kClasses), cls = (int)(t % ScanCoder::kClasses);
    ScanCoder &c = coders_[p];
    for (const ScanSpan &e : spans[p])
      for (uint64_t i = e.start; i < e.start + e.count; ++i) c.set_packed(records[i], cls);
  });
}

void ScanSet::flush(uint8_t init[9][32], double result_bytes[9]) {
  struct Task { uint8_t i, k, j; uint64_t w; };             // j = 6: the base cost
  std::vector<Task> tasks;
  for (int i = 0; i < 9; ++i)
    for (uint32_t k = 2; k < 31u; ++k) {
      const uint64_t w = coders_[i].symbols(k);
      if (!w) continue;                                       // (all costs of an empty k are 0.0, as in the reference)
      for (uint32_t j = 0; j <= 6; ++j) tasks.push_back(Task{(uint8_t)i, (uint8_t)k, (uint8_t)j, j == 6 ? w / 16 + 1 : w * (k + 8)});
    }
  std::stable_sort(tasks.begin(), tasks.end(), [](const Task &a, const Task &b) { return a.w > b.w; });   // longest first
  std::vector<std::array<double, 32>> vb(9);
  std::vector<std::array<std::array<double, 6>, 32>> vt(9);
  for (auto &a : vb) a.fill(0.0);
  for (auto &a : vt) for (auto &b : a) b.fill(0.0);
  run_tasks(threads_, tasks.size(), [&](size_t t) {
    const Task &q = tasks[t];
    if (q.j == 6) vb[q.i][q.k] = coders_[q.i].base_cost(q.k);
    else vt[q.i][q.k][q.j] = coders_[q.i].trial_cost(q.k, q.j);
  });
  for (int i = 0; i < 9; ++i) {                               // coder_[i].flush() in order, then main(-1): :1135-1149
    double b[32], tr[32][6];
    for (int k = 0; k < 32; ++k) { b[k] = vb[i][k]; for (int j = 0; j < 6; ++j) tr[k][j] = vt[i][k][j]; }
    const double r = coders_[i].finish(init, b, tr);
    if (result_bytes) result_bytes[i] = r;
  }
}

}  // namespace bce
