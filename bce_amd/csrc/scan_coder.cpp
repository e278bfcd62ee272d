// scan_coder.cpp -- see scan_coder.h.  Plain C++ (no HIP).
#include "scan_coder.h"

#include <sched.h>
#include <sys/mman.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <exception>
#include <mutex>
#include <new>
#include <thread>

namespace bce {

namespace {
// key of stat_[k]: the 8-bit quantised c2 | c1 (:743, uint32 wrap kept)
inline uint32_t scan_key(uint32_t c1, uint32_t c2, uint32_t cs) { return (((uint32_t)(c2 << 8) / cs) << 16) | ((uint32_t)(c1 << 8) / cs); }

template <class F> void run_tasks(unsigned threads, size_t ntasks, F &&fn) {
  std::atomic<size_t> next{0};
  std::exception_ptr failed;                                  // the first exception of any worker (std::bad_alloc), rethrown by the caller
  std::mutex mu;
  auto work = [&] {
    try {
      for (size_t t; (t = next.fetch_add(1)) < ntasks;) fn(t);
    } catch (...) {
      next.store(ntasks);
      std::lock_guard<std::mutex> lk(mu);
      if (!failed) failed = std::current_exception();
    }
  };
  std::vector<std::thread> th;
  const unsigned extra = (unsigned)std::min<size_t>(threads > 0 ? threads - 1 : 0, ntasks > 0 ? ntasks - 1 : 0);
  for (unsigned i = 0; i < extra; ++i) th.emplace_back(work);
  work();
  for (auto &t : th) t.join();
  if (failed) std::rethrow_exception(failed);
}
}  // namespace

uint8_t *ScanArena::take(size_t bytes) {
  bytes = (bytes + 63) & ~(size_t)63;
  std::lock_guard<std::mutex> lk(mu_);
  if (regions_.empty() || regions_.back().used + bytes > regions_.back().cap) {
    constexpr size_t kRegion = (size_t)256 << 20;             // address space only: pages come when touched
    const size_t cap = std::max(kRegion, (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1));
    void *p = mmap(nullptr, cap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) throw std::bad_alloc();
    (void)madvise(p, cap, MADV_HUGEPAGE);                     // 2 MB pages where the kernel allows: 512 x fewer faults
    regions_.push_back(Region{static_cast<uint8_t *>(p), cap, 0});
  }
  Region &r = regions_.back();
  uint8_t *out = r.p + r.used;
  r.used += bytes;
  return out;
}

void ScanArena::release() {
  std::lock_guard<std::mutex> lk(mu_);
  for (Region &r : regions_) (void)munmap(r.p, r.cap);
  regions_.clear();
}

void ScanCoder::learn(PerK &t, uint32_t key16) {
  if (!t.seen) {
    t.seen = arena_->take(65536);                             // (zero pages)
    ScanBump *bump = new (arena_->take(sizeof(ScanBump))) ScanBump{arena_.get()};
    t.order = new (arena_->take(sizeof(PerK::Order))) PerK::Order(ScanBumpAlloc<std::pair<const uint32_t, uint32_t>>(bump));
  }
  if (t.seen[key16]) return;
  t.seen[key16] = 1;
  // the reference's key: (c2 << 8) / cs << 16 | (c1 << 8) / cs  (:743)
  t.order->emplace(((key16 >> 8) << 16) | (key16 & 0xFFu), 0u);
}

void ScanCoder::put(uint32_t k, uint32_t q1, uint32_t q2, uint32_t sym) {
  PerK &t = stat_[k];
  const uint32_t key16 = (q2 << 8) | q1;
  learn(t, key16);
  Open &o = t.own;
  if (o.fill == kChunk) { o.chunk = arena_->take((size_t)kChunk * 3); o.fill = 0; }
  reinterpret_cast<uint16_t *>(o.chunk)[o.fill] = (uint16_t)key16;
  o.chunk[2 * (size_t)kChunk + o.fill] = (uint8_t)sym;
  if (!t.segs.empty() && t.segs.back().chunk == o.chunk && t.segs.back().off + t.segs.back().count == o.fill) ++t.segs.back().count;
  else t.segs.push_back(Seg{o.chunk, o.fill, 1});
  ++o.fill;
  ++t.n;
  t.ready = false;
}

void ScanCoder::set(uint32_t s, uint32_t k, uint32_t c1, uint32_t c2, uint32_t cs) {
  // k > 31: one uniform bit is charged and the range halves -- with ScanCoder's OWN formula
  // (k >> 1) + (~s & 1), which differs from AdaptiveCoder's (k + (~s & 1)) >> 1 for even k (quirk Q2)
  while (k > 31u) {
    z_ += std::log(2);
    const uint32_t s0 = s;
    s = s0 >> 1;
    k = (k >> 1) + ((~s0) & 1u);
  }
  const uint32_t key = scan_key(c1, c2, cs);                  // both quotients < 256 (c < cs)
  put(k, key & 0xFFFFu, key >> 16, s);
}

uint64_t ScanCoder::symbols(uint32_t k) const { return stat_[k].n; }

void ScanCoder::prepare(uint32_t k) const {
  PerK &t = stat_[k];
  if (t.ready) return;
  const size_t ng = t.order ? t.order->size() : 0, n = t.n;
  std::vector<uint32_t> pos_of(65536, 0);                     // key16 -> place in the iteration order
  t.gkey.resize(ng);
  size_t at = 0;
  if (t.order)
    for (auto &g : *t.order) { t.gkey[at] = g.first; pos_of[((g.first >> 16) << 8) | (g.first & 0xFFu)] = (uint32_t)at; ++at; }
  std::vector<uint64_t> cnt(ng + 1, 0);
  auto each = [&](auto &&fn) {                                 // fn(key16, symbol) over the stream, in order
    for (const Seg &sg : t.segs) {
      const uint16_t *key = reinterpret_cast<const uint16_t *>(sg.chunk) + sg.off;
      const uint8_t *sym = sg.chunk + 2 * (size_t)kChunk + sg.off;
      for (uint32_t i = 0; i < sg.count; ++i) fn(key[i], sym[i]);
    }
  };
  each([&](uint16_t g, uint8_t) { ++cnt[pos_of[g] + 1]; });
  for (size_t g = 0; g < ng; ++g) cnt[g + 1] += cnt[g];      // cnt[g] = where key g (in iteration order) starts
  t.gend.assign(cnt.begin() + 1, cnt.end());
  t.sorted = n ? arena_->take(n) : nullptr;
  each([&](uint16_t g, uint8_t s) { t.sorted[cnt[pos_of[g]]++] = s; });
  t.ready = true;
}

double ScanCoder::base_cost(uint32_t k) const {
  prepare(k);
  const PerK &t = stat_[k];
  double z_min = 0;
  uint64_t lo = 0;
  for (size_t g = 0; g < t.gkey.size(); ++g) { z_min += std::log(k) * (t.gend[g] - lo); lo = t.gend[g]; }   // cost with no model at all
  return z_min;
}

double ScanCoder::trial_cost(uint32_t k, uint32_t j) const {
  prepare(k);
  const PerK &t = stat_[k];
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
  uint64_t at = 0;
  for (size_t g = 0; g < t.gkey.size(); ++g) {
    uint16_t q1 = (uint16_t)(t.gkey[g] >> 0), q2 = (uint16_t)(t.gkey[g] >> 16);   // 8-bit quantised c1, c2
    q1 >>= 8 - j;
    q2 >>= 8 - j;
    uint16_t *ctx = &ctr[(size_t)((q1 << j) | q2) * k];
    for (; at < t.gend[g]; ++at) {
      const uint8_t sym = t.sorted[at];
      uint32_t l = k;
      for (uint32_t i = 0; i < k; ++i) l += ctx[i];
      if (tabled) {
        double &e = tab[(size_t)l * 256 + (1u + ctx[sym])];
        if (e < 0) e = std::log(static_cast<double>(l) / (1 + ctx[sym]));
        z += e;
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

// The six trials of one k in ONE pass over the recorded symbols: six counter tables, six accumulators -- each z_j adds up the
// very terms trial_cost(k, j) adds, in the same order (a symbol's term for j depends on table j alone), so the doubles
// are the same; the symbols, the map and the log table are walked once instead of six times.
void ScanCoder::trial_costs(uint32_t k, double out[6]) const {
  prepare(k);
  const PerK &t = stat_[k];
  std::vector<uint16_t> ctr[6];
  for (uint32_t j = 0; j <= 5; ++j) ctr[j].assign((size_t)k << (2 * j), 0);
  const bool tabled = k <= 4;
  std::vector<double> tab;
  if (tabled) tab.assign((size_t)(255 * k + 1) * 256, -1.0);
  struct Ent { uint32_t key; double v; };
  std::vector<Ent> cache;
  if (!tabled) cache.assign(1u << 15, Ent{0u, 0.0});
  auto lg = [&](uint32_t l, uint32_t c) -> double {           // log(l / c) as the reference evaluates it, remembered
    if (tabled) {
      double &t = tab[(size_t)l * 256 + c];
      if (t < 0) t = std::log(static_cast<double>(l) / c);
      return t;
    }
    const uint32_t key = (l << 8) | c;
    Ent &e = cache[(key * 2654435761u) >> 17];
    if (e.key != key) { e.key = key; e.v = std::log(static_cast<double>(l) / c); }
    return e.v;
  };
  double z[6] = {0, 0, 0, 0, 0, 0};
  uint64_t at = 0;
  for (size_t g = 0; g < t.gkey.size(); ++g) {
    uint16_t *ctx[6];
    for (uint32_t j = 0; j <= 5; ++j) {
      uint16_t q1 = (uint16_t)(t.gkey[g] >> 0), q2 = (uint16_t)(t.gkey[g] >> 16);
      q1 >>= 8 - j;
      q2 >>= 8 - j;
      ctx[j] = &ctr[j][(size_t)((q1 << j) | q2) * k];
    }
    for (; at < t.gend[g]; ++at) {
      const uint8_t sym = t.sorted[at];
#pragma GCC unroll 6
      for (uint32_t j = 0; j <= 5; ++j) {
        uint16_t *c = ctx[j];
        uint32_t l = k;
        for (uint32_t i = 0; i < k; ++i) l += c[i];
        z[j] += lg(l, 1u + c[sym]);
        if (++c[sym] == 0xFF)
          for (uint32_t i = 0; i < k; ++i) c[i] >>= 1;
      }
    }
  }
  for (uint32_t j = 0; j <= 5; ++j) out[j] = z[j];
}

// z after `z += c` executed m times (c > 0, z >= 0), the same double the loop gives, in ~one step per binade instead of m.
// While z and z + c stay in one binade [2^(e-1), 2^e) every z is a multiple of that binade's ulp, so fl(z + c) - z is one
// fixed multiple d of the ulp (c rounded to the ulp: the rounding sees only c's remainder, the same at every step) -- unless
// c lies exactly half way between two multiples (ties go by the parity of z: those steps are taken one by one).  The steps
// that stay below 2^e are therefore one exact multiplication; the step that crosses into the next binade is a real addition.
double scan_add_repeated(double z, double c, uint64_t m) {
  while (m) {
    const double z1 = z + c;
    int e0 = 0, e1 = 0;
    (void)std::frexp(z, &e0);
    (void)std::frexp(z1, &e1);
    if (!(z > 0) || e0 != e1) { z = z1; --m; continue; }
    const double d = z1 - z;                                  // exact (same binade)
    const double ulp = std::ldexp(1.0, e0 - 53);
    if (!(d > 0) || std::fabs(c - d) * 2 >= ulp) { z = z1; --m; continue; }   // a tie (or c below half an ulp): single steps
    const double top = std::ldexp(1.0, e0);
    const uint64_t Z = (uint64_t)((top - z) / ulp), D = (uint64_t)(d / ulp);  // exact integers < 2^53
    uint64_t s = D ? (Z - 1) / D : 0;                         // the most steps with z + s d < top
    if (s == 0) { z = z1; --m; continue; }
    if (s > m) s = m;
    z += (double)s * d;                                       // exact: s D < 2^53, the sum a multiple of ulp below top
    m -= s;
  }
  return z;
}

double ScanCoder::finish(uint8_t init[9][32], const double base[32], const double trial[32][6]) {
  z_ = scan_add_repeated(z_, std::log(2), nesc_);            // the escapes (set() / ScanSet::consume), `z_ += log(2)` each (:739), in order
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

ScanSet::ScanSet(unsigned threads) : arena_(std::make_shared<ScanArena>()), threads_(threads) {
  for (int i = 0; i < 8; ++i) coders_.emplace_back(i, arena_);
  coders_.emplace_back(-1, arena_);
  if (const char *e = getenv("BCE_HIP_SCAN_THREADS")) threads_ = (unsigned)atoi(e);
  if (const char *e = getenv("BCE_HIP_SCAN_MIN_RANGE")) min_range_ = std::max(1, atoi(e));   // (tests: many ranges on small inputs)
  if (threads_ == 0) {
    cpu_set_t set;
    CPU_ZERO(&set);
    threads_ = sched_getaffinity(0, sizeof set, &set) == 0 ? (unsigned)CPU_COUNT(&set) : std::thread::hardware_concurrency();
    threads_ = std::max(1u, std::min(threads_, 64u));
    // a container's CPU quota (cgroup v2 cpu.max "quota period"): more runnable threads than the quota pays for are frozen
    // together for the rest of each period -- the thread feeding the GPU with them -- so the pool stays within it
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      long long quota = 0, period = 0;
      if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0)
        threads_ = std::max(1u, std::min(threads_, (unsigned)((quota + period - 1) / period)));
      fclose(f);
    }
  }
}

void ScanSet::consume(const uint32_t *records, const std::vector<ScanSpan> spans[8]) {
  // 1. Every plane's records are cut into ranges of about equal length; a task records one range, all k at once, into the
  //    chunks of its slot.  What must happen in stream order -- a key entering the map -- is only noted (the slot's
  //    `fresh` list: keys this slot meets for the first time).
  constexpr uint32_t kChunk = ScanCoder::kChunk;
  uint64_t total = 0, np[8];
  for (int p = 0; p < 8; ++p) { np[p] = 0; for (const ScanSpan &e : spans[p]) np[p] += e.count; total += np[p]; }
  if (!total) return;
  const uint64_t target = std::max<uint64_t>(min_range_, (total + 3 * threads_ - 1) / (3 * threads_));
  struct Task { uint8_t p, s; uint64_t lo, hi; };              // records [lo, hi) of plane p's spans laid end to end
  std::vector<Task> tasks;
  unsigned pieces[8];
  for (int p = 0; p < 8; ++p) {
    pieces[p] = np[p] ? (unsigned)std::min<uint64_t>(kSlots, (np[p] + target - 1) / target) : 0;
    for (unsigned s = 0; s < pieces[p]; ++s) tasks.push_back(Task{(uint8_t)p, (uint8_t)s, np[p] * s / pieces[p], np[p] * (s + 1) / pieces[p]});
  }
  if (slots_.empty()) slots_.resize(8 * kSlots);
  for (const Task &q : tasks)
    if (!slots_[q.p * kSlots + q.s]) slots_[q.p * kSlots + q.s].reset(new Slot());
  run_tasks(threads_, tasks.size(), [&](size_t ti) {
    const Task &q = tasks[ti];
    Slot &sl = *slots_[q.p * kSlots + q.s];
    uint32_t start[32];
    for (int k = 0; k < 32; ++k) start[k] = sl.open[k].fill;
    uint64_t nesc = 0, at = 0;
    for (const ScanSpan &e : spans[q.p]) {
      const uint64_t lo = std::max(at, q.lo), hi = std::min(at + e.count, q.hi);
      for (uint64_t i = lo; i < hi; ++i) {
        const uint32_t w = records[e.start + (i - at)];
        const uint32_t k = (w >> 5) & 31u, key16 = (w >> 10) & 0xFFFFu;     // [17:10] q1, [25:18] q2: q2 << 8 | q1
        nesc += w >> 26;
        ScanCoder::Open &o = sl.open[k];
        if (o.fill == kChunk) {
          if (o.chunk && start[k] < kChunk) sl.segs[k].push_back(ScanCoder::Seg{o.chunk, start[k], kChunk - start[k]});
          o.chunk = arena_->take((size_t)kChunk * 3);
          o.fill = 0;
          start[k] = 0;
        }
        reinterpret_cast<uint16_t *>(o.chunk)[o.fill] = (uint16_t)key16;
        o.chunk[2 * (size_t)kChunk + o.fill] = (uint8_t)(w & 31u);
        ++o.fill;
        uint8_t *&seen = sl.seen[k];
        if (!seen) seen = arena_->take(8192);
        if (!((seen[key16 >> 3] >> (key16 & 7u)) & 1u)) { seen[key16 >> 3] |= (uint8_t)(1u << (key16 & 7u)); sl.fresh[k].push_back((uint16_t)key16); }
      }
      at += e.count;
      if (at >= q.hi) break;
    }
    for (int k = 0; k < 32; ++k)
      if (sl.open[k].chunk && sl.open[k].fill > start[k] && start[k] < kChunk)
        sl.segs[k].push_back(ScanCoder::Seg{sl.open[k].chunk, start[k], sl.open[k].fill - start[k]});
    sl.nesc += nesc;
  });
  // 2. Per (plane, k), the slots in range order: their pieces join the stream, their fresh keys enter the map -- a key
  //    more than one slot met enters where the first of them met it, which is where the sequential set() would have.
  run_tasks(threads_, 8 * 32, [&](size_t ti) {
    const unsigned p = (unsigned)(ti / 32), k = (unsigned)(ti % 32);
    ScanCoder &c = coders_[p];
    ScanCoder::PerK &t = c.stat_[k];
    for (unsigned s = 0; s < pieces[p]; ++s) {
      Slot &sl = *slots_[p * kSlots + s];
      for (const ScanCoder::Seg &sg : sl.segs[k]) { t.segs.push_back(sg); t.n += sg.count; t.ready = false; }
      sl.segs[k].clear();
      for (uint16_t key16 : sl.fresh[k]) c.learn(t, key16);
      sl.fresh[k].clear();
    }
  });
  for (int p = 0; p < 8; ++p)
    for (unsigned s = 0; s < pieces[p]; ++s) { Slot &sl = *slots_[p * kSlots + s]; coders_[p].nesc_ += sl.nesc; sl.nesc = 0; }
}

void ScanSet::release() {
  for (ScanCoder &c : coders_) for (auto &t : c.stat_) t = ScanCoder::PerK();
  slots_.clear();
  arena_->release();
}

void ScanSet::flush(uint8_t init[9][32], double result_bytes[9]) {
  struct Task { uint8_t i, k; uint64_t w; };                // one task per (coder, k): group the stream, base cost, the six trials
  std::vector<Task> tasks;
  for (int i = 0; i < 9; ++i)
    for (uint32_t k = 2; k < 31u; ++k) {
      const uint64_t w = coders_[i].symbols(k);
      if (!w) continue;                                       // (all costs of an empty k are 0.0, as in the reference)
      tasks.push_back(Task{(uint8_t)i, (uint8_t)k, w * (k + 8)});
    }
  std::stable_sort(tasks.begin(), tasks.end(), [](const Task &a, const Task &b) { return a.w > b.w; });   // longest first
  std::vector<std::array<double, 32>> vb(9);
  std::vector<std::array<std::array<double, 6>, 32>> vt(9);
  for (auto &a : vb) a.fill(0.0);
  for (auto &a : vt) for (auto &b : a) b.fill(0.0);
  std::vector<double> took(tasks.size(), 0.0), cpu(tasks.size(), 0.0);
  const bool dbg = getenv("BCE_HIP_SCAN_DEBUG") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  auto tcpu = [] { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; };
  const double t_start = dbg ? now() : 0.0;
  run_tasks(threads_, tasks.size(), [&](size_t t) {
    const Task &q = tasks[t];
    const double t0 = dbg ? now() : 0.0, c0 = dbg ? tcpu() : 0.0;
    vb[q.i][q.k] = coders_[q.i].base_cost(q.k);
    coders_[q.i].trial_costs(q.k, vt[q.i][q.k].data());
    if (dbg) { took[t] = now() - t0; cpu[t] = tcpu() - c0; }
  });
  const double t_tasks = dbg ? now() : 0.0;
  if (dbg) {
    double sum = 0, csum = 0;
    for (double d : took) sum += d;
    for (double d : cpu) csum += d;
    fprintf(stderr, "scan flush: %zu tasks ran %.3f s wall, %.3f s summed wall, %.3f s of thread CPU; longest:", tasks.size(), t_tasks - t_start, sum, csum);
    for (size_t t = 0; t < tasks.size() && t < 6; ++t)
      fprintf(stderr, " (plane %d k %d: %llu symbols) %.3f s", tasks[t].i, tasks[t].k,
              (unsigned long long)coders_[tasks[t].i].symbols(tasks[t].k), took[t]);
    fprintf(stderr, "\n");
  }
  for (int i = 0; i < 9; ++i) {                               // coder_[i].flush() in order, then main(-1): :1135-1149
    double b[32], tr[32][6];
    for (int k = 0; k < 32; ++k) { b[k] = vb[i][k]; for (int j = 0; j < 6; ++j) tr[k][j] = vt[i][k][j]; }
    const double r = coders_[i].finish(init, b, tr);
    if (result_bytes) result_bytes[i] = r;
  }
}

}  // namespace bce
