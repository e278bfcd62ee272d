// k4_model.hip -- K4: AdaptiveCoder probability estimation on the GPU.  Replaces the model half of
// AdaptiveCoder<31>::set(s,k,c1,c2,cs) (bce.cpp:506-518,529,531-533) and get_context (bce.cpp:671-677);
// the range-coder half stays on the host (host_coder.cpp).
//
// The model is sequential PER CONTEXT SLOT (k <= 31 byte counters: increment, halve all when one hits
// 0xFF) and independent across slots.  A flush takes the symbol key words K3 emitted since the last
// flush, in stream order (round, plane, s):
//   1. stable LSD radix sort of (key word, record index) on key bits 10..28 = plane|slot (3 passes):
//      every slot's records become one contiguous run, still in stream order; sym and k ride along in
//      the low key bits, so the replay needs no gather.
//   2. replay, 64 events (one wave) at a time.  Between two halvings a counter is
//      base + (number of earlier events with that symbol), so for a 64-event chunk
//        freq  = C[s] + #{earlier lanes with the same symbol}            (+1)
//        cum   = sum_{i<s} C[i] + #{earlier lanes with a smaller symbol} (+s)
//        total = sum_i C[i] + lane                                       (+k)
//      with C held one counter per lane (lane i = counter i), the two counts taken from 5 ballots
//      (bit-sliced less-than / equal masks) and the prefix of C from a 5-step wave scan.  The first
//      lane whose counter reaches 0xFF ends the chunk: events up to it are committed, all counters are
//      halved, and the next chunk starts right after it (k4_replay).
//      The only truly sequential thing in a slot is WHERE its halvings fall, so long runs (>= 256 events)
//      are split in three kernels over the 64-aligned windows of the sorted array:
//        A  k4_window_kernel (all windows in parallel): per-window symbol histogram of the window's
//           leading run fragment; short runs are replayed on the spot; long runs go to a queue.
//        B  k4_long_kernel (one wave per long run): walks the run's windows reading only the 32-byte
//           histograms (prefetched); a window with no counter crossing 0xFF costs ~10 instructions, a
//           crossing one an exact search inside the window; records the counter state at every window
//           start.
//        C  k4_emit_kernel (all windows in parallel): replays each long-run window from its recorded
//           state and writes the outputs.
//   3. (cum, freq, total) go back to the record's ORIGINAL index; the host range coders read them in
//      stream order.
// Counter state persists in HBM between flushes, so flushes can be arbitrarily small.
#include <algorithm>

#include <stdlib.h>
#include <sys/mman.h>

#include "common.h"
#include "scan_util.h"

namespace bce {

constexpr int K4_T = 256;
constexpr uint32_t K4_LONG = 256;        // runs at least this long take the A/B/C route
constexpr uint8_t K4_NOT_LONG = 0xFE;    // haltW: window's leading fragment is not part of a queued run
constexpr uint8_t K4_NO_HALVE = 0xFF;    // haltW: long-run window without a halving

struct K4Args {
  const uint32_t *keys, *vals;   // sorted
  uint8_t *stat;
  const PlaneCfg *cfg;
  uint64_t *out;
  const uint32_t *esc;           // escape words in original record order (K3 wrote them)
  uint8_t *histT;                // [nwin][32] symbol counts of each window's leading run fragment
  uint8_t *stateW;               // [nwin][32] counters at the window start (long runs)
  uint32_t *winfo;               // [nwin] runid<<7 | length of the leading fragment
  uint8_t *haltW;                // [nwin]
  unsigned long long *bitsW;     // [nwin] bit i = low symbol bit of record i of the leading fragment (k = 2 runs)
  uint2 *queue;                  // long runs: {start, key word of the first record}
  uint32_t *qcount;
  uint32_t stat_off[8];
  uint32_t nsym;
};

__global__ __launch_bounds__(K4_T) void k4_iota_kernel(uint32_t nsym, uint32_t *__restrict__ vals) {
  for (uint64_t i = (uint64_t)blockIdx.x * K4_T + threadIdx.x; i < nsym; i += (uint64_t)gridDim.x * K4_T)
    vals[i] = (uint32_t)i;
}

__device__ __forceinline__ uint8_t *k4_counters(const K4Args &a, uint32_t hkey) {
  const uint32_t runid = hkey >> kSymRunShift, k = key_k(hkey), p = runid >> 16, slot = runid & 0xFFFFu;
  const PlaneCfg &cfg = a.cfg[p];
  return a.stat + a.stat_off[p] + cfg.off[k] + (slot - cfg.ctxoff[k]) * k;
}

// lanes with an equal / a smaller symbol among the lanes of `vm`, bit-sliced from the MSB down
__device__ __forceinline__ void k4_eq_lt(uint32_t s, uint64_t vm, uint64_t &eq, uint64_t &lt) {
  eq = vm; lt = 0;
#pragma unroll
  for (int b = 4; b >= 0; --b) {
    const bool bit = (s >> b) & 1u;
    const uint64_t Bb = __ballot(bit);
    if (bit) { lt |= eq & ~Bb; eq &= Bb; } else { eq &= ~Bb; }
  }
}

// lane i (< 32) <- number of lanes in `mask` whose symbol is i   (eq = per-lane equal-symbol mask)
__device__ __forceinline__ uint32_t k4_symbol_counts(uint32_t s, uint64_t eq, uint64_t mask, bool member,
                                                    volatile uint32_t *cb, uint32_t lane) {
  const uint64_t eqm = eq & mask;
  if (lane < 32) cb[lane] = 0;
  __builtin_amdgcn_wave_barrier();
  if (member && (eqm >> lane) == 1ull) cb[s] = (uint32_t)__popcll(eqm);   // highest member lane of its group
  __builtin_amdgcn_wave_barrier();
  const uint32_t r = lane < 32 ? cb[lane] : 0u;
  __builtin_amdgcn_wave_barrier();
  return r;
}

// Replay the events [pos, limit) of run `runid` (stops earlier if the run ends) with the whole wave:
// emits (cum,freq,total), updates the per-lane counters C.  Returns the position reached.
__device__ __forceinline__ uint64_t k4_replay(const K4Args &a, uint64_t pos, uint64_t limit, uint32_t runid,
                                              uint32_t k, uint32_t &C, volatile uint32_t *cb, uint32_t lane,
                                              uint64_t ltm) {
  if (limit > a.nsym) limit = a.nsym;
  // the escape word of a record is looked up (a random 4-byte read per record: 38 B fetched per record when every record did
  // it) only where there can be one: an escape halves k until it is <= 31 (pack_symbol), so what is left is >= 16
  const bool esc_possible = k >= 16u;
  uint64_t j = pos + lane;
  uint32_t kk = j < limit ? a.keys[j] : 0xFFFFFFFFu;
  for (;;) {
    const bool valid = (kk >> kSymRunShift) == runid;     // the run is contiguous: valid lanes form a prefix
    const uint64_t vm = __ballot(valid);
    if (!vm) break;
    const uint32_t idx = valid ? a.vals[j] : 0u;
    // speculative load of the next chunk (correct unless a halving cuts this one short)
    const uint64_t jn = j + 64;
    const uint32_t kn = jn < limit ? a.keys[jn] : 0xFFFFFFFFu;
    const uint32_t s = kk & 31u;
    // exclusive prefix of the counters over lanes 0..31 and their total
    uint32_t inc = C;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
      const uint32_t t = __shfl_up(inc, o);
      if (lane >= (uint32_t)o) inc += t;
    }
    const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)inc, 31);
    const uint32_t Ps = __shfl(inc - C, (int)s);
    const uint32_t Cs = __shfl(C, (int)s);
    uint64_t eq, lt;
    k4_eq_lt(s, vm, eq, lt);
    const uint32_t eqb = (uint32_t)__popcll(eq & ltm), ltb = (uint32_t)__popcll(lt & ltm);
    const uint32_t before = Cs + eqb;                       // counter value this event sees
    const uint64_t hm = __ballot(valid && before + 1u == 0xFFu);
    const uint32_t nc = hm ? (uint32_t)__ffsll((long long)hm) : (uint32_t)__popcll(vm);   // committed events
    const bool commit = lane < nc;
    if (commit) {
      const uint32_t cum = Ps + ltb + s, total = T + lane + k, freq = before + 1u;
      a.out[idx] = pack_model_out(cum, freq, total, esc_possible ? a.esc[idx] : 0u);
    }
    const uint64_t cm = nc >= 64 ? ~0ull : ((1ull << nc) - 1ull);
    C += k4_symbol_counts(s, eq, cm, commit, cb, lane);
    if (hm) C >>= 1;                                       // bce.cpp:531-533
    pos += nc;
    if (nc == 64) { j = jn; kk = kn; }
    else { j = pos + lane; kk = j < limit ? a.keys[j] : 0xFFFFFFFFu; }
  }
  return pos;
}

// A: every 64-record window of the sorted array
__global__ __launch_bounds__(K4_T) void k4_window_kernel(K4Args a) {
  __shared__ uint32_t cntbuf[K4_T / 64][32];
  const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  volatile uint32_t *cb = cntbuf[w];
  const uint64_t ltm = (1ull << lane) - 1ull;
  const uint64_t nwaves = (uint64_t)gridDim.x * (K4_T / 64);
  for (uint64_t win = (uint64_t)blockIdx.x * (K4_T / 64) + w; win * 64 < a.nsym; win += nwaves) {
    const uint64_t j0 = win * 64 + lane;
    const bool inb = j0 < a.nsym;
    const uint32_t key = inb ? a.keys[j0] : 0xFFFFFFFFu;
    const uint32_t prev = (j0 > 0 && inb) ? a.keys[j0 - 1] : 0xFFFFFFFFu;
    const uint32_t rid = key >> kSymRunShift;
    const bool head = inb && (j0 == 0 || rid != (prev >> kSymRunShift));
    uint64_t heads = __ballot(head);
    // histogram of the leading fragment (the lanes that continue the run of lane 0)
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rid);
    const bool lead = inb && rid == r0;
    const uint64_t leadm = __ballot(lead);
    const uint32_t s = key & 31u;
    uint64_t eq, lt;
    k4_eq_lt(s, leadm, eq, lt);
    const uint32_t h = k4_symbol_counts(s, eq, leadm, lead, cb, lane);
    if (lane < 32) a.histT[win * 32 + lane] = (uint8_t)h;
    const uint64_t b1 = __ballot(lead && (key & 1u));
    if (lane == 0) {
      a.winfo[win] = (r0 << 7) | (uint32_t)__popcll(leadm);
      a.haltW[win] = K4_NOT_LONG;
      a.bitsW[win] = b1;
    }
    while (heads) {                                         // runs that START in this window
      const int hl = __ffsll((long long)heads) - 1;
      heads &= heads - 1;
      const uint32_t hkey = (uint32_t)__builtin_amdgcn_readlane((int)key, hl);
      const uint64_t start = win * 64 + (uint64_t)hl;
      const uint64_t probe = start + K4_LONG - 1;
      const bool is_long = probe < a.nsym && (a.keys[probe] >> kSymRunShift) == (hkey >> kSymRunShift);
      if (is_long) {
        if (lane == 0) {
          const uint32_t q = atomicAdd(a.qcount, 1u);
          a.queue[q] = make_uint2((uint32_t)start, hkey);
        }
      } else {
        const uint32_t k = key_k(hkey);
        uint8_t *ctr = k4_counters(a, hkey);
        uint32_t C = lane < k ? (uint32_t)ctr[lane] : 0u;
        (void)k4_replay(a, start, ~0ull, hkey >> kSymRunShift, k, C, cb, lane, ltm);
        if (lane < k) ctr[lane] = (uint8_t)C;
      }
    }
  }
}

// B: one wave per long run.  The walk is serial in the counter state but its inputs (window info, window
// histogram, window keys) do not depend on that state, so they are streamed in groups of K4_GW windows:
// the next group's global loads are issued before the current group is processed out of LDS, which keeps
// HBM/L2 latency off the serial chain.
constexpr uint32_t K4_GW = 16;                      // windows per staged group
struct K4Stage {                                    // per wave
  uint32_t keys[K4_GW][64];
  uint32_t hist[K4_GW][8];                          // 32 bytes per window
  uint32_t info[K4_GW];
};

__global__ __launch_bounds__(K4_T) void k4_long_kernel(K4Args a) {
  __shared__ uint32_t cntbuf[K4_T / 64][32];
  __shared__ K4Stage stage[K4_T / 64];
  const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  volatile uint32_t *cb = cntbuf[w];
  volatile K4Stage *st = &stage[w];
  const uint64_t ltm = (1ull << lane) - 1ull;
  const uint32_t nq = *a.qcount;
  const uint32_t nwaves = gridDim.x * (K4_T / 64);
  const uint64_t nwin = ((uint64_t)a.nsym + 63) / 64;
  const uint32_t *hist32 = reinterpret_cast<const uint32_t *>(a.histT);
  for (uint32_t q = blockIdx.x * (K4_T / 64) + w; q < nq; q += nwaves) {
    const uint2 e = a.queue[q];
    const uint32_t hkey = e.y, runid = hkey >> kSymRunShift, k = key_k(hkey);
    uint8_t *ctr = k4_counters(a, hkey);
    uint32_t C = lane < k ? (uint32_t)ctr[lane] : 0u;
    // head fragment: from the run start to the end of its window, replayed directly
    uint64_t gbase = (uint64_t)e.x / 64 + 1;
    (void)k4_replay(a, e.x, gbase * 64, runid, k, C, cb, lane, ltm);
    if (k == 2) {
      // ---- binary slots (every really long run is one: 0.5-1.4 M records of a 16 M flush).  Lane = window, 64 windows
      // (4096 events) per step: with the inclusive prefix sums of the windows' zero / one counts, "the next window in
      // which a counter reaches 0xFF" is one ballot, so the serial work is per HALVING (one every 128-255 events: a
      // counter is <= 127 after one), not per window; the windows in between only add.  The exact event inside the
      // window is "the n-th zero or the n-th one" of its 64-bit mask (mbcnt + one ballot).  ~5x the window-by-window
      // walk this replaces (0.2 us per window, 4.5 ms for the longest run of a flush). ----
      uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)C, 0), c1 = (uint32_t)__builtin_amdgcn_readlane((int)C, 1);
      uint64_t gb = gbase;
      unsigned long long nbits = gb + lane < nwin ? a.bitsW[gb + lane] : 0ull;
      uint32_t ninfo = gb + lane < nwin ? a.winfo[gb + lane] : 0xFFFFFFFFu;
      for (;;) {
        const unsigned long long cbits = nbits;
        const uint32_t cinfo = ninfo;
        const uint64_t g2 = gb + 64;
        nbits = g2 + lane < nwin ? a.bitsW[g2 + lane] : 0ull;        // next group in flight
        ninfo = g2 + lane < nwin ? a.winfo[g2 + lane] : 0xFFFFFFFFu;
        const uint64_t vm = __ballot((cinfo >> 7) == runid);          // (also false beyond nwin: info = ~0)
        const uint32_t nv = ~vm ? (uint32_t)__ffsll((long long)~vm) - 1u : 64u;   // the run's windows of this group: lanes [0, nv)
        if (nv == 0) break;
        const bool inrun = lane < nv;
        const uint32_t f = inrun ? (cinfo & 127u) : 0u;               // 64, except in the window where the run ends
        const unsigned long long B = cbits & (f >= 64u ? ~0ull : ((1ull << f) - 1ull));
        const uint32_t h1 = (uint32_t)__popcll(B), h0 = f - h1;
        const uint32_t I0 = wave_incl_sum(h0), I1 = wave_incl_sum(h1), E0 = I0 - h0, E1 = I1 - h1;
        uint32_t o0 = 0, o1 = 0;                                      // zeros / ones of this group behind the base state (c0, c1)
        int last = -1;                                                // window of the latest halving
        uint32_t st = 0, halt = K4_NO_HALVE;
        for (;;) {
          // a counter is <= 127 right after a halving and a window holds 64 events: never two halvings in one window
          const bool cross = inrun && (int)lane > last && (c0 + I0 - o0 >= 0xFFu || c1 + I1 - o1 >= 0xFFu);
          const uint64_t m = __ballot(cross);
          const int i = m ? __ffsll((long long)m) - 1 : 64;
          if (inrun && (int)lane > last && (int)lane <= i) st = (c0 + E0 - o0) | ((c1 + E1 - o1) << 8);   // state at my window's start
          if (!m) break;
          const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)E0, i), e1 = (uint32_t)__builtin_amdgcn_readlane((int)E1, i);
          const uint32_t s0 = c0 + e0 - o0, s1 = c1 + e1 - o1, fi = (uint32_t)__builtin_amdgcn_readlane((int)f, i);
          const unsigned long long Bi = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(B >> 32), i) << 32) |
                                        (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)B, i);
          const uint32_t pb = __builtin_amdgcn_mbcnt_hi((uint32_t)(Bi >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)Bi, 0u));
          const bool bit = (Bi >> lane) & 1ull;
          const bool hit = lane < fi && (bit ? (s1 + pb + 1u == 0xFFu) : (s0 + (lane - pb) + 1u == 0xFFu));
          const uint64_t hm = __ballot(hit);                          // != 0: a counter crosses in this window
          const uint32_t t = (uint32_t)__ffsll((long long)hm) - 1u;
          const unsigned long long le = (2ull << t) - 1ull;
          const uint32_t cle1 = (uint32_t)__popcll(Bi & le), cle0 = t + 1u - cle1;
          c0 = (s0 + cle0) >> 1;                                      // halve (bce.cpp:531-533): the state right behind event t
          c1 = (s1 + cle1) >> 1;
          o0 = e0 + cle0; o1 = e1 + cle1;
          if ((int)lane == i) halt = t;
          last = i;
        }
        if (inrun) {
          *reinterpret_cast<uint16_t *>(a.stateW + (gb + lane) * 32) = (uint16_t)st;
          a.haltW[gb + lane] = (uint8_t)halt;
        }
        c0 += (uint32_t)__builtin_amdgcn_readlane((int)I0, (int)nv - 1) - o0;   // the state behind the group's last window
        c1 += (uint32_t)__builtin_amdgcn_readlane((int)I1, (int)nv - 1) - o1;
        if (nv < 64u || (uint32_t)__builtin_amdgcn_readlane((int)f, (int)nv - 1) < 64u) break;   // the run ends in this group
        gb += 64;
      }
      if (lane == 0) { ctr[0] = (uint8_t)c0; ctr[1] = (uint8_t)c1; }
      continue;
    }
    // group loads into registers: 16 keys, 2 histogram dwords, 1 info dword per lane
    uint32_t rk[K4_GW], rh[2], ri;
    auto load_group = [&](uint64_t g0) {
#pragma unroll
      for (uint32_t i = 0; i < K4_GW; ++i) {
        const uint64_t j = (g0 + i) * 64 + lane;
        rk[i] = j < a.nsym ? a.keys[j] : 0xFFFFFFFFu;
      }
#pragma unroll
      for (uint32_t r = 0; r < 2; ++r) {                 // K4_GW * 8 dwords = 128 dwords, two per lane
        const uint64_t d = g0 * 8 + r * 64 + lane;
        rh[r] = d < nwin * 8 ? hist32[d] : 0u;
      }
      ri = (lane < K4_GW && g0 + lane < nwin) ? a.winfo[g0 + lane] : 0xFFFFFFFFu;
    };
    load_group(gbase);
    bool running = true;
    while (running) {
      // registers -> LDS (the previous group has been fully consumed)
#pragma unroll
      for (uint32_t i = 0; i < K4_GW; ++i) st->keys[i][lane] = rk[i];
#pragma unroll
      for (uint32_t r = 0; r < 2; ++r) st->hist[(r * 64 + lane) >> 3][(r * 64 + lane) & 7u] = rh[r];
      if (lane < K4_GW) st->info[lane] = ri;
      __builtin_amdgcn_wave_barrier();
      load_group(gbase + K4_GW);                        // in flight while this group is processed
      for (uint32_t i = 0; i < K4_GW; ++i) {
        const uint64_t win = gbase + i;
        const uint32_t info = st->info[i];
        if (win >= nwin || (info >> 7) != runid) { running = false; break; }
        const uint32_t f = info & 127u;
        // byte `lane` of the window's 32-byte histogram
        const uint32_t h = lane < 32 ? (st->hist[i][lane >> 2] >> (8u * (lane & 3u))) & 0xFFu : 0u;
        if (lane < 32) a.stateW[win * 32 + lane] = (uint8_t)C;
        const uint64_t hm = __ballot(lane < k && C + h >= 0xFFu);
        if (!hm) {
          C += h;
          if (lane == 0) a.haltW[win] = K4_NO_HALVE;
        } else {
          // some counter reaches 0xFF inside this window: find the first event that does
          const bool valid = lane < f;
          const uint32_t s = st->keys[i][lane] & 31u;
          const uint64_t vm = __ballot(valid);
          uint64_t eq, lt;
          k4_eq_lt(s, vm, eq, lt);
          const uint32_t before = __shfl(C, (int)s) + (uint32_t)__popcll(eq & ltm);
          const uint64_t hit = __ballot(valid && before + 1u == 0xFFu);
          const uint32_t t = (uint32_t)__ffsll((long long)hit) - 1u;     // hit != 0 by construction
          const uint64_t cm = (2ull << t) - 1ull;                         // lanes <= t
          const uint32_t cle = k4_symbol_counts(s, eq, cm, valid && lane <= t, cb, lane);
          C = ((C + cle) >> 1) + (h - cle);                               // halve (bce.cpp:531-533), then the rest
          if (lane == 0) a.haltW[win] = (uint8_t)t;
        }
        if (f < 64) { running = false; break; }                           // the run ends inside this window
      }
      __builtin_amdgcn_wave_barrier();
      gbase += K4_GW;
    }
    if (lane < k) ctr[lane] = (uint8_t)C;
  }
}

// C: every window whose leading fragment belongs to a long run (beyond the run's first window)
__global__ __launch_bounds__(K4_T) void k4_emit_kernel(K4Args a) {
  __shared__ uint32_t cntbuf[K4_T / 64][32];
  const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  volatile uint32_t *cb = cntbuf[w];
  const uint64_t ltm = (1ull << lane) - 1ull;
  const uint64_t nwaves = (uint64_t)gridDim.x * (K4_T / 64);
  for (uint64_t win = (uint64_t)blockIdx.x * (K4_T / 64) + w; win * 64 < a.nsym; win += nwaves) {
    if (a.haltW[win] == K4_NOT_LONG) continue;
    const uint32_t info = a.winfo[win];
    const uint32_t hkey = a.keys[win * 64];
    const uint32_t k = key_k(hkey);
    uint32_t C = lane < k ? (uint32_t)a.stateW[win * 32 + lane] : 0u;    // (the k = 2 walk records two bytes only)
    (void)k4_replay(a, win * 64, win * 64 + (info & 127u), info >> 7, k, C, cb, lane, ltm);
  }
}

int k4_prepare(bce_hip_ctx *c) {
  uint32_t off = 0;
  for (int p = 0; p < 8; ++p) {
    plane_cfg_init(c->cfg[p], c->config[p]);
    c->stat_off[p] = off;
    off += (c->cfg[p].stat_bytes + 15u) & ~15u;
  }
  BCE_TRY(ensure(c, c->stat, off ? off : 16));
  BCE_TRY(ensure(c, c->dcfg, sizeof(PlaneCfg) * 8));
  BCE_HIP_TRY(c, hipMemsetAsync(c->stat.p, 0, off ? off : 16, c->stream));
  BCE_HIP_TRY(c, hipMemcpyAsync(c->dcfg.p, c->cfg, sizeof(PlaneCfg) * 8, hipMemcpyHostToDevice, c->stream));
  BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
  return BCE_HIP_OK;
}

int k4_flush_async(bce_hip_ctx *c, uint64_t nsym64, FlushSlot &slot) {
  if (nsym64 == 0) return BCE_HIP_OK;
  if (nsym64 >= (1ull << 31)) return BCE_HIP_E_OVERFLOW;
  const uint32_t nsym = (uint32_t)nsym64;
  const size_t b4 = (size_t)nsym * 4;
  BCE_TRY(ensure(c, c->skey[1], b4));
  for (int i = 0; i < 2; ++i) BCE_TRY(ensure(c, c->sval[i], b4));
  BCE_TRY(ensure(c, c->sout, (size_t)nsym * 8 + 16));      // + slack: the copy-out moves 16-byte units
  if (slot.pin_th.joinable()) slot.pin_th.join();                    // pinned ahead (k4_prepin): adopt it
  if (slot.pin_p) {
    if (slot.pin_cap > slot.cap) {
      slot_free_host(slot, &c->reg_unmaps);
      slot.h_out = slot.pin_p; slot.cap = slot.pin_cap; slot.registered = true;
      c->reg_maps++;
      c->pin_s += slot.pin_s; c->pin_bytes += slot.pin_cap * 8 + 16; c->pin_calls++;
    } else {
      (void)hipHostUnregister(slot.pin_p); huge_unmap(slot.pin_p, slot_map_bytes(slot.pin_cap));
    }
    slot.pin_p = nullptr; slot.pin_cap = 0;
  }
  if (slot.cap < nsym) {
    slot_free_host(slot, &c->reg_unmaps);
    size_t cap = (size_t)c->sym_cap > nsym ? (size_t)c->sym_cap : nsym;
    // A slot that has to grow in the middle of a compression (the symbol buffer grew: a high-entropy input of GBs has rounds of
    // 10^9 records, 8-17 GB of staging per slot): from the registered-mapping size on, a mapping of its own, touched by four
    // threads and registered where it is (common.h) -- hipHostMalloc takes ~0.15 s per GB, this 0.015.
    const size_t bytes = slot_map_bytes(cap);
    void *q = bytes >= reg_min_bytes((size_t)8 << 20) ? huge_map(bytes) : nullptr;
    if (q) {
      const double t0 = now_s();
      constexpr unsigned nt = 4;
      auto touch = [q, bytes](unsigned t) {
        const size_t lo = bytes / nt * t, hi = t + 1 == nt ? bytes : bytes / nt * (t + 1);
        for (size_t o = lo & ~(size_t)4095; o < hi; o += 4096) static_cast<volatile uint8_t *>(q)[o] = 0;
      };
      std::vector<std::thread> th;
      try { for (unsigned t = 1; t < nt; ++t) th.emplace_back(touch, t); } catch (...) {}
      const unsigned started = (unsigned)th.size() + 1;
      touch(0);
      for (unsigned t = started; t < nt; ++t) touch(t);
      for (auto &x : th) x.join();
      reg_map_settle(q, bytes);
      if (hipHostRegister(q, bytes, hipHostRegisterDefault) == hipSuccess) {
        slot.h_out = static_cast<uint64_t *>(q);
        slot.registered = true;
        c->reg_maps++;
        c->pin_s += now_s() - t0; c->pin_bytes += bytes; c->pin_calls++;
      } else {
        (void)hipGetLastError();
        huge_unmap(q, bytes);
        q = nullptr;
      }
    }
    if (!q) BCE_TRY(pin_alloc(c, (void **)&slot.h_out, cap * 8 + 16));
    slot.cap = cap;
  }
  // the stream the flush runs on: its own (beside the next K3 rounds) or the main one
  const bool own = c->overlap && c->k4_stream && !c->scan_mode;
  hipStream_t ks = own ? c->k4_stream : c->stream;
  if (own) {
    BCE_HIP_TRY(c, hipEventRecord(c->ev_k3_batch, c->stream));          // the symbols are complete ...
    BCE_HIP_TRY(c, hipStreamWaitEvent(ks, c->ev_k3_batch, 0));
  }
  if (!slot.ev_start) BCE_HIP_TRY(c, hipEventCreate(&slot.ev_start));
  if (!slot.ev_kend) BCE_HIP_TRY(c, hipEventCreate(&slot.ev_kend));
  // blocking sync: the coder thread that waits for the copy sleeps instead of spinning next to the busy coders
  if (!slot.ev_copy) BCE_HIP_TRY(c, hipEventCreateWithFlags(&slot.ev_copy, hipEventBlockingSync));
  if (c->copy_busy) BCE_HIP_TRY(c, hipStreamWaitEvent(ks, c->copy_busy, 0));   // `sout` is still being copied out
  BCE_HIP_TRY(c, hipEventRecord(slot.ev_start, ks));
  uint32_t *key[2] = {c->skey[0].as<uint32_t>(), c->skey[1].as<uint32_t>()};
  uint32_t *val[2] = {c->sval[0].as<uint32_t>(), c->sval[1].as<uint32_t>()};
  uint64_t gb = ((uint64_t)nsym + K4_T - 1) / K4_T;
  const uint32_t grid = (uint32_t)(gb < 8192 ? gb : 8192);
  hipLaunchKernelGGL(k4_iota_kernel, dim3(grid), dim3(K4_T), 0, ks, nsym, val[0]);
  int res = 0;
  BCE_TRY(radix_sort_pairs_on(c, ks, own ? c->rs_hist_k4 : c->rs_hist, key, val, nsym, kSymRunShift, kSymRunBits, &res, 10));
  // per-window work arrays, carved from one buffer: histT | stateW | winfo | queue | haltW | qcount
  const size_t nwin = ((size_t)nsym + 63) / 64;
  auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
  const size_t o_hist = 0, o_state = o_hist + nwin * 32, o_info = o_state + nwin * 32,
               o_queue = up16(o_info + nwin * 4), o_halt = o_queue + (nwin / 4 + 2) * 8, o_bits = up16(o_halt + nwin),
               o_qc = up16(o_bits + nwin * 8);
  BCE_TRY(ensure(c, c->k4w, o_qc + 16));
  uint8_t *wbuf = c->k4w.as<uint8_t>();
  K4Args a;
  a.keys = key[res]; a.vals = val[res];
  a.stat = c->stat.as<uint8_t>();
  a.cfg = c->dcfg.as<PlaneCfg>();
  a.out = c->sout.as<uint64_t>();
  a.esc = c->sesc.as<uint32_t>();
  a.histT = wbuf + o_hist; a.stateW = wbuf + o_state;
  a.winfo = reinterpret_cast<uint32_t *>(wbuf + o_info);
  a.queue = reinterpret_cast<uint2 *>(wbuf + o_queue);
  a.haltW = wbuf + o_halt;
  a.bitsW = reinterpret_cast<unsigned long long *>(wbuf + o_bits);
  a.qcount = reinterpret_cast<uint32_t *>(wbuf + o_qc);
  for (int p = 0; p < 8; ++p) a.stat_off[p] = c->stat_off[p];
  a.nsym = nsym;
  BCE_HIP_TRY(c, hipMemsetAsync(a.qcount, 0, 16, ks));
  uint64_t wb = ((uint64_t)nwin + (K4_T / 64) - 1) / (K4_T / 64);
  const uint32_t sgrid = (uint32_t)(wb < 16384 ? (wb ? wb : 1) : 16384);
  hipLaunchKernelGGL(k4_window_kernel, dim3(sgrid), dim3(K4_T), 0, ks, a);
  if (getenv("BCE_HIP_K4_DEBUG")) {
    // the long runs of this flush: how many, how long, which alphabet; and the time of k4_long_kernel alone
    BCE_HIP_TRY(c, hipStreamSynchronize(ks));
    uint32_t nq = 0;
    BCE_HIP_TRY(c, hipMemcpy(&nq, a.qcount, 4, hipMemcpyDeviceToHost));
    std::vector<uint2> q(nq);
    std::vector<uint32_t> hk(nsym);
    if (nq) BCE_HIP_TRY(c, hipMemcpy(q.data(), a.queue, (size_t)nq * 8, hipMemcpyDeviceToHost));
    BCE_HIP_TRY(c, hipMemcpy(hk.data(), a.keys, (size_t)nsym * 4, hipMemcpyDeviceToHost));
    std::vector<std::pair<uint32_t, uint32_t>> runs;       // (length, k)
    uint64_t in_long = 0;
    for (const uint2 &e : q) {
      uint32_t len = 0;
      while (e.x + len < nsym && (hk[e.x + len] >> kSymRunShift) == (e.y >> kSymRunShift)) ++len;
      runs.push_back({len, key_k(e.y)});
      in_long += len;
    }
    std::sort(runs.rbegin(), runs.rend());
    const double t0 = now_s();
    hipLaunchKernelGGL(k4_long_kernel, dim3(1024), dim3(K4_T), 0, ks, a);
    BCE_HIP_TRY(c, hipStreamSynchronize(ks));
    fprintf(stderr, "k4 flush: %u records, %u long runs holding %llu records, k4_long %.3f ms; longest:", nsym, nq, (unsigned long long)in_long, (now_s() - t0) * 1e3);
    for (size_t i = 0; i < runs.size() && i < 6; ++i) fprintf(stderr, " %u (k=%u)", runs[i].first, runs[i].second);
    fprintf(stderr, "\n");
  } else
  hipLaunchKernelGGL(k4_long_kernel, dim3(1024), dim3(K4_T), 0, ks, a);
  hipLaunchKernelGGL(k4_emit_kernel, dim3(sgrid), dim3(K4_T), 0, ks, a);
  BCE_HIP_TRY(c, hipGetLastError());
  // the copy runs on its own stream so that the next rounds (K3) overlap it
  BCE_HIP_TRY(c, hipEventRecord(c->ev_k4, ks));
  BCE_HIP_TRY(c, hipEventRecord(slot.ev_kend, ks));
  if (own) { BCE_HIP_TRY(c, hipEventRecord(c->ev_k4_done[c->flush_seq & 1u], ks)); ++c->flush_seq; }   // ... and have been read
  BCE_HIP_TRY(c, hipStreamWaitEvent(c->copy_stream, c->ev_k4, 0));
  BCE_HIP_TRY(c, hipMemcpyAsync(slot.h_out, c->sout.p, (size_t)nsym * 8, hipMemcpyDeviceToHost, c->copy_stream));
  BCE_HIP_TRY(c, hipEventRecord(slot.ev_copy, c->copy_stream));
  c->copy_busy = slot.ev_copy;
  slot.timed = true;
  return BCE_HIP_OK;
}

// A cold context's largest fixed cost is the pinned staging of its three flush slots (sym_cap records x 8 B each: 3 x 128 MB
// at the default flush size).  hipHostMalloc takes ~25 ms per 128 MB, nearly all of it the kernel handing out and zeroing
// the pages; registering pages that are already there takes 3-6 ms (tools/hip_floor.hip).  So each slot gets a thread that
// allocates and touches plain memory -- which needs no runtime, and bce_hip_create_sized starts it before the runtime's own
// 0.1 s of initialisation --, waits for the runtime, registers the memory (hipHostRegister) and leaves it for the slot's
// first flush to adopt.  (The first version called hipHostMalloc on the threads: three of them beside the input's
// host-to-device copy made that copy, from pageable memory, take 0.1 s instead of 0.011.)  Staging below 8 MB is pinned on
// the spot as before.
void k4_prepin(bce_hip_ctx *c, uint32_t n) {
  const size_t cap = (size_t)k3_symbol_capacity(c, n);
  if (cap * 8 < reg_min_bytes((size_t)8 << 20) || c->scan_mode) return;
  for (FlushSlot &slot : c->slot) {
    if (slot.cap >= cap || slot.pin_th.joinable() || slot.pin_p) continue;
    slot.pin_cap = cap;
    FlushSlot *sp = &slot;
    const int dev = c->device;
    try {
      slot.pin_th = std::thread([sp, dev, c] {
        const double t0 = now_s();
        // (a mapping of its own, never the C library's heap: common.h, "host memory registered with the runtime")
        const size_t bytes = slot_map_bytes(sp->pin_cap);
        void *q = huge_map(bytes);
        if (q) {
          for (size_t o = 0; o < bytes; o += 4096) static_cast<volatile uint8_t *>(q)[o] = 0;
          reg_map_settle(q, bytes);                                     // (its pages are there: they stay what they are)
        }
        int go;
        { std::unique_lock<std::mutex> lk(c->stage_mu); c->stage_cv.wait(lk, [c] { return c->stage_state != 0; }); go = c->stage_state; }
        if (q && (go != 1 || hipSetDevice(dev) != hipSuccess || hipHostRegister(q, bytes, hipHostRegisterDefault) != hipSuccess)) {
          (void)hipGetLastError();
          huge_unmap(q, bytes); q = nullptr;
        }
        sp->pin_p = static_cast<uint64_t *>(q);
        sp->pin_s = now_s() - t0;
      });
    } catch (...) { slot.pin_cap = 0; }                                // no thread: the flush pins on the spot
  }
}

void k4_prepin_join(bce_hip_ctx *c, bool drop) {
  for (FlushSlot &slot : c->slot) {
    if (slot.pin_th.joinable()) slot.pin_th.join();
    if (drop && slot.pin_p) { (void)hipHostUnregister(slot.pin_p); huge_unmap(slot.pin_p, slot_map_bytes(slot.pin_cap)); slot.pin_p = nullptr; slot.pin_cap = 0; }
  }
}

bool k4_in_flight(bce_hip_ctx *c) {
  if (!c->overlap || !c->k4_stream || c->flush_seq == 0) return false;
  return hipEventQuery(c->ev_k4_done[(c->flush_seq - 1u) & 1u]) != hipSuccess;
}

int k4_flush(bce_hip_ctx *c, uint64_t nsym, FlushSlot &slot) {
  if (nsym == 0) return BCE_HIP_OK;
  BCE_TRY(k4_flush_async(c, nsym, slot));
  BCE_HIP_TRY(c, hipEventSynchronize(slot.ev_copy));
  slot.timed = false;
  return BCE_HIP_OK;
}

}  // namespace bce
