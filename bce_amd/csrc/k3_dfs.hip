// k3_dfs.hip -- K3, depth-first tail.  Finishes BCE::code (bce.cpp:1236-1374) once few nodes are alive.
//
// Why: the number of rounds is the length in bits of the longest repeated string; a 20 KB repeat keeps one
// two-row interval alive for 160 000 rounds, each of which is a pass-through that emits nothing.  The round
// structure only matters for the ORDER of the symbols inside each coder's stream (round, then s).  So when the
// live set is small and almost all of the 8n-8 nodes are done, every live node is handed to one thread that walks
// its whole subtree depth-first with a private stack -- no barriers, no launches -- and tags each symbol with
// (plane, round, s); the tagged symbols are sorted into stream order afterwards and go through K4 as usual.
//
// A walker at plane 0 (= BWT row order) whose x rows all have the same preceding bytes can SKIP the chain: the
// next k bytes are 8k pass-through rounds, the rows stay adjacent (LF-mapping keeps the order of rows with equal
// bytes), and they end at row ISA[SA[s] - k], 8k rounds later, with (x0, x1) unchanged.  k is found by comparing
// the text backwards from the x suffix-array positions (exact, no hashing).  SA / ISA are K1's arrays; the skip
// is disabled when K1 did not end with all rotations distinct (periodic inputs) or the BWT was injected.
//
// Chains that cannot be skipped but are regular -- the rows are equally spaced text positions inside a run of one
// byte or a periodic table, and one row leaves every period -- are expanded in closed form: stair_run (one region),
// stairs_run (the same pattern in up to 8 places).  Their events are independent of each other; long stretches go on
// a job list that a grid-wide kernel works off after the pass (kd_jobs_kernel).
//
// Passes: lane = walker while many nodes are queued; wave = walker (scalar registers, scalar loads, a tight loop
// over pass-through nodes) once <= KD_UNI_MAX are left.  A walker hands its work on after a.budget nodes.
//
// Chains of MANY rows that lose a few rows per byte (the all-zero context of a binary: neither a skip nor a staircase)
// are taken 64 byte levels at a time by the wave-walkers: spine_burst below.
//
// Before the walkers, while more than KD_LOCAL_FROM nodes are alive: workgroup-local rounds (k3_local_kernel).
#include <stdio.h>
#include <stdlib.h>

#include "common.h"
#include "scan_util.h"
#include "k3_args.h"

namespace bce {

constexpr int KD_T = 64;                       // one wave per block: walkers spread over the CUs
constexpr uint32_t KD_STACK = 32;              // pending siblings per walker (the older half is handed on when full)
constexpr uint32_t KD_MAXX = 2048;             // rows compared for a chain skip
constexpr uint32_t KD_NEAR = 256;              // bytes compared one lane per row before the whole wave takes one pair
constexpr uint32_t KD_WALKERS = 1u << 17;      // walkers per pass (each takes several queued nodes when more are queued)
constexpr uint32_t KD_QUEUE = 4u << 20;        // queued nodes between passes
constexpr uint32_t KD_UNI_MAX = 4096;           // up to this many queued walkers a pass gives every walker a wave of its own
constexpr uint32_t KD_SBLK = 8;                // tagged-symbol slots a walker reserves at a time (one atomic with return per 8 symbols)
constexpr uint32_t KD_HOLE = 0x7FFu;           // (plane, round high) of an unused slot: sorts behind every real symbol
constexpr uint32_t KD_STAIR_MIN = 32;            // rows from which a walker at plane 0 looks for a staircase (see stair_run)
constexpr uint32_t KD_STAIR_RETRY = 256;         // nodes a walker waits after a failed look
constexpr uint32_t K3_DFS_PASS = 4096;         // nodes one WAVE-walker classifies per pass before it hands its work on
constexpr uint32_t K3_DFS_LANE_PASS = 256;     // ... and one lane-walker (64 of them advance in lockstep: a pass lasts as long as its longest walk)

struct DfsCtl {
  uint32_t nsym;         // tagged symbols emitted
  uint32_t err;          // 2 symbol capacity, 4 queue capacity, 8 a spine burst did not arrive where it had to (cannot happen)
  uint64_t nodes;        // nodes visited (skipped pass-through nodes included)
  uint64_t maxround;
  uint32_t cntp[8];      // symbols per plane
  uint32_t queued;       // nodes handed to the next pass
  uint32_t spilled;      // k3_local_kernel: children that did not fit a workgroup's LDS list (worked off by another local pass)
  uint32_t njobs;        // staircase jobs queued so far
  uint32_t spine_levels; // byte levels done by spine bursts
  uint32_t dbg_hist[32]; uint32_t dbg_maxvis; uint32_t dbg_skips; uint64_t dbg_skipbytes;
  uint32_t dbg_stairs, dbg_stairsyms;
  unsigned long long dbg_slow;   // slowest chain-skip comparison: cycles >> 10 in the high half, x << 20 | min(kk, 2^20 - 1) below
  unsigned long long dbg_maxwave;   // slowest wave: cycles >> 12 of (walk, chain-skip comparisons, staircases), 20 bits each
  uint64_t dbg_cyc[4];   // wave cycles: whole walk, chain-skip comparisons, staircases; [3] = staircase looks
  uint64_t dbg_gen[4];   // wave-walkers: general-path nodes, their cycles, pass-through-loop nodes, their cycles
  uint32_t dbg_spine[4]; // spine bursts: tried, done, byte levels, nodes queued
  uint32_t dbg_cont[4];  // which boundary's run ended a scan's stretch: B only, E only, both, all lanes fine
  uint32_t dbg_why[8];   // why a burst's chain ended: small node, no c-row in A / B / E, full, run ended at once; [6] scans, [7] levels with x1 < 4
};

struct DNode { uint32_t s, x0, x1, plane; uint64_t round; };

// A staircase stretch (see stair_run / stairs_run) whose events are independent of each other.  Short ones are written
// by the walker's own wave; long ones (10^4..10^6 events) are put on a job list and written by a grid-wide kernel
// after the pass (kd_jobs_kernel), so that no single wave sits on them.
constexpr uint32_t KD_REGIONS = 8;
constexpr uint32_t KD_JOBS = 4096;               // jobs per depth-first tail (when the list is full the walker writes the events itself)
constexpr uint32_t KD_JOB_MIN = 2048;            // events from which a stretch becomes a job
struct StairJob {
  uint32_t kind;                                  // 1 one region, 2 several
  uint32_t p, x0, x1, base, events;
  uint64_t round;
  uint32_t as, asc, B, c, js;                     // one region
  uint32_t nr, beta, nbar, hi0;                   // several regions
  uint32_t S[KD_REGIONS], E[KD_REGIONS], Bv[KD_REGIONS], X[KD_REGIONS], D[KD_REGIONS], PE[KD_REGIONS + 1];
  uint32_t tS[KD_REGIONS], tE[KD_REGIONS];        // scratch of the decomposition
};
typedef __attribute__((address_space(4))) Granule ConstGranule;

// A wave-walker's block has KD_HELP waves when few walkers are left (round 3): wave 0 walks, the others sleep on an LDS word
// and wake up for the one thing a single wave is hopeless at -- the scan of ALL rows of a large node for the starts and
// ends of its staircases (three dependent loads per 256 rows: 8 M cycles for a 10^6-row node, a whole pass waiting).
constexpr int KD_HELP = 8;
constexpr uint32_t KD_HELP_MINX = 4096;          // rows from which the helpers are called
constexpr uint32_t KD_HELP_MAXW = 1024;          // walkers of a pass up to which every walker gets helpers
struct ScanJob { uint32_t seq, s, x, p, nS, nE, fail, done; };
// LDS written by this wave is visible to this wave (one wave = in-order LDS): fence the compiler, nothing to wait for
__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

struct DfsArgs {
  K3Args k;
  uint32_t *traw;        // scan mode: scan_pack's word per tagged symbol (raw_symbol), else null
  const uint8_t *text;
  const uint8_t *bwt;    // K1's output: bwt[r] = the byte before row r (spine bursts)
  const uint32_t *sa, *isa;
  uint32_t skip_ok;
  DfsCtl *dctl;
  uint32_t *tkey, *tesc, *ts, *trlo, *trhi;
  DNode *stacks;
  StairJob *jobs;        // [KD_JOBS]
  const DNode *in;       // queued nodes of this pass; nullptr = the planes' node lists (first pass)
  DNode *out;            // nodes handed to the next pass
  uint32_t in_count, out_cap;
  uint64_t round0;
  uint32_t symcap;
  uint32_t budget;
  uint32_t dbg;
  DNode *spill;          // k3_local_kernel: overflow of a workgroup's LDS list
  uint32_t spill_cap;
  uint32_t lb_budget;    // rounds a workgroup runs in one pass
  uint32_t lane_budget;  // nodes a LANE-walker classifies per pass (the wave runs as long as its longest walker: keep it short)
  uint32_t lpw;          // lane-walkers per wave (the other lanes only help with the cooperative comparisons)
  uint32_t lb_quiet;     // pass-through levels after which a child arriving at plane 0 is a chain for the walkers
  uint32_t no_spine;     // debug: no spine bursts
};

// granule_rank1 with 64-bit masks (fewer scalar instructions)
__device__ __forceinline__ uint32_t rank1_64(const Granule &g, uint32_t o) {
  const uint64_t lo = (uint64_t)g.w0 | ((uint64_t)g.w1 << 32);
  const uint64_t m = o >= 64u ? ~0ull : ((1ull << o) - 1ull);
  const uint32_t m2 = o > 64u ? ((1u << (o - 64u)) - 1u) : 0u;
  return g.cum + (uint32_t)__popcll(lo & m) + (uint32_t)__popc(g.w2 & m2);
}
__device__ __forceinline__ uint32_t ld32u(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o; o >>= 1) { const uint32_t w = (uint32_t)__shfl_xor((int)v, o); v = w < v ? w : v; }
  return v;
}
__device__ __forceinline__ uint32_t cyc_back(uint32_t p, uint32_t d, uint32_t n) { return p >= d ? p - d : p + n - d; }   // d < n

// Number of bytes on which the rotations starting at p and q agree going BACKWARDS (cyclic), capped at lim.
// Executed by the WHOLE wave with uniform arguments: while neither side wraps, each of the 64 lanes compares 16
// bytes, i.e. 1 KB per step (a multi-megabyte repeat is a few thousand steps); the rest goes byte by byte.
__device__ __forceinline__ uint32_t lce_back_wave(const uint8_t *__restrict__ T, uint32_t n, uint32_t p, uint32_t q,
                                                  uint32_t lim, uint32_t lane) {
  uint32_t t = 0;
  uint32_t cp = p, cq = q;                                   // the next bytes to compare are T[cp-1], T[cq-1] (cyclic)
  while (t < lim) {
    if (cp == 0) cp = n;
    if (cq == 0) cq = n;
    uint32_t run = cp < cq ? cp : cq;                        // bytes before either side wraps
    if (run > lim - t) run = lim - t;
    if (cp >= 1024u && cq >= 1024u) {
      // lane L owns the bytes at distances [16 L, 16 L + 16) from the current position (a whole KB is read even when
      // fewer bytes are wanted: a difference beyond the limit does not count)
      const uint8_t *a = T + (cp - 16u * lane - 16u), *b = T + (cq - 16u * lane - 16u);
      uint32_t match = 16;                                   // bytes matching from the NEAREST (highest address) end
#pragma unroll
      for (int w = 3; w >= 0; --w) {
        const uint32_t d = ld32u(a + 4 * w) ^ ld32u(b + 4 * w);
        if (d && match == 16) match = (uint32_t)(3 - w) * 4u + ((uint32_t)__clz((int)d) >> 3);
      }
      const uint64_t mm = __ballot(match < 16);
      if (mm) {
        const uint32_t L = (uint32_t)__ffsll((long long)mm) - 1u;     // nearest lane with a mismatch
        const uint32_t at = 16u * L + (uint32_t)__builtin_amdgcn_readlane((int)match, (int)L);
        return at < lim - t ? t + at : lim;
      }
      if (lim - t <= 1024u) return lim;
      t += 1024; cp -= 1024; cq -= 1024;
    } else {                                                 // within 1 KB of the start of the text: byte by byte up to the wrap
      uint32_t i = 0;
      while (i < run && T[cp - 1 - i] == T[cq - 1 - i]) ++i;
      t += i; cp -= i; cq -= i;
      if (i < run) return t;
    }
  }
  return t;
}

// One lane: how many of the 16 bytes at backward distances [t, t + 16) from p and q agree (counted from the nearest).
__device__ __forceinline__ uint32_t match16_back(const uint8_t *__restrict__ T, uint32_t n, uint32_t p, uint32_t q,
                                                 uint32_t t) {
  const uint32_t cp = cyc_back(p, t, n), cq = cyc_back(q, t, n);      // next bytes: T[cp-1], T[cq-1] (cyclic)
  if (cp >= 16u && cq >= 16u) {
    const uint8_t *a = T + (cp - 16u), *b = T + (cq - 16u);
    uint32_t match = 16;
#pragma unroll
    for (int w = 3; w >= 0; --w) {
      const uint32_t d = ld32u(a + 4 * w) ^ ld32u(b + 4 * w);
      if (d && match == 16) match = (uint32_t)(3 - w) * 4u + ((uint32_t)__clz((int)d) >> 3);
    }
    return match;
  }
  uint32_t i = 0;
  for (; i < 16u; ++i) {
    const uint32_t ia = cp > i ? cp - 1u - i : cp + n - 1u - i, ib = cq > i ? cq - 1u - i : cq + n - 1u - i;
    if (T[ia] != T[ib]) break;
  }
  return i;
}

// Length in bytes of the pass-through chain below the x rows [s, s + x) of plane 0: the number of preceding bytes
// on which all x rotations agree (0 = none).  Whole wave, uniform arguments.  First one lane per row for the
// nearest KD_NEAR bytes (most chains of many rows end there); if every row survives that, the whole wave takes
// the rows one after the other (long repeats: 1 KB per step).
__device__ __forceinline__ uint32_t chain_bytes(const DfsArgs &a, uint32_t s, uint32_t x, uint32_t lane) {
  const uint32_t n = a.k.n;
  const uint32_t pa = a.sa[s];
  const uint32_t ra = a.isa[pa];
  uint32_t kk = n - 1;
  for (uint32_t base = 1; base < x && kk; base += 64) {
    const uint32_t i = base + lane;
    bool have = i < x;
    uint32_t pb = pa;
    if (have) {
      pb = a.sa[s + i];
      if (a.isa[pb] == ra) have = false;                     // identical rotations (periodic input): agree for ever
    }
    const uint32_t lim = kk < KD_NEAR ? kk : KD_NEAR;
    for (uint32_t t = 0; t < lim; t += 16) {
      const uint32_t m = have ? match16_back(a.text, n, pa, pb, t) : 16u;
      if (__ballot(m < 16u)) {
        const uint32_t e = t + wave_min_u32(m);
        kk = e < kk ? e : kk;
        break;
      }
    }
  }
  // Every row agrees with the first on the nearest KD_NEAR bytes.  Deepen in stages (4 KB, 64 KB, 1 MB, ...): the
  // answer is the MINIMUM over the rows, so no pair is compared further than the stage -- two rows that happen
  // to share megabytes (copies of a file) cost nothing when a third row differs after a kilobyte.
  uint32_t done = KD_NEAR;                                    // all rows agree with the first on [0, done)
  for (uint32_t lim = 4096; kk > done; lim = lim < (1u << 28) ? lim * 16u : n) {      // n - 1 > KD_NEAR here, so offsets stay < n
    uint32_t upto = kk < lim ? kk : lim;
    const uint32_t pa2 = cyc_back(pa, done, n);
    for (uint32_t i = 1; i < x && upto > done; ++i) {
      const uint32_t pb = a.sa[s + i];
      if (a.isa[pb] == ra) continue;
      const uint32_t l = done + lce_back_wave(a.text, n, pa2, cyc_back(pb, done, n), upto - done, lane);
      if (l < upto) { upto = l; kk = l; }
    }
    done = upto;
  }
  return kk == n - 1 ? 0u : kk;                               // every row identical: cannot happen for a live node
}

// Staircase: the chain below a node whose x rows are the rotations at text positions a, a + p, ..., a + (x-1) p
// inside a region of period p (a long run of one byte is p = 1; tables of equal records are p = record size).
// Such a chain cannot be skipped -- every p bytes the row that reaches the start of the region meets a different
// preceding byte and leaves, which codes one symbol -- and walking it is 8 nodes per byte of the region, one after
// the other.  But it is entirely regular.  With a' = the start of the region (T[a'-1] = d differs from
// c = T[a'-1+p], T[j] = T[j+p] from a' on), B = a - a', j* = the lowest bit in which c and d differ:
//   * the first B bytes are pass-through for all rows;
//   * event e = 0, 1, ... happens 8 (B + e p) rounds below the node, at plane j*: the rows are the positions
//     a' + m p, m < x - e; all of them are preceded by c except a' (the first row if rows ascend with positions,
//     else the last), so the node there has one row with the other bit: k = 2, the symbol says on which side of
//     the x0 | x1 boundary that row is (always the same side), contexts (_0x, x1, x) = (x-e-1 or 1, x1, x-e);
//     the row leaves without a child of its own, the other x-e-1 rows go on;
//   * the side the leaving row is on shrinks by one per event; when it is empty the chain ends: xs = x0 or x1 events.
// The plane-0 position of the node at event e is ISA[a'] or ISA[a' + (x-e-1) p]; its position in plane j* (the
// symbol's sort key) follows through j* pass-through levels.  So the events are independent of each other: one
// lane each.  Everything is checked exactly (suffix-array stride, periodicity by comparing the text), nothing is
// assumed about why the rows are there.  Whole wave, uniform arguments; returns false if the node is no staircase.
// `bce -s` (scan mode): the ScanCoders want scan_pack's word of every symbol (bce_core.h), not the model record
__device__ __forceinline__ void raw_symbol(const DfsArgs &a, uint32_t i, uint32_t sym, uint32_t kk, uint32_t c1, uint32_t c2, uint32_t cs) {
  if (a.traw) a.traw[i] = scan_pack(sym, kk, c1, c2, cs);
}
__device__ __forceinline__ uint32_t rank1_plane(const K3Args &k, uint32_t p, uint32_t pos) {
  const uint32_t g = div96(pos);
  return granule_rank1((k.gran + (size_t)p * k.ngran)[g], pos - g * 96u);
}
// Event e of a one-region staircase (see stair_run).
__device__ __forceinline__ void stair_event(const DfsArgs &a, const StairJob &J, uint32_t e) {
  const K3Args &k = a.k;
  const uint32_t L0 = J.asc, p = J.p, c = J.c, js = J.js, cb = (c >> js) & 1u;
  const uint32_t xe0 = J.x0 - (L0 ? e : 0u), xe1 = J.x1 - (L0 ? 0u : e), xe = xe0 + xe1;
  uint32_t pos = a.isa[L0 ? J.as : J.as + (xe - 1u) * p];
  for (uint32_t q = 0; q < js; ++q) {
    const uint32_t r = rank1_plane(k, q, pos);
    pos = ((c >> q) & 1u) ? k.zeros[q] + r : pos - r;
  }
  uint32_t kw, ew;
  pack_symbol(k.cfg[js], js, cb ? L0 : 1u - L0, 2u, cb ? 1u : xe - 1u, xe1, xe, kw, ew);
  const uint64_t r = J.round + 8ull * ((uint64_t)J.B + (uint64_t)e * p) + js;
  const uint32_t i = J.base + e;
  a.tkey[i] = kw; a.tesc[i] = ew; a.ts[i] = pos;
  raw_symbol(a, i, cb ? L0 : 1u - L0, 2u, cb ? 1u : xe - 1u, xe1, xe);
  a.trlo[i] = (uint32_t)r;
  a.trhi[i] = (uint32_t)(r >> 32) | (js << 8);
}

// Put a long stretch on the job list (whole wave; `J` uniform, or in LDS when `lds`).  False: write the events yourself.
__device__ __forceinline__ bool stair_defer(const DfsArgs &a, const StairJob &J, uint32_t lane, bool lds) {
  if (J.events < KD_JOB_MIN) return false;
  uint32_t idx = 0;
  if (lane == 0) idx = atomicAdd(&a.dctl->njobs, 1u);
  idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
  if (idx >= KD_JOBS) return false;                           // (the counter stays above KD_JOBS: the host clamps it)
  const uint32_t *src = reinterpret_cast<const uint32_t *>(&J);
  uint32_t *dst = reinterpret_cast<uint32_t *>(a.jobs + idx);
  if (lds) { for (uint32_t w = lane; w < sizeof(StairJob) / 4; w += 64) dst[w] = src[w]; }
  else if (lane == 0) a.jobs[idx] = J;
  return true;
}

__device__ __forceinline__ bool stair_run(const DfsArgs &a, uint32_t s, uint32_t x0, uint32_t x1, uint64_t round, uint32_t lane,
                                       uint64_t &nodes_out, uint64_t &maxround_out) {
  const K3Args &k = a.k;
  const uint32_t n = k.n, x = x0 + x1;
  const uint32_t pa = a.sa[s], pb = a.sa[s + 1], pz = a.sa[s + x - 1];
  const bool asc = pb > pa;
  const uint32_t p = asc ? pb - pa : pa - pb;
  const uint64_t span = (uint64_t)(x - 1) * p;
  if (p == 0 || (asc ? (uint64_t)pa + span != pz : (uint64_t)pz + span != pa)) return false;
  const uint32_t lo = asc ? pa : pz;                          // a
  if ((uint64_t)lo + span + p > n) return false;              // the comparison below would wrap
  bool bad = false;
  for (uint32_t i0 = 0; i0 < x && !__any(bad); i0 += 256) {
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
      const uint32_t i = i0 + 64u * j + lane;
      if (i < x) bad |= a.sa[s + i] != (asc ? pa + i * p : pa - i * p);
    }
  }
  if (__any(bad)) return false;
  // T[j] == T[j + p] on [a', a + (x-1) p) and the first byte before that where it fails
  const uint32_t p1 = lo + (uint32_t)span;
  const uint32_t lce = lce_back_wave(a.text, n, p1, p1 + p, p1, lane);      // down to position 0, no wrap
  if (lce < span) return false;                               // not periodic
  const uint32_t B = lce - (uint32_t)span, as = lo - B;
  const uint32_t d = as ? a.text[as - 1u] : a.text[n - 1u], c = a.text[as + p - 1u];     // rotations: position 0 follows n - 1
  if (d == c) return false;                                   // (only when as == 0: the period goes on around the end)
  const uint32_t js = (uint32_t)__ffs((int)(c ^ d)) - 1u;
  const uint32_t L0 = asc ? 1u : 0u;                          // the leaving row is among the first x0 rows
  const uint32_t xs = L0 ? x0 : x1;
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(&a.dctl->nsym, xs);
  base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
  if ((uint64_t)base + xs > a.symcap) { a.dctl->err = 2; nodes_out = 0; maxround_out = 0; return true; }
  StairJob J;
  J.kind = 1; J.p = p; J.x0 = x0; J.x1 = x1; J.base = base; J.events = xs; J.round = round;
  J.as = as; J.asc = asc ? 1u : 0u; J.B = B; J.c = c; J.js = js;
  if (!stair_defer(a, J, lane, false))
    for (uint32_t e = lane; e < xs; e += 64) stair_event(a, J, e);
  if (lane == 0) {
    atomicAdd(&a.dctl->cntp[js], xs);
    if (a.dbg) { atomicAdd(&a.dctl->dbg_stairs, 1u); atomicAdd(&a.dctl->dbg_stairsyms, xs); }
  }
  const uint64_t bytes = (uint64_t)B + (uint64_t)(xs - 1u) * p;
  nodes_out = 8ull * bytes + js + 1u;
  maxround_out = round + 8ull * bytes + js;
  return true;
}

// Several staircases in one node: the same periodic pattern occurs in R places (2 <= R <= KD_REGIONS), so the rows are R
// progressions lo_r, lo_r + p, ..., hi_r.  Per row the depth (in bytes) at which it leaves is known in closed form --
// row m of region r meets the byte before its region after B_r + m p bytes, B_r = lo_r - (start of the region) --
// so the main chain at depth t is "the rows with leave depth >= t", in their old order, and every event (a depth at
// which rows leave) can be worked out on its own from R-term sums: the counts, the node's row (the minimum of the
// ISA of the regions' first and last surviving positions), and through the 8 planes the symbols of the node that
// the leaving rows split off (rows leaving with different bytes leave at different planes; at most one symbol per
// leaving row).  The stretch stops before any region is down to two rows, so that the leaving rows are never
// end rows: all of them are then on the same side of the x0 | x1 boundary (only a region's LAST row can be on
// the other side -- checked), rows that leave never form a node of their own, and the main chain cannot die
// inside the stretch.  The walker continues normally from the node at the stop depth; when a region has died it looks
// again.  Everything is verified exactly: decomposition (every row has its neighbour at -p / +p in the node or is a
// start / an end; starts and ends pair up as disjoint position intervals), periodicity of every region and equality
// of the pattern across regions by text comparison.  Whole wave, uniform arguments.  Returns 0 if not applicable.
constexpr uint32_t KD_STAIRS_MAXX = 1u << 24;
typedef StairJob StairRegs;                      // the walker's copy lives in LDS

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
  return v;
}

// State of a chain of several staircases at depth t (bytes below the node): rows left, the first row, and the rows
// that leave exactly at t (their bytes and their event numbers).
__device__ __forceinline__ void stairs_state(const DfsArgs &a, const StairJob &J, uint32_t t, uint32_t &xall, uint32_t &smin,
                                             uint32_t &nl, uint32_t *ld, uint32_t *le) {
  const uint32_t p = J.p;
  xall = 0; smin = 0xFFFFFFFFu; nl = 0;
  for (uint32_t r = 0; r < J.nr; ++r) {
    const uint32_t B = J.Bv[r];
    const uint32_t mm = t <= B ? 0u : (t - B + p - 1u) / p;            // first row of region r that is still there
    xall += J.X[r] - mm;
    const uint32_t r1 = a.isa[J.S[r] + mm * p - t], r2 = a.isa[J.E[r] - t];
    smin = r1 < smin ? r1 : smin; smin = r2 < smin ? r2 : smin;
    if (t >= B && (t - B) % p == 0) { ld[nl] = J.D[r]; le[nl] = J.PE[r] + mm; ++nl; }
  }
}

// Event e (numbered region by region) of a chain of several staircases (see stairs_run).
__device__ __forceinline__ void stairs_event(const DfsArgs &a, const StairJob &J, uint32_t e) {
  const K3Args &k = a.k;
  const uint32_t p = J.p, beta = J.beta, nbar = J.nbar;
  uint32_t r = 0;
  while (e >= J.PE[r + 1u]) ++r;
  const uint32_t t = J.Bv[r] + (e - J.PE[r]) * p;
  uint32_t xall, pos, nl, ld[KD_REGIONS], le[KD_REGIONS];
  stairs_state(a, J, t, xall, pos, nl, ld, le);
  if (le[0] != e) return;                                    // several regions may leave a row at this depth: the first one works
  const uint32_t c = a.text[J.hi0 - t - 1u];
  const uint32_t mainb = xall - nbar - nl;                   // rows that stay, on the common side
  uint32_t lm = (1u << nl) - 1u, nsy = 0;
  for (uint32_t q = 0; q < 8u && lm; ++q) {
    const uint32_t cq = (c >> q) & 1u;
    uint32_t mino = 0, ones = 0, lq = 0;
    for (uint32_t l = 0; l < nl; ++l) if ((lm >> l) & 1u) {
      const uint32_t bit = (ld[l] >> q) & 1u;
      ++lq; ones += bit;
      if (bit != cq) mino |= 1u << l;
    }
    if (mino) {
      const uint32_t xb = mainb + lq, x0n = beta ? nbar : xb, x1n = beta ? xb : nbar, xn = x0n + x1n;
      const uint32_t n1x = (cq ? mainb + nbar : 0u) + ones, n0x = xn - n1x;
      const int32_t u = (int32_t)(x0n - n1x), v = (int32_t)(n1x - x1n);
      const uint32_t mn = u < 0 ? 0u : (uint32_t)u, mx = x0n - (v < 0 ? 0u : (uint32_t)v);
      if (mx != mn) {
        const uint32_t n0x0 = beta ? (cq ? 0u : nbar) : ((cq ? 0u : mainb) + (lq - ones));   // zeros among the first x0 rows
        uint32_t kw, ew;
        pack_symbol(k.cfg[q], q, n0x0 - mn, mx - mn + 1u, n0x, x1n, xn, kw, ew);
        const uint64_t rr = J.round + 8ull * t + q;
        const uint32_t i = J.base + le[nsy];
        a.tkey[i] = kw; a.tesc[i] = ew; a.ts[i] = pos;
        raw_symbol(a, i, n0x0 - mn, mx - mn + 1u, n0x, x1n, xn);
        a.trlo[i] = (uint32_t)rr;
        a.trhi[i] = (uint32_t)(rr >> 32) | (q << 8);
        atomicAdd(&a.dctl->cntp[q], 1u);
        ++nsy;
      }
      lm &= ~mino;
    }
    const uint32_t r1 = rank1_plane(k, q, pos);
    pos = cq ? k.zeros[q] + r1 : pos - r1;
  }
  for (uint32_t j = nsy; j < nl; ++j) { const uint32_t i = J.base + le[j]; a.ts[i] = 0; a.trlo[i] = 0; a.trhi[i] = KD_HOLE; }
}

// The events of the stretches queued during a pass, spread over the whole grid.
__global__ __launch_bounds__(256) void kd_jobs_kernel(DfsArgs a, uint32_t first) {
  uint32_t nj = a.dctl->njobs;
  nj = nj < KD_JOBS ? nj : KD_JOBS;
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  for (uint32_t j = first; j < nj; ++j) {
    const StairJob &J = a.jobs[j];
    const uint32_t ne = J.events;
    if (J.kind == 1u) { for (uint32_t e = tid; e < ne; e += nt) stair_event(a, J, e); }
    else { for (uint32_t e = tid; e < ne; e += nt) stairs_event(a, J, e); }
  }
}

// The scan of a node's rows for the starts and the ends of its progressions (stride p): wave wv of nw takes the chunks
// wv, wv + nw, ... (256 rows each, 4 per lane so that the dependent loads of 4 rows overlap; from both ends inwards: the
// starts and ends of many regions gather there, and too many = give up early).  Results go to R->tS / R->tE through the
// counters of the job; more than KD_REGIONS of either sets job->fail.
__device__ __forceinline__ void stairs_scan_share(const DfsArgs &a, StairRegs *R, ScanJob *job, uint32_t wv, uint32_t nw, uint32_t lane) {
  const uint32_t n = a.k.n;
  const uint32_t s = job->s, x = job->x, p = job->p;
  const uint64_t lt = (1ull << lane) - 1ull;
  const uint32_t nchunk = (x + 255u) / 256u;
  for (uint32_t it = wv; it < nchunk; it += nw) {
    if (__hip_atomic_load(&job->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return;
    const uint32_t base = 256u * ((it & 1u) ? nchunk - 1u - (it >> 1) : (it >> 1));
    uint32_t q[4];
    bool have[4], st[4], en[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const uint32_t i = base + 64u * j + lane; have[j] = i < x; q[j] = have[j] ? a.sa[s + i] : 0u; }
    uint32_t rp[4], rn[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      rp[j] = (have[j] && q[j] >= p) ? a.isa[q[j] - p] : 0xFFFFFFFFu;
      rn[j] = (have[j] && (uint64_t)q[j] + p < n) ? a.isa[q[j] + p] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      st[j] = have[j] && !(rp[j] != 0xFFFFFFFFu && rp[j] - s < x);
      en[j] = have[j] && !(rn[j] != 0xFFFFFFFFu && rn[j] - s < x);
      const uint64_t bs = __ballot(st[j]), be = __ballot(en[j]);
      const uint32_t cs = (uint32_t)__popcll(bs), ce = (uint32_t)__popcll(be);
      if (cs | ce) {
        uint32_t oS = 0, oE = 0;
        if (lane == 0) { if (cs) oS = atomicAdd(&job->nS, cs); if (ce) oE = atomicAdd(&job->nE, ce); }
        oS = (uint32_t)__builtin_amdgcn_readfirstlane((int)oS);
        oE = (uint32_t)__builtin_amdgcn_readfirstlane((int)oE);
        if (oS + cs > KD_REGIONS || oE + ce > KD_REGIONS) {
          if (lane == 0) __hip_atomic_store(&job->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          return;
        }
        if (st[j]) R->tS[oS + (uint32_t)__popcll(bs & lt)] = q[j];
        if (en[j]) R->tE[oE + (uint32_t)__popcll(be & lt)] = q[j];
      }
    }
  }
}
// the waves 1 .. KD_HELP - 1 of a wave-walker's block: sleep until the walker posts a scan, take a share, report, sleep
__device__ __forceinline__ void stairs_helper(const DfsArgs &a, StairRegs *R, ScanJob *job, uint32_t wv, uint32_t lane) {
  uint32_t seen = 0;
  for (;;) {
    uint32_t sq;
    do {
      __builtin_amdgcn_s_sleep(16);
      sq = __hip_atomic_load(&job->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
    } while (sq == seen);
    if (sq == 0xFFFFFFFFu) return;
    seen = sq;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    stairs_scan_share(a, R, job, wv, (uint32_t)KD_HELP, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(&job->done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

__device__ __forceinline__ uint32_t stairs_run(const DfsArgs &a, StairRegs *R, ScanJob *job, bool job_helpers, uint32_t s, uint32_t x0, uint32_t x1, uint64_t round,
                                               uint32_t lane, DNode &next, uint64_t &nodes_out) {
  const K3Args &k = a.k;
  const uint32_t n = k.n, x = x0 + x1;
  if (x > KD_STAIRS_MAXX) return 0;
  const uint64_t tz0 = a.dbg ? clock64() : 0;
  // stride: the smallest distance between neighbouring rows among 64 samples (rows of one region are neighbours
  // almost everywhere; a wrong guess fails the decomposition)
  uint32_t p;
  {
    const uint32_t i = (uint32_t)(((uint64_t)(x - 2u) * lane) / 63u);
    const uint32_t qa = a.sa[s + i], qb = a.sa[s + i + 1u];
    p = wave_min_u32(qa > qb ? qa - qb : qb - qa);
  }
  if (p == 0 || p > 65536u) return 0;
  // starts and ends of the progressions
  uint32_t nS = 0, nE = 0;
  const uint64_t lt = (1ull << lane) - 1ull;
  // Fast path (round 3): 64 sample rows GALLOP along the stride -- q, q + p, q + 2p, q + 4p, ... while the position is a
  // row of the node, then bisect -- to the two ends of their progression: O(64 log x) dependent loads instead of the scan
  // of all x rows below (one wave, three dependent loads per 256 rows: 8 M cycles for the 10^6-row nodes of the binary
  // corpus, which made a whole pass wait for one wave).  A progression found this way is only a claim that every step
  // between its ends is a row; the checks below do not rely on it: the periodicity test proves it (lo and lo + p are
  // rows, so the period reaches the node's depth before lo, and every lo + jp <= hi has the context of lo), the
  // intervals must be disjoint, and the rows must add up to x -- x distinct rows of an x-row node are the node.  If
  // the samples miss a region the sum falls short and the scan runs as before.
  bool fast = false;
  if (x >= 1024u) {
    const uint32_t i = (uint32_t)(((uint64_t)(x - 1u) * lane) / 63u);
    const uint32_t q0 = a.sa[s + i];
    auto member = [&](uint64_t q) -> bool { return q < n && (a.isa[q] - s) < x; };
    uint32_t kf = 0, kb = 0;
    {
      uint64_t k = 1;
      while (k <= x && member((uint64_t)q0 + k * p)) { kf = (uint32_t)k; k <<= 1; }
      uint64_t hi_k = k;                                     // not a row (or beyond the text, or more steps than rows)
      while (hi_k - kf > 1) { const uint64_t mid = (kf + hi_k) >> 1; if (member((uint64_t)q0 + mid * p)) kf = (uint32_t)mid; else hi_k = mid; }
    }
    {
      uint64_t k = 1;
      while (k <= x && k * p <= q0 && member((uint64_t)q0 - k * p)) { kb = (uint32_t)k; k <<= 1; }
      uint64_t hi_k = k;
      while (hi_k - kb > 1) { const uint64_t mid = (kb + hi_k) >> 1; if (mid * p <= q0 && member((uint64_t)q0 - mid * p)) kb = (uint32_t)mid; else hi_k = mid; }
    }
    const uint32_t S0 = q0 - kb * p, E0 = q0 + kf * p;
    bool first = true;                                       // the first lane of every distinct progression keeps it
    for (uint32_t j = 0; j < 64u; ++j) { const uint32_t sj = (uint32_t)__shfl((int)S0, (int)j); if (j < lane && sj == S0) first = false; }
    const uint64_t bf = __ballot(first);
    const uint32_t nr0 = (uint32_t)__popcll(bf);
    const uint32_t rows = wave_sum_u32(first ? kf + kb + 1u : 0u);
    if (nr0 >= 2u && nr0 <= KD_REGIONS && rows == x) {
      if (first) { const uint32_t r = (uint32_t)__popcll(bf & lt); R->tS[r] = S0; R->tE[r] = E0; }
      nS = nE = nr0;
      fast = true;
    }
    if (a.dbg && lane == 0) atomicAdd(&a.dctl->dbg_hist[fast ? 26 : 27], 1u);
  }
  if (!fast) {
    const uint32_t nw = (job_helpers && x >= KD_HELP_MINX) ? (uint32_t)KD_HELP : 1u;
    if (lane == 0) { job->s = s; job->x = x; job->p = p; job->nS = 0; job->nE = 0; job->fail = 0; job->done = 0; }
    wsync();
    if (nw > 1u) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_store(&job->seq, job->seq + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    stairs_scan_share(a, R, job, 0u, nw, lane);
    if (nw > 1u) {
      while (__hip_atomic_load(&job->done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < nw - 1u) __builtin_amdgcn_s_sleep(2);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    wsync();
    if (__hip_atomic_load(&job->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return 0;
    nS = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&job->nS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    nE = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&job->nE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
  }
  if (a.dbg && lane == 0) atomicAdd(&a.dctl->dbg_hist[28], (uint32_t)((clock64() - tz0) >> 10));
  if (nS != nE || nS < 2) return 0;
  const uint32_t nr = nS;
  const uint64_t tz1 = a.dbg ? clock64() : 0;
  wsync();
  if (lane < nr) {                                            // sort both lists (distinct values): rank by counting
    uint32_t rs = 0, re = 0;
    const uint32_t vs = R->tS[lane], ve = R->tE[lane];
    for (uint32_t j = 0; j < nr; ++j) { rs += R->tS[j] < vs ? 1u : 0u; re += R->tE[j] < ve ? 1u : 0u; }
    R->S[rs] = vs; R->E[re] = ve;
  }
  wsync();
  {
    bool bad = false;
    uint32_t xr = 0;
    if (lane < nr) {
      const uint32_t lo = R->S[lane], hi = R->E[lane];
      bad = hi < lo || (hi - lo) % p != 0 || (lane + 1u < nr && hi >= R->S[lane + 1u]) || (uint64_t)hi + p > n;
      if (!bad) { xr = (hi - lo) / p + 1u; bad = xr < 3u; }
      R->X[lane] = xr;
    }
    const uint32_t tot = wave_sum_u32(xr);
    if (__any(bad) || tot != x) return 0;
  }
  wsync();
  // periodicity of every region, the bytes before its start, and the same pattern in all regions
  const uint32_t hi0 = R->E[0];
  for (uint32_t r = 0; r < nr; ++r) {
    const uint32_t lo = R->S[r], hi = R->E[r], span = hi - lo;
    const uint32_t lce = lce_back_wave(a.text, n, hi, hi + p, hi, lane);    // down to position 0, no wrap
    if (lce < span) return 0;
    if (r && lce_back_wave(a.text, n, hi0, hi, p, lane) < p) return 0;
    const uint32_t as = lo - (lce - span);
    const uint32_t d = as ? a.text[as - 1u] : a.text[n - 1u];
    if (d == a.text[as + p - 1u]) return 0;                   // (only when as == 0: the period goes on around the end)
    if (lane == 0) { R->Bv[r] = lce - span; R->D[r] = d; }
  }
  wsync();
  if (a.dbg && lane == 0) atomicAdd(&a.dctl->dbg_hist[29], (uint32_t)((clock64() - tz1) >> 10));
  const uint64_t tz2 = a.dbg ? clock64() : 0;
  // sides: every row is on the side of region 0's first row except, possibly, the regions' last rows
  const uint32_t beta = (a.isa[R->S[0]] - s >= x0) ? 1u : 0u;
  uint32_t nbar = 0;
  for (uint32_t r = 0; r < nr; ++r) nbar += ((a.isa[R->E[r]] - s >= x0) ? 1u : 0u) != beta ? 1u : 0u;
  if (nbar == 0 || nbar != (beta ? x0 : x1)) return 0;
  // the stretch: events at depths below tstop
  uint32_t tstop = 0xFFFFFFFFu;
  for (uint32_t r = 0; r < nr; ++r) { const uint32_t t = R->Bv[r] + (R->X[r] - 2u) * p; tstop = t < tstop ? t : tstop; }
  if (tstop == 0) return 0;
  uint32_t etot = 0;
  for (uint32_t r = 0; r < nr; ++r) {
    uint32_t e = 0;
    if (tstop > R->Bv[r]) { e = (tstop - R->Bv[r] + p - 1u) / p; const uint32_t cap = R->X[r] - 2u; e = e < cap ? e : cap; }
    if (lane == 0) R->PE[r] = etot;
    etot += e;
  }
  if (lane == 0) R->PE[nr] = etot;
  wsync();
  uint32_t base = 0;
  if (etot) {
    if (lane == 0) base = atomicAdd(&a.dctl->nsym, etot);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if ((uint64_t)base + etot > a.symcap) { a.dctl->err = 2; return 2; }
  }
  if (lane == 0) {
    R->kind = 2; R->p = p; R->x0 = x0; R->x1 = x1; R->base = base; R->events = etot; R->round = round;
    R->nr = nr; R->beta = beta; R->nbar = nbar; R->hi0 = hi0;
  }
  wsync();
  if (!stair_defer(a, *R, lane, true))
    for (uint32_t e = lane; e < etot; e += 64) stairs_event(a, *R, e);
  {
    uint32_t xall, smin, nl, ld[KD_REGIONS], le[KD_REGIONS];
    stairs_state(a, *R, tstop, xall, smin, nl, ld, le);
    const uint32_t xb = xall - nbar;
    next = DNode{smin, beta ? nbar : xb, beta ? xb : nbar, 0u, round + 8ull * tstop};
  }
  nodes_out = 8ull * tstop;
  if (a.dbg && lane == 0) { atomicAdd(&a.dctl->dbg_stairs, 1u << 16); atomicAdd(&a.dctl->dbg_stairsyms, etot); atomicAdd(&a.dctl->dbg_hist[30], (uint32_t)((clock64() - tz2) >> 10)); }
  return 1;
}

// ------------------------------------------------------------------------------------------------------
// Spine burst (wave-walkers).  A node of many rows most of which are preceded by the same byte c (the rows inside long
// runs of zeros of a binary: hundreds of thousands of rows, a few hundred of which leave per byte) is a chain that can
// neither be skipped nor expanded in closed form, and walking it is 8 dependent nodes per byte, ~1.5 us each, for
// thousands of bytes.  But the node of the chain one byte further down is known without the eight levels between: the
// wavelet levels of one byte are a stable LSD sort, so it is rows LF(first c-row) .. LF(last c-row), split at
// LF(first c-row of the x1 half), and LF(r) = ISA[SA[r] - 1].  While the three boundary rows are themselves preceded
// by c -- they sit inside runs of c, so for as many bytes as the shortest of the three runs still has -- the next
// boundary rows are again their images: LEVEL j of the chain is three ISA reads at text positions P - j, all levels at
// once, one lane each.  Then the 64 lanes walk the eight planes of THEIR level side by side: symbols as usual, the child on
// c's side is the next plane's chain node, the other child (the rows that leave) is queued for the next pass.
// Everything is exact: the chain child exists at every plane because it has at least the rows of the byte-level node
// below it on both sides of the split, which has at least one each; the walked level must end at the next level's node
// (checked; a mismatch abandons the tail, err 8).
// ------------------------------------------------------------------------------------------------------
#ifndef KD_SPINE_MIN_VALUE
#define KD_SPINE_MIN_VALUE 64
#endif
constexpr uint32_t KD_SPINE_MIN = KD_SPINE_MIN_VALUE;   // rows from which a wave-walker at plane 0 tries a burst
constexpr uint32_t KD_SPINE_LEVELS = 64;         // byte levels per burst: one lane each
struct SpineLds {
  uint32_t s[KD_SPINE_LEVELS + 1], x0[KD_SPINE_LEVELS + 1], x1[KD_SPINE_LEVELS + 1];   // the chain's node at plane 0, level by level
  uint32_t c[KD_SPINE_LEVELS];                   // the byte that leads from level i to level i + 1
  uint32_t pA[64], pB[64], pE[64];               // text positions of the boundary candidates of the current look
  uint8_t own[64];
  DNode side[8 * KD_SPINE_LEVELS];
};

// Whole wave, uniform arguments.  Returns the number of byte levels done (0: not a spine, nothing was changed; the node
// 8 * levels rounds further down is then in `next`), or 0xFFFFFFFF after an error (err is set).
__device__ __forceinline__ uint32_t spine_burst(const DfsArgs &a, SpineLds *S, uint32_t s, uint32_t x0, uint32_t x1, uint64_t round,
                                                uint32_t lane, DNode &next) {
  const K3Args &k = a.k;
  const uint32_t n = k.n;
  auto uni = [](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
  auto rdl = [](uint32_t v, int l) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); };
  const uint64_t lt = (1ull << lane) - 1ull;
  uint32_t L = 0, cs = s, cx0 = x0, cx1 = x1;
  const uint64_t tph0 = a.dbg ? clock64() : 0;
  if (lane == 0) { S->s[0] = s; S->x0[0] = x0; S->x1[0] = x1; }
  // number of leading bytes (going backwards from p, capped at 64) on which position p agrees with position pm, whose 64
  // bytes are in `mid` when pm >= 64 (loaded once per look: every candidate is compared with the same middle row).  16 bytes
  // per load: a look was ~200 four-byte loads per lane and bound by issuing them (58 K cycles per burst on the binary corpus).
  auto load16 = [&](uint32_t pos) -> uint4 { uint4 v; __builtin_memcpy(&v, a.text + pos, 16); return v; };
  auto agree = [&](uint32_t p, uint32_t pm, const uint4 *mid) -> uint32_t {
    if (p >= 64u && pm >= 64u) {
      uint32_t m = 0;
#pragma unroll
      for (uint32_t kq = 0; kq < 4u; ++kq) {
        const uint4 v = load16(p - 16u * kq - 16u);
        // (little-endian: the nearest byte is the most significant one of the highest word)
        const uint32_t d3 = v.w ^ mid[kq].w, d2 = v.z ^ mid[kq].z, d1 = v.y ^ mid[kq].y, d0 = v.x ^ mid[kq].x;
        if (m == 16u * kq) m += d3 ? (uint32_t)__clz((int)d3) >> 3 : 4u;
        if (m == 16u * kq + 4u) m += d2 ? (uint32_t)__clz((int)d2) >> 3 : 4u;
        if (m == 16u * kq + 8u) m += d1 ? (uint32_t)__clz((int)d1) >> 3 : 4u;
        if (m == 16u * kq + 12u) m += d0 ? (uint32_t)__clz((int)d0) >> 3 : 4u;
      }
      return m;
    }
    uint32_t m = 0;
    while (m < 64u && a.text[cyc_back(p, m + 1u, n)] == a.text[cyc_back(pm, m + 1u, n)]) ++m;
    return m;
  };
  auto wave_max = [](uint32_t v) -> uint32_t {
#pragma unroll
    for (int o = 32; o; o >>= 1) { const uint32_t w = (uint32_t)__shfl_xor((int)v, o); v = w > v ? w : v; }
    return v;
  };
  // lane <- the first lane i (in lane order) with m_i > lane, i.e. the first candidate that is still in the chain lane + 1
  // levels down (64 = none): the lanes publish the level ranges they are the first for (prefix maximum), then look up
  auto first_alive = [&](uint32_t m, volatile uint8_t *own) -> uint32_t {
    uint32_t pm = m;                                           // inclusive prefix max
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)pm, o); if (lane >= (uint32_t)o && t > pm) pm = t; }
    uint32_t before = (uint32_t)__shfl_up((int)pm, 1);
    if (lane == 0) before = 0;
    own[lane] = 64;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t j = before; j < m; ++j) own[j] = (uint8_t)lane;   // levels before + 1 .. m: I am the first candidate alive
    __builtin_amdgcn_wave_barrier();
    const uint32_t r = own[lane];
    __builtin_amdgcn_wave_barrier();
    return r;
  };
  while (L < KD_SPINE_LEVELS) {
    const uint32_t x = cx0 + cx1;
    if (x < KD_SPINE_MIN) { if (a.dbg && lane == 0) atomicAdd(&a.dctl->dbg_why[0], 1u); break; }
    if (a.dbg && lane == 0) { atomicAdd(&a.dctl->dbg_why[6], 1u); if (cx1 < 4u) atomicAdd(&a.dctl->dbg_why[7], 1u); }
    // The chain follows the MIDDLE row of the larger half: c_j = the byte j + 1 positions before it.  A row of the node
    // is in the chain's node j + 1 levels down iff the j + 1 bytes before it are c_0 .. c_j, and the images keep their
    // order, so that node is [LF^(j+1)(first such row), LF^(j+1)(last such row)], split at the image of the first such
    // row of the x1 half -- LF^(j+1) of a row at text position p being the row of position p - (j + 1).  The first / last
    // such rows are looked for among 64 CANDIDATES each (the first rows of either half, the last rows of the node): a
    // candidate knows from the 64 bytes before it for how many levels it stays, and per level the first candidate that
    // is still there is the boundary.  (The rows at the front of a context of zero runs are the run starts: each
    // stays for as many levels as it has zeros before it, one more than the row before it -- 64 levels from 64 rows.)
    const uint32_t rmid = cx0 >= cx1 ? cs + cx0 / 2u : cs + cx0 + cx1 / 2u;
    const bool vA = lane < cx0, vB = lane < cx1;
    const uint32_t pM = uni(a.sa[rmid]);
    const uint32_t pA = vA ? a.sa[cs + lane] : 0u, pB = vB ? a.sa[cs + cx0 + lane] : 0u, pE = vB ? a.sa[cs + x - 1u - lane] : 0u;
    uint4 mid[4] = {};
    if (pM >= 64u) {
#pragma unroll
      for (uint32_t kq = 0; kq < 4u; ++kq) mid[kq] = load16(pM - 16u * kq - 16u);
    }
    const uint32_t mA = vA ? agree(pA, pM, mid) : 0u, mB = vB ? agree(pB, pM, mid) : 0u, mE = vB ? agree(pE, pM, mid) : 0u;
    const uint32_t room = KD_SPINE_LEVELS - L;
    uint32_t J = wave_max(mA);
    { const uint32_t jb = wave_max(mB), je = wave_max(mE); J = J < jb ? J : jb; J = J < je ? J : je; J = J < room ? J : room; }
    if (J == 0) { if (a.dbg && lane == 0) atomicAdd(&a.dctl->dbg_why[1], 1u); break; }
    S->pA[lane] = pA; S->pB[lane] = pB; S->pE[lane] = pE;
    const uint32_t fa = first_alive(mA, S->own), fb = first_alive(mB, S->own), fe = first_alive(mE, S->own);
    uint32_t ia = 0, ib = 0, ie = 0, cj = 0;
    if (lane < J) {                                            // level L + 1 + lane
      ia = a.isa[cyc_back(S->pA[fa], lane + 1u, n)];
      ib = a.isa[cyc_back(S->pB[fb], lane + 1u, n)];
      ie = a.isa[cyc_back(S->pE[fe], lane + 1u, n)] + 1u;
      cj = a.text[cyc_back(pM, lane + 1u, n)];
    }
    // stop before a level at which the chain keeps less than half of its rows: the middle row has left the main branch
    // (its run has ended); the next look takes a new middle row.  At least one level is taken.
    const uint32_t xj = ie - ia;
    uint32_t xprev = (uint32_t)__shfl_up((int)xj, 1);
    if (lane == 0) xprev = x;
    const uint64_t thin = __ballot(lane < J && lane > 0 && 2u * xj < xprev);
    uint32_t cnt = thin ? (uint32_t)__ffsll((long long)thin) - 1u : J;
    if (cnt == 0) cnt = 1;
    if (lane < cnt) { S->c[L + lane] = cj; S->s[L + 1u + lane] = ia; S->x0[L + 1u + lane] = ib - ia; S->x1[L + 1u + lane] = ie - ib; }
    if (a.dbg && lane == 0) atomicAdd(&a.dctl->dbg_cont[cnt < J ? 0 : 3], 1u);
    cs = rdl(ia, (int)cnt - 1); cx0 = rdl(ib, (int)cnt - 1) - cs; cx1 = rdl(ie, (int)cnt - 1) - cs - cx0;
    L += cnt;
    __builtin_amdgcn_wave_barrier();
  }
  if (L == 0) return 0;
  wsync();
  const uint64_t tph1 = a.dbg ? clock64() : 0;
  if (a.dbg && lane == 0) atomicAdd(&a.dctl->dbg_why[4], (uint32_t)((tph1 - tph0) >> 10));
  // ---- the eight planes of every level, one lane per level ----
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(&a.dctl->nsym, 8u * L);
  base = uni(base);
  if ((uint64_t)base + 8u * L > a.symcap) { a.dctl->err = 2; return 0xFFFFFFFFu; }
  const bool act = lane < L;
  Node nd{act ? S->s[lane] : 0u, act ? S->x0[lane] : 1u, act ? S->x1[lane] : 1u};
  const uint32_t c = act ? S->c[lane] : 0u;
  uint32_t used = 0, nside = 0;
#pragma unroll 1
  for (uint32_t q = 0; q < 8u; ++q) {
    const Granule *G = k.gran + (size_t)q * k.ngran;
    const uint32_t ga = div96(nd.s), gb = div96(nd.s + nd.x0 + nd.x1), gm = div96(nd.s + nd.x0);
    const Granule ua = G[ga], ub = G[gb], um = G[gm];
    NodeFlat nf;
    node_flat_pre(nd, granule_rank1(ua, nd.s - ga * 96u), granule_rank1(ub, nd.s + nd.x0 + nd.x1 - gb * 96u), nf);
    uint32_t has0, has1, sym, kq;
    Node c0, c1;
    node_flat_post(nd, k.zeros[q], nf, granule_rank1(um, nd.s + nd.x0 - gm * 96u), has0, c0, has1, c1, sym, kq);
    const uint64_t rr = round + 8ull * lane + q;
    const bool emit = act && nf.need_mid;
    const uint64_t be = __ballot(emit);
    if (emit) {
      const uint32_t i = base + used + (uint32_t)__popcll(be & lt);
      uint32_t kw, ew;
      pack_symbol(k.cfg[q], q, sym, kq, nf.n0x, nd.x1, nd.x0 + nd.x1, kw, ew);
      a.tkey[i] = kw; a.tesc[i] = ew; a.ts[i] = nd.s;
      raw_symbol(a, i, sym, kq, nf.n0x, nd.x1, nd.x0 + nd.x1);
      a.trlo[i] = (uint32_t)rr;
      a.trhi[i] = (uint32_t)(rr >> 32) | (q << 8);
    }
    if (be && lane == 0) atomicAdd(&a.dctl->cntp[q], (uint32_t)__popcll(be));
    used += (uint32_t)__popcll(be);
    const uint32_t bit = (c >> q) & 1u;
    const bool hside = act && (bit ? has0 : has1);
    const uint64_t bh = __ballot(hside);
    if (hside) {
      const Node sd = bit ? c0 : c1;
      S->side[nside + (uint32_t)__popcll(bh & lt)] = DNode{sd.s, sd.x0, sd.x1, (q + 1u) & 7u, rr + 1ull};
    }
    nside += (uint32_t)__popcll(bh);
    nd = bit ? c1 : c0;
  }
  // the level must have arrived at the next level's node
  const bool bad = act && (nd.s != S->s[lane + 1u] || nd.x0 != S->x0[lane + 1u] || nd.x1 != S->x1[lane + 1u]);
  if (__any(bad)) { a.dctl->err = 8; return 0xFFFFFFFFu; }
  for (uint32_t j = used + lane; j < 8u * L; j += 64u) { a.ts[base + j] = 0; a.trlo[base + j] = 0; a.trhi[base + j] = KD_HOLE; }
  wsync();
  if (nside) {
    uint32_t o = 0;
    if (lane == 0) o = atomicAdd(&a.dctl->queued, nside);
    o = uni(o);
    if (o + nside > a.out_cap) { a.dctl->err = 4; return 0xFFFFFFFFu; }
    for (uint32_t j = lane; j < nside; j += 64u) a.out[o + j] = S->side[j];
  }
  if (lane == 0) atomicAdd(&a.dctl->spine_levels, L);
  if (a.dbg && lane == 0) atomicAdd(&a.dctl->dbg_spine[3], (uint32_t)((clock64() - tph1) >> 10));
  next = DNode{S->s[L], S->x0[L], S->x1[L], 0u, round + 8ull * L};
  return L;
}

// One pass of the walkers.  Lane = walker (UNI = false), or WAVE = walker (UNI = true, few walkers left: every value
// of the walk then depends on blockIdx only, so the compiler keeps it in scalar registers and runs the ~300
// instructions of a node on the scalar unit instead of issuing them for 64 lanes of which one works; the lanes
// are still there for the cooperative text comparisons of the chain skip).  A walker takes queued nodes gid, gid + W, ... and walks each subtree
// depth-first; after a.budget classified nodes it hands everything it still holds (current node, stack, queued
// nodes not started) to the next pass, where each of those gets a walker of its own: that is the load balancing.
template <bool UNI>
__global__ __launch_bounds__(UNI ? KD_T * KD_HELP : KD_T) void k3_dfs_kernel(DfsArgs a) {
  __shared__ DNode lstack[UNI ? KD_STACK : 1];
  __shared__ StairRegs sregs;
  __shared__ ScanJob sjob;
  const bool helpers = UNI && blockDim.x > (unsigned)KD_T;
  if (UNI && threadIdx.x == 0) sjob.seq = 0;
  if (helpers) {
    __syncthreads();                                         // (the only workgroup barrier of this kernel: before anything diverges)
    if (threadIdx.x >= (unsigned)KD_T) { stairs_helper(a, &sregs, &sjob, threadIdx.x >> 6, threadIdx.x & 63u); return; }
  }
  // wave 0 walks; on every way out it sends the helpers home
  struct Bye { ScanJob *j; bool on; __device__ ~Bye() { if (on && (threadIdx.x & 63u) == 0) __hip_atomic_store(&j->seq, 0xFFFFFFFFu, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); } } bye{&sjob, helpers};
  __shared__ __attribute__((aligned(8))) uint32_t spine_raw[UNI ? sizeof(SpineLds) / 4 : 1];
  SpineLds *const spine = reinterpret_cast<SpineLds *>(spine_raw);
  const K3Args &k = a.k;
  const EnumCtl *ctl = k.ctl;
  const uint32_t lane = threadIdx.x & 63u;
  // (lane-walkers: only the first a.lpw lanes of a wave walk -- a wave serves its walkers' chain comparisons one after
  //  the other, so when there are fewer walkers than lanes on the chip they are spread over more waves)
  const uint32_t gid = UNI ? blockIdx.x : blockIdx.x * a.lpw + threadIdx.x;
  const uint32_t W = UNI ? gridDim.x : gridDim.x * a.lpw;
  const bool writer = !UNI || lane == 0;                     // side effects of a wave-walker: one lane
  // A value a vector load returned is "divergent" to the compiler even when every lane loaded the same address:
  // readfirstlane moves it to a scalar register, and everything computed from it follows.
  auto uni = [&](uint32_t v) -> uint32_t { return UNI ? (uint32_t)__builtin_amdgcn_readfirstlane((int)v) : v; };
  auto uni_node = [&](const DNode &d) -> DNode {
    return DNode{uni(d.s), uni(d.x0), uni(d.x1), uni(d.plane), ((uint64_t)uni((uint32_t)(d.round >> 32)) << 32) | uni((uint32_t)d.round)};
  };
  // an atomic whose result every lane of a wave-walker needs
  auto uni_add = [&](uint32_t *ptr, uint32_t v) -> uint32_t {
    if (!UNI) return atomicAdd(ptr, v);
    uint32_t r = 0;
    if (lane == 0) r = atomicAdd(ptr, v);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
  };
  const uint32_t n = k.n;
  uint32_t pbase[9];
  {
    uint32_t acc = 0;
#pragma unroll
    for (uint32_t p = 0; p < 8; ++p) { pbase[p] = acc; acc += ctl->cnt[k.par][p][0] + ctl->cnt[k.par][p][1]; }
    pbase[8] = acc;
  }
  auto fetch = [&](uint32_t q) -> DNode {                     // queued node q of this pass
    if (a.in) return uni_node(a.in[q]);
    uint32_t p = 0;
#pragma unroll
    for (uint32_t j = 1; j < 8; ++j) p += q >= pbase[j] ? 1u : 0u;
    const uint32_t idx = q - pbase[p], c0 = ctl->cnt[k.par][p][0];
    const Node nd = plane_nodes(k, k.par, p)[idx < c0 ? idx : (list_cap(k, k.par) - 1u - (idx - c0))];
    return uni_node(DNode{nd.s, nd.x0, nd.x1, p, a.round0});
  };
  uint32_t next = gid;                                        // my next queued node
  bool alive = next < a.in_count && (UNI || lane < a.lpw);
  DNode cur{0u, 1u, 1u, 1u, 0ull};
  if (alive) { cur = fetch(next); next += W; }
  DNode *stack = UNI ? lstack : a.stacks + (size_t)gid * KD_STACK;     // a wave-walker's stack: LDS, every lane writes the same
  uint32_t sp = 0;
  uint64_t nodes = 0, maxround = 0;
  uint32_t visited = 0;
  uint32_t quiet = 8;                                         // pass-through nodes in a row (8 = one whole byte)
  constexpr uint32_t SBLK = UNI ? 8u * KD_SBLK : KD_SBLK;     // (a wave-walker waits for the atomic's answer with nothing else to do: fewer, larger blocks)
  uint32_t sbase = 0, sused = SBLK;                           // my block of tagged-symbol slots (none yet)
  uint32_t seen_err = 0;
  uint32_t skip_next = 0, skip_wait = 64u;                    // chain skip: back-off while it does not pay
  uint32_t stair_next = 0, stair_wait = KD_STAIR_RETRY;       // visited count from which I look for a staircase again; back-off
  uint32_t spine_next = 0, spine_wait = 16u;                  // the same for spine bursts
  auto pop = [&]() {                                          // this subtree is finished: the next pending one
    if (sp) { cur = uni_node(stack[--sp]); quiet = 8; }
    else if (next < a.in_count) { cur = fetch(next); next += W; quiet = 8; }
    else alive = false;
  };
  // All 64 lanes stay in the loop until every walker of the wave is finished: finished lanes help with the
  // cooperative text comparisons.
  uint64_t cyc_skip = 0, cyc_stair = 0, n_look = 0;
  uint64_t g_n = 0, g_c = 0, f_n = 0, f_c = 0;
  const uint64_t cyc0 = a.dbg ? clock64() : 0;
  while (__any(alive)) {
    if ((visited & 31u) == 0) seen_err = uni(a.dctl->err);          // a long chain should not wait for this load on every node
    if (alive && (visited >= (UNI ? a.budget : a.lane_budget) || seen_err)) {
      // hand on: current node, stack, queued nodes not started
      uint32_t rest = 0;
      if (next < a.in_count) rest = (a.in_count - next + W - 1u) / W;
      const uint32_t cnt = 1u + sp + rest;
      const uint32_t o = uni_add(&a.dctl->queued, cnt);
      if (o + cnt > a.out_cap) { a.dctl->err = 4; }
      else if (writer) {
        a.out[o] = cur;
        for (uint32_t j = 0; j < sp; ++j) a.out[o + 1u + j] = stack[j];
        for (uint32_t j = 0; j < rest; ++j) a.out[o + 1u + sp + j] = fetch(next + j * W);
      }
      alive = false;
    }
    // ---- chain skip: lanes at plane 0 after a whole byte of pass-through, served one after the other by the wave ----
    const uint32_t x = cur.x0 + cur.x1;
    uint64_t want = __ballot(alive && a.skip_ok && cur.plane == 0 && x <= KD_MAXX && quiet >= 8u && visited >= skip_next);
    if (UNI) want &= 1ull;                                     // one walker per wave: one request
    uint32_t mykk = 0;
    const uint64_t c1 = (a.dbg && want) ? clock64() : 0;
    while (want) {
      const int L = __ffsll((long long)want) - 1;
      want &= want - 1;
      const uint32_t sL = (uint32_t)__builtin_amdgcn_readlane((int)cur.s, L);
      const uint32_t xL = (uint32_t)__builtin_amdgcn_readlane((int)x, L);
      const uint64_t tq = UNI ? 0 : clock64();
      const uint32_t kk = uni(chain_bytes(a, sL, xL, lane));
      if (UNI || (int)lane == L) {
        // lane-walkers: the whole wave worked for this one; long comparisons count against its budget, so that it
        // moves on to a pass where it has a wave of its own (~2^12 cycles = one node of the walk)
        if (!UNI) visited += (uint32_t)((clock64() - tq) >> 12);
        if (a.dbg && !UNI) atomicMax(&a.dctl->dbg_slow, (((unsigned long long)(clock64() - tq) >> 10) << 32) | ((unsigned long long)xL << 20) | (kk < 0xFFFFFu ? kk : 0xFFFFFu));
        mykk = kk;
        if (!kk) quiet = 0;
        // comparing the rows costs ~x/64 batches of dependent loads: a chain that keeps yielding a byte or two for
        // that (rows of a periodic table: something leaves every period) is looked at less and less often
        if (8u * kk < xL / 8u) { skip_next = visited + skip_wait; skip_wait = skip_wait < 2048u ? skip_wait * 2u : skip_wait; }
        else skip_wait = 64u;
      }
    }
    if (c1) cyc_skip += clock64() - c1;
    // ---- staircase: a node of many rows at plane 0 whose rows are equally spaced text positions (stair_run) ----
    bool consumed = false;
    {
      const bool cand = alive && a.skip_ok && cur.plane == 0 && x >= KD_STAIR_MIN && visited >= stair_next && mykk == 0;
      bool single = false;
      if (cand) {                                              // cheap look by the walker itself: first, second and last row
        const uint32_t pa = a.sa[cur.s], pb = a.sa[cur.s + 1], pz = a.sa[cur.s + x - 1];
        const uint64_t span = (uint64_t)(x - 1) * (pb > pa ? pb - pa : pa - pb);
        single = pb > pa ? (uint64_t)pa + span == pz : (uint64_t)pz + span == pa;
      }
      uint64_t wants = __ballot(cand);
      const uint64_t singles = __ballot(single);
      if (UNI) wants &= 1ull;
      const uint64_t c2 = (a.dbg && wants) ? clock64() : 0;
      if (c2) n_look += (uint64_t)__popcll(wants);
      while (wants) {
        const int L = __ffsll((long long)wants) - 1;
        wants &= wants - 1;
        const uint32_t sL = (uint32_t)__builtin_amdgcn_readlane((int)cur.s, L);
        const uint32_t x0L = (uint32_t)__builtin_amdgcn_readlane((int)cur.x0, L), x1L = (uint32_t)__builtin_amdgcn_readlane((int)cur.x1, L);
        const uint64_t rL = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(cur.round >> 32), L) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)cur.round, L);
        uint64_t nn = 0, mr = 0;
        bool ok = false;
        uint32_t res = 0;
        DNode nx{0u, 1u, 1u, 0u, 0ull};
        const uint64_t tq = UNI ? 0 : clock64();
        if ((singles >> L) & 1ull) ok = uni(stair_run(a, sL, x0L, x1L, rL, lane, nn, mr) ? 1u : 0u) != 0u;
        if (!ok) res = uni(stairs_run(a, &sregs, &sjob, helpers, sL, x0L, x1L, rL, lane, nx, nn));
        if (UNI || (int)lane == L) {
          if (!UNI) visited += (uint32_t)((clock64() - tq) >> 12);
          if (ok) { consumed = true; nodes += nn; maxround = mr > maxround ? mr : maxround; }
          else if (res == 1u) {                                // the chain continues at the end of the stretch
            cur = uni_node(nx); nodes += nn; quiet = 0;
            stair_wait = KD_STAIR_RETRY; stair_next = visited + 16u;
          } else if (res == 2u) alive = false;                 // symbol buffer full: the tail is abandoned (err is set)
          else { stair_next = visited + stair_wait; stair_wait = stair_wait < 2048u ? stair_wait * 2u : stair_wait; }
        }
      }
      if (c2) cyc_stair += clock64() - c2;
    }
    // ---- spine burst: a wave-walker at plane 0 on a node of many rows that is neither a skip nor a staircase ----
    if (UNI && alive && !consumed && a.skip_ok && !a.no_spine && cur.plane == 0 && cur.x0 + cur.x1 >= KD_SPINE_MIN && mykk == 0 && visited >= spine_next) {
      DNode nx{0u, 1u, 1u, 0u, 0ull};
      const uint32_t ls = uni(spine_burst(a, spine, cur.s, cur.x0, cur.x1, cur.round, lane, nx));
      if (a.dbg && writer) { atomicAdd(&a.dctl->dbg_spine[0], 1u); if (ls && ls != 0xFFFFFFFFu) { atomicAdd(&a.dctl->dbg_spine[1], 1u); atomicAdd(&a.dctl->dbg_spine[2], ls); } }
      if (ls == 0xFFFFFFFFu) alive = false;
      else if (ls) {
        cur = uni_node(nx);
        nodes += 8ull * ls; visited += ls; quiet = 0;
        maxround = cur.round - 1ull > maxround ? cur.round - 1ull : maxround;
        spine_wait = 16u;
        continue;
      } else { spine_next = visited + spine_wait; spine_wait = spine_wait < 2048u ? spine_wait * 2u : spine_wait; }
    }
    if (alive && consumed) {
      ++visited;
      if (uni(a.dctl->err)) alive = false; else pop();
    } else if (alive) {
      ++visited;
      if (mykk) {                                    // mykk whole bytes of pass-through: 8*mykk rounds, no symbols
        const uint32_t pa = uni(a.sa[cur.s]);
        cur.s = uni(a.isa[cyc_back(pa, mykk, n)]);
        cur.round += 8ull * mykk;
        nodes += 8ull * mykk;
        if (a.dbg && writer) { atomicAdd(&a.dctl->dbg_skips, 1u); atomicAdd((unsigned long long *)&a.dctl->dbg_skipbytes, (unsigned long long)mykk); }
      }
      if (UNI) {
        // A wave-walker is on a long chain, and most of a chain is pass-through (all rows have the same bit): take those
        // nodes in a loop that does nothing else -- two ranks, one comparison -- up to the next byte boundary, where
        // the looks for a skip / a staircase above happen.
        uint32_t s = cur.s, p = cur.plane, steps = 0;
        const uint32_t xx = cur.x0 + cur.x1;
        const uint64_t tf = a.dbg ? clock64() : 0;
        do {
          // (the rank directory does not change during K3: constant address space + uniform index = scalar loads
          //  through the scalar cache, which a chain that moves a row per byte keeps hitting)
          const ConstGranule *G = (const ConstGranule *)(k.gran + (size_t)p * k.ngran);
          const uint32_t ga = div96(s), gb = div96(s + xx);
          const uint32_t zp = uni(k.zeros[p]);               // all three loads in flight together
          const Granule qa{G[ga].cum, G[ga].w0, G[ga].w1, G[ga].w2}, qb{G[gb].cum, G[gb].w0, G[gb].w1, G[gb].w2};
          const uint32_t rs = rank1_64(qa, s - ga * 96u), n1x = rank1_64(qb, s + xx - gb * 96u) - rs;
          if (n1x == 0) s -= rs;
          else if (n1x == xx) s = zp + rs;
          else break;
          p = (p + 1u) & 7u;
          ++steps;
        } while (p != 0);
        cur.s = s; cur.plane = p; cur.round += steps;
        nodes += steps; quiet += steps;
        if (a.dbg) { f_n += steps; f_c += clock64() - tf; }
        if (steps && p == 0) { visited += steps - 1u; continue; }
        visited += steps;
      }
      const uint32_t p = cur.plane;
      const uint64_t tg = (UNI && a.dbg) ? clock64() : 0;
      const Granule *G = k.gran + (size_t)p * k.ngran;
      auto ldg = [&](uint32_t g) -> Granule {                 // wave-walker: scalar load (see above)
        if (!UNI) return G[g];
        const ConstGranule *C = (const ConstGranule *)G;
        return Granule{C[g].cum, C[g].w0, C[g].w1, C[g].w2};
      };
      const Node nd{cur.s, cur.x0, cur.x1};
      const uint32_t ga = div96(nd.s), gb = div96(nd.s + nd.x0 + nd.x1), gm = div96(nd.s + nd.x0);
      const Granule qa = ldg(ga), qb = ldg(gb);
      // wave-walker: a node is a chain of dependent loads and nothing else runs in the wave, so the split point's granule
      // comes with the other two instead of in a round trip of its own when it turns out to be needed
      Granule qm = UNI ? ldg(gm) : qa;
      NodeFlat nf;
      node_flat_pre(nd, granule_rank1(qa, nd.s - ga * 96u), granule_rank1(qb, nd.s + nd.x0 + nd.x1 - gb * 96u), nf);
      if (!UNI) {
        qm = gm == ga ? qa : qb;
        if (nf.need_mid && gm != ga && gm != gb) qm = ldg(gm);
      }
      uint32_t has0, has1, sym, kq;
      Node c0, c1;
      node_flat_post(nd, uni(k.zeros[p]), nf, granule_rank1(qm, nd.s + nd.x0 - gm * 96u), has0, c0, has1, c1, sym, kq);
      ++nodes;
      if (a.dbg && writer) atomicAdd(&a.dctl->dbg_hist[31 - __clz((int)(nd.x0 + nd.x1))], 1u);
      maxround = cur.round > maxround ? cur.round : maxround;
      quiet = (nf.need_mid || (has0 && has1)) ? 0u : quiet + 1u;
      if (nf.need_mid) {
        if (sused == SBLK) { sbase = uni_add(&a.dctl->nsym, SBLK); sused = 0; }
        const uint32_t i = sbase + sused;
        if (sbase + SBLK > a.symcap) { a.dctl->err = 2; alive = false; sused = SBLK; }
        else {
          ++sused;
          uint32_t kw, ew;
          pack_symbol(k.cfg[p], p, sym, kq, nf.n0x, nd.x1, nd.x0 + nd.x1, kw, ew);
          if (writer) {
            a.tkey[i] = kw; a.tesc[i] = ew; a.ts[i] = nd.s;
            raw_symbol(a, i, sym, kq, nf.n0x, nd.x1, nd.x0 + nd.x1);
            a.trlo[i] = (uint32_t)cur.round;
            a.trhi[i] = (uint32_t)(cur.round >> 32) | (p << 8);      // round < 2^40
            atomicAdd(&a.dctl->cntp[p], 1u);
          }
        }
      }
      const uint32_t pn = (p + 1u) & 7u;
      if (has0 && has1) {
        if (sp >= KD_STACK) {                                // full: the older half becomes queued nodes of the next pass
          const uint32_t h = KD_STACK / 2;
          const uint32_t o = uni_add(&a.dctl->queued, h);
          if (o + h > a.out_cap) { a.dctl->err = 4; alive = false; }
          else {
            if (writer) for (uint32_t j = 0; j < h; ++j) a.out[o + j] = stack[j];
            for (uint32_t j = h; j < sp; ++j) stack[j - h] = stack[j];
            sp -= h;
          }
        }
        stack[sp++] = DNode{c1.s, c1.x0, c1.x1, pn, cur.round + 1};
      }
      if (has0) { cur = DNode{c0.s, c0.x0, c0.x1, pn, cur.round + 1}; }
      else if (has1) { cur = DNode{c1.s, c1.x0, c1.x1, pn, cur.round + 1}; }
      else pop();
      if (UNI && a.dbg) { ++g_n; g_c += clock64() - tg; }
    }
  }
  // the unused slots of my last block are holes: they sort behind every real symbol and are cut off by the host
  if (!writer) return;
  for (uint32_t j = sused; j < SBLK; ++j) { a.ts[sbase + j] = 0; a.trlo[sbase + j] = 0; a.trhi[sbase + j] = KD_HOLE; }
  if (a.dbg) atomicMax(&a.dctl->dbg_maxvis, visited);
  if (a.dbg && lane == 0) {
    atomicAdd((unsigned long long *)&a.dctl->dbg_cyc[0], (unsigned long long)(clock64() - cyc0));
    atomicAdd((unsigned long long *)&a.dctl->dbg_cyc[1], (unsigned long long)cyc_skip);
    atomicAdd((unsigned long long *)&a.dctl->dbg_cyc[2], (unsigned long long)cyc_stair);
    atomicAdd((unsigned long long *)&a.dctl->dbg_cyc[3], (unsigned long long)n_look);
    if (UNI) {
      atomicAdd((unsigned long long *)&a.dctl->dbg_gen[0], (unsigned long long)g_n);
      atomicAdd((unsigned long long *)&a.dctl->dbg_gen[1], (unsigned long long)g_c);
      atomicAdd((unsigned long long *)&a.dctl->dbg_gen[2], (unsigned long long)f_n);
      atomicAdd((unsigned long long *)&a.dctl->dbg_gen[3], (unsigned long long)f_c);
    }
    const unsigned long long tw = (unsigned long long)(clock64() - cyc0) >> 12, ts = cyc_skip >> 12, tt = cyc_stair >> 12;
    atomicMax(&a.dctl->dbg_maxwave, ((tw & 0xFFFFFull) << 40) | ((ts & 0xFFFFFull) << 20) | (tt & 0xFFFFFull));
  }
  atomicAdd((unsigned long long *)&a.dctl->nodes, (unsigned long long)nodes);
  atomicMax((unsigned long long *)&a.dctl->maxround, (unsigned long long)maxround);
}

// ------------------------------------------------------------------------------------------------------
// Workgroup-local rounds.  Between the wide rounds (millions of nodes, two launches each) and the walkers (<= 65 536
// nodes, one lane each) lie hundreds of rounds of 10^5..10^6 nodes: too few to hide a launch (a round costs 16-50 us
// whatever it holds) and too bushy for one lane per subtree.  Subtrees are independent and the round structure only
// orders the symbols, so here every workgroup takes LB_IN consecutive live nodes and runs THEIR rounds on its own:
// both list buffers in LDS, one classification per node (no count pass: the compaction is a block scan), no launch
// and no grid-wide step between rounds.  Symbols are tagged (plane, round, s) like the walkers' and sorted into stream
// order with theirs.  A workgroup stops after lb_budget rounds or when fewer than LB_MIN nodes are left and hands those
// back through a global queue that the host re-chunks into the next launch of this kernel (or gives to the walkers when
// few are left); children that do not fit the LDS list go the same way.  A child that arrives at plane 0 after lb_quiet
// pass-through levels can be sent straight to the walkers (a chain they could skip): off by default, see k3_dfs_tail.
// ------------------------------------------------------------------------------------------------------
// Geometry, measured on the natural corpus (K3 in all): ONE node per thread (63 VGPRs = 7 waves per SIMD), 256-node lists
// started with 128 / 160 / 192 / 224 / 256 nodes: 25.0 / 22.8 / 22.0 / 21.5 / 22.2 ms; three nodes per thread, lists of 736
// started half full (115 VGPRs, 4 waves): 25.7; two per thread: 26.7; 512 threads x 1: 21.9; 128 threads x 1: 28.7; one wave per
// workgroup: 11.4 instead of 4.4 ms for the first pass alone.
constexpr int LB_T = 256;
constexpr int LB_NPT = 1;                      // nodes per thread and round
constexpr uint32_t LB_CAP = LB_T * LB_NPT;     // nodes per LDS list
constexpr uint32_t LB_IN = LB_CAP - LB_CAP / 8;   // nodes a workgroup starts with
constexpr uint32_t LB_MIN = 48;                // fewer nodes than this: handed back (re-chunked with others, or the walkers; 24: 26.4 ms, 96: 25.0)
constexpr uint32_t LB_SBLK = 256;              // tagged-symbol slots a workgroup reserves at a time (what is left of a block at the end is holes)
constexpr uint32_t LB_XCAP = 64;               // nodes on their way to the walkers, buffered in LDS

struct LNode { uint32_t s, x0, x1, meta; };    // meta: [2:0] plane, [8:3] pass-through levels in a row (saturating), [31:9] round - round0
constexpr uint32_t LB_RSH = 9;

__device__ __forceinline__ DNode lnode_out(const DfsArgs &a, const LNode &l) {
  return DNode{l.s, l.x0, l.x1, l.meta & 7u, a.round0 + (uint64_t)(l.meta >> LB_RSH)};
}

__global__ __launch_bounds__(LB_T) void k3_local_kernel(DfsArgs a) {
  __shared__ LNode buf[2][LB_CAP];
  __shared__ LNode xbuf[LB_XCAP];
  __shared__ uint64_t ws[LB_T / 64];
  __shared__ uint32_t s_bc[4];                 // results of thread 0's atomics: symbol block, queue slot, spill slot, stop
  __shared__ uint32_t s_cntp[8], s_maxrel;
  const K3Args &k = a.k;
  const EnumCtl *ctl = k.ctl;
  const uint32_t tid = threadIdx.x;
  const uint32_t first = blockIdx.x * LB_IN;
  uint32_t total = a.in_count - first < LB_IN ? a.in_count - first : LB_IN;
  if (tid < 8) s_cntp[tid] = 0;
  if (tid == 8) s_maxrel = 0;
  if (a.in) {
    for (uint32_t q = tid; q < total; q += LB_T) {
      const DNode d = a.in[first + q];
      buf[0][q] = LNode{d.s, d.x0, d.x1, d.plane | ((uint32_t)(d.round - a.round0) << LB_RSH)};
    }
  } else {                                     // first pass: the planes' node lists, all at round0
    uint32_t pbase[9];
    uint32_t acc = 0;
#pragma unroll
    for (uint32_t p = 0; p < 8; ++p) { pbase[p] = acc; acc += ctl->cnt[k.par][p][0] + ctl->cnt[k.par][p][1]; }
    pbase[8] = acc;
    for (uint32_t q = tid; q < total; q += LB_T) {
      const uint32_t g = first + q;
      uint32_t p = 0;
#pragma unroll
      for (uint32_t j = 1; j < 8; ++j) p += g >= pbase[j] ? 1u : 0u;
      const uint32_t idx = g - pbase[p], c0 = ctl->cnt[k.par][p][0];
      const Node nd = plane_nodes(k, k.par, p)[idx < c0 ? idx : (list_cap(k, k.par) - 1u - (idx - c0))];
      buf[0][q] = LNode{nd.s, nd.x0, nd.x1, p};
    }
  }
  // replicated in every thread (all update them identically): my symbol block, nodes waiting in xbuf
  uint32_t sbase = 0, ssize = 0, sused = 0, xn = 0;
  uint32_t cur = 0, rounds = 0, maxrel = 0;
  uint64_t nodes = 0;
  bool abort = false;
  __syncthreads();
  for (;;) {
    if (total < LB_MIN || rounds >= a.lb_budget) break;
    if ((rounds & 15u) == 15u) {               // somebody ran out of room: the whole tail is abandoned
      if (tid == 0) s_bc[3] = a.dctl->err;
      __syncthreads();
      if (s_bc[3]) { abort = true; break; }
    }
    // ---- classify my nodes: all loads of a phase in flight together (as k3_classify) ----
    // (items beyond the list are skipped as a whole: a workgroup spends most of its rounds on a fraction of its capacity)
    const uint32_t nit = (total + LB_T - 1u) / LB_T;
    LNode nd[LB_NPT];
    bool valid[LB_NPT];
    uint32_t ga[LB_NPT], gb[LB_NPT], gm[LB_NPT];
    Granule qa[LB_NPT], qb[LB_NPT], qm[LB_NPT];
#pragma unroll
    for (int it = 0; it < LB_NPT; ++it) {
      const uint32_t q = (uint32_t)it * LB_T + tid;
      valid[it] = q < total;
      if ((uint32_t)it >= nit) continue;
      nd[it] = buf[cur][valid[it] ? q : 0u];
      const Granule *G = k.gran + (size_t)(nd[it].meta & 7u) * k.ngran;
      ga[it] = div96(nd[it].s);
      gb[it] = div96(nd[it].s + nd[it].x0 + nd[it].x1);
      gm[it] = div96(nd[it].s + nd[it].x0);
      qa[it] = G[ga[it]];
      qb[it] = G[gb[it]];
    }
    NodeFlat nf[LB_NPT];
#pragma unroll
    for (int it = 0; it < LB_NPT; ++it) {
      if ((uint32_t)it >= nit) continue;
      const Node n3{nd[it].s, nd[it].x0, nd[it].x1};
      node_flat_pre(n3, granule_rank1(qa[it], nd[it].s - ga[it] * 96u),
                    granule_rank1(qb[it], nd[it].s + nd[it].x0 + nd[it].x1 - gb[it] * 96u), nf[it]);
      qm[it] = gm[it] == ga[it] ? qa[it] : qb[it];      // (loading it unconditionally with the other two: 4.33 vs 4.44 ms, not worth the registers)
      if (valid[it] && nf[it].need_mid && gm[it] != ga[it] && gm[it] != gb[it])
        qm[it] = (k.gran + (size_t)(nd[it].meta & 7u) * k.ngran)[gm[it]];
    }
    uint32_t has0[LB_NPT], has1[LB_NPT], ex0[LB_NPT], ex1[LB_NPT], sym[LB_NPT], kq[LB_NPT], cmeta[LB_NPT];
    Node c0[LB_NPT], c1[LB_NPT];
    uint64_t mine = 0;                         // [15:0] children that stay, [31:16] symbols, [47:32] children for the walkers
#pragma unroll
    for (int it = 0; it < LB_NPT; ++it) {
      if ((uint32_t)it >= nit) { has0[it] = has1[it] = ex0[it] = ex1[it] = 0u; nf[it].need_mid = 0u; continue; }
      const Node n3{nd[it].s, nd[it].x0, nd[it].x1};
      const uint32_t p = nd[it].meta & 7u, pn = (p + 1u) & 7u, quiet = (nd[it].meta >> 3) & 63u, rel = nd[it].meta >> LB_RSH;
      node_flat_post(n3, k.zeros[p], nf[it], granule_rank1(qm[it], nd[it].s + nd[it].x0 - gm[it] * 96u), has0[it], c0[it],
                     has1[it], c1[it], sym[it], kq[it]);
      has0[it] = valid[it] ? has0[it] : 0u;
      has1[it] = valid[it] ? has1[it] : 0u;
      nf[it].need_mid = valid[it] ? nf[it].need_mid : 0u;
      const uint32_t cq = (nf[it].need_mid || (has0[it] && has1[it])) ? 0u : (quiet < 63u ? quiet + 1u : 63u);
      cmeta[it] = pn | (cq << 3) | ((rel + 1u) << LB_RSH);
      // a chain worth a walker's skip: at plane 0 after lb_quiet levels (whole bytes) without a symbol or a split
      const bool chain = a.skip_ok && pn == 0u && cq >= a.lb_quiet;
      ex0[it] = (has0[it] && chain && c0[it].x0 + c0[it].x1 <= KD_MAXX) ? 1u : 0u;
      ex1[it] = (has1[it] && chain && c1[it].x0 + c1[it].x1 <= KD_MAXX) ? 1u : 0u;
      if (valid[it]) maxrel = rel > maxrel ? rel : maxrel;
      mine += (uint64_t)(has0[it] - ex0[it] + has1[it] - ex1[it]) | ((uint64_t)nf[it].need_mid << 16) |
              ((uint64_t)(ex0[it] + ex1[it]) << 32);
    }
    // ---- one packed block scan ----
    uint64_t tot;
    uint64_t ex;
    {
      const uint32_t lane = tid & 63u, wid = tid >> 6;
      const uint64_t inc = wave_incl_sum64(mine);
      if (lane == 63) ws[wid] = inc;
      __syncthreads();
      uint64_t wbase = 0;
      tot = 0;
#pragma unroll
      for (int i = 0; i < LB_T / 64; ++i) { const uint64_t t = ws[i]; if ((uint32_t)i < wid) wbase += t; tot += t; }
      ex = wbase + inc - mine;
    }
    const uint32_t tch = (uint32_t)(tot & 0xFFFFu), tsy = (uint32_t)((tot >> 16) & 0xFFFFu), tex = (uint32_t)(tot >> 32);
    // ---- room: a new symbol block, the walkers' queue when xbuf is full, the spill queue when the list overflows ----
    const bool need_s = sused + tsy > ssize, need_x = xn + tex > LB_XCAP, need_p = tch > LB_CAP;
    uint32_t xdirect = 0;                      // this round's exits go straight to the queue, behind xbuf's content
    if (need_s || need_x || need_p) {
      if (need_s)                              // what is left of the old block becomes holes (sorted behind everything, cut off)
        for (uint32_t j = sbase + sused + tid; j < sbase + ssize; j += LB_T) { a.ts[j] = 0; a.trlo[j] = 0; a.trhi[j] = KD_HOLE; }
      if (tid == 0) {
        uint32_t stop = 0;
        if (need_s) {
          const uint32_t sz = tsy > LB_SBLK ? tsy : LB_SBLK;
          const uint32_t b = atomicAdd(&a.dctl->nsym, sz);
          if ((uint64_t)b + sz > a.symcap) { a.dctl->err = 2; stop = 1; }
          s_bc[0] = b;
        }
        if (need_x) {
          const uint32_t o = atomicAdd(&a.dctl->queued, xn + tex);
          if ((uint64_t)o + xn + tex > a.out_cap) { a.dctl->err = 4; stop = 1; }
          s_bc[1] = o;
        }
        if (need_p) {
          const uint32_t o = atomicAdd(&a.dctl->spilled, tch - LB_CAP);
          if ((uint64_t)o + (tch - LB_CAP) > a.spill_cap) { a.dctl->err = 4; stop = 1; }
          s_bc[2] = o;
        }
        s_bc[3] = stop;
      }
      __syncthreads();
      if (s_bc[3]) { abort = true; break; }
      if (need_s) { sbase = s_bc[0]; ssize = tsy > LB_SBLK ? tsy : LB_SBLK; sused = 0; }
      if (need_x) {
        for (uint32_t j = tid; j < xn; j += LB_T) a.out[s_bc[1] + j] = lnode_out(a, xbuf[j]);
        xdirect = s_bc[1] + xn + 1u;           // (+1: 0 means "into xbuf")
      }
    }
    const uint32_t spill0 = need_p ? s_bc[2] : 0u;
    // ---- place: children into the other list (any order), symbols into my block, exits ----
    {
      uint32_t rc = (uint32_t)(ex & 0xFFFFu), rs = (uint32_t)((ex >> 16) & 0xFFFFu), rx = (uint32_t)(ex >> 32);
#pragma unroll
      for (int it = 0; it < LB_NPT; ++it) {
        if ((uint32_t)it >= nit) continue;
        const uint32_t p = nd[it].meta & 7u;
#pragma unroll
        for (int side = 0; side < 2; ++side) {
          const uint32_t has = side ? has1[it] : has0[it], exq = side ? ex1[it] : ex0[it];
          if (!has) continue;
          const Node &c = side ? c1[it] : c0[it];
          const LNode l{c.s, c.x0, c.x1, cmeta[it]};
          if (exq) {
            if (xdirect) a.out[xdirect - 1u + rx] = lnode_out(a, l); else xbuf[xn + rx] = l;
            ++rx;
          } else {
            if (rc < LB_CAP) buf[cur ^ 1u][rc] = l; else a.spill[spill0 + rc - LB_CAP] = lnode_out(a, l);
            ++rc;
          }
        }
        if (nf[it].need_mid) {
          const uint32_t i = sbase + sused + rs;
          ++rs;
          uint32_t kw, ew;
          pack_symbol(k.cfg[p], p, sym[it], kq[it], nf[it].n0x, nd[it].x1, nd[it].x0 + nd[it].x1, kw, ew);
          const uint64_t r = a.round0 + (uint64_t)(nd[it].meta >> LB_RSH);
          a.tkey[i] = kw; a.tesc[i] = ew; a.ts[i] = nd[it].s;      // (five 4-byte stores: without them the pass is as long)
          raw_symbol(a, i, sym[it], kq[it], nf[it].n0x, nd[it].x1, nd[it].x0 + nd[it].x1);
          a.trlo[i] = (uint32_t)r;
          a.trhi[i] = (uint32_t)(r >> 32) | (p << 8);
          atomicAdd(&s_cntp[p], 1u);
        }
      }
    }
    sused += tsy;
    xn = xdirect ? 0u : xn + tex;
    nodes += total;
    total = tch < LB_CAP ? tch : LB_CAP;
    cur ^= 1u;
    ++rounds;
    __syncthreads();
  }
  if (abort) return;
  // ---- the chains in xbuf go to the walkers, the nodes that are left to the queue of the next local pass (the host
  //      decides whether there is one); the rest of my symbol block becomes holes ----
  __syncthreads();
  if (total + xn) {
    if (tid == 0) {
      uint32_t stop = 0, o = 0, o2 = 0;
      if (xn) { o = atomicAdd(&a.dctl->queued, xn); if ((uint64_t)o + xn > a.out_cap) stop = 1; }
      if (total) { o2 = atomicAdd(&a.dctl->spilled, total); if ((uint64_t)o2 + total > a.spill_cap) stop = 1; }
      if (stop) a.dctl->err = 4;
      s_bc[1] = o; s_bc[2] = o2; s_bc[3] = stop;
    }
    __syncthreads();
    if (!s_bc[3]) {
      for (uint32_t j = tid; j < xn; j += LB_T) a.out[s_bc[1] + j] = lnode_out(a, xbuf[j]);
      for (uint32_t j = tid; j < total; j += LB_T) a.spill[s_bc[2] + j] = lnode_out(a, buf[cur][j]);
    }
  }
  for (uint32_t j = sbase + sused + tid; j < sbase + ssize; j += LB_T) { a.ts[j] = 0; a.trlo[j] = 0; a.trhi[j] = KD_HOLE; }
  atomicMax(&s_maxrel, maxrel);
  __syncthreads();
  if (tid < 8 && s_cntp[tid]) atomicAdd(&a.dctl->cntp[tid], s_cntp[tid]);
  if (tid == 0 && nodes) {
    atomicAdd((unsigned long long *)&a.dctl->nodes, (unsigned long long)nodes);
    atomicMax((unsigned long long *)&a.dctl->maxround, (unsigned long long)(a.round0 + s_maxrel));
  }
}

// keys[i] = src[perm[i]]
__global__ void kd_gather_kernel(const uint32_t *__restrict__ src, const uint32_t *__restrict__ perm, uint32_t m,
                                 uint32_t *__restrict__ keys) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) keys[i] = src[perm[i]];
}
__global__ void kd_iota_kernel(const uint32_t *__restrict__ src, uint32_t m, uint32_t *__restrict__ keys,
                               uint32_t *__restrict__ vals) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) { keys[i] = src[i]; vals[i] = i; }
}
__global__ void kd_place_kernel(const uint32_t *__restrict__ tkey, const uint32_t *__restrict__ tesc,
                                const uint32_t *__restrict__ perm, uint32_t m, uint32_t *__restrict__ symkey,
                                uint32_t *__restrict__ symesc) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
    symkey[i] = tkey[perm[i]];
    symesc[i] = tesc[perm[i]];
  }
}

__global__ void kd_place_raw_kernel(const uint32_t *__restrict__ traw, const uint32_t *__restrict__ perm, uint32_t m, uint32_t *__restrict__ rec) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) rec[i] = traw[perm[i]];
}

// Host side.  The tail starts when the node count has stopped growing, at most `enter` nodes are alive and an
// eighth of all nodes has been visited (i.e. not in the ramp-up).  With more than KD_LOCAL_FROM nodes alive the
// workgroup-local rounds come first (k3_local_kernel: one launch over the planes' lists, further ones while it
// spills); what they hand on, or the planes' lists themselves, goes to the walkers, whose passes follow each other
// until no node is queued.  Preconditions (checked by the caller): the symbol buffer is empty (everything emitted so
// far has been flushed), `ctl` is current.  On success the symbol buffer holds the tail's symbols in stream order,
// run_log holds one run per plane, and *done = true.  On an error (more symbols than the tagged buffer holds, queue
// full) nothing has been changed and the caller continues with the round-based kernels.
constexpr uint32_t KD_LOCAL_FROM = 16384;       // live nodes above which the tail starts with workgroup-local rounds
constexpr uint32_t KD_LOCAL_QUEUE = 16u << 20;  // walker queue / spill queue capacity in that case
constexpr uint32_t KD_LOCAL_BUDGET = 192;       // rounds per workgroup and pass (BCE_HIP_LOCAL_BUDGET overrides; natural corpus, K3 in all: 96: 23.3 ms, 128: 21.4, 192: 21.2, 256: 21.7, 384: 21.9)
int k3_dfs_tail(bce_hip_ctx *c, const EnumCtl &ctl, uint32_t enter, bool *done) {
  *done = false;
  if (c->dbg_no_dfs || (c->scan_mode && getenv("BCE_HIP_SCAN_NO_DFS"))) return BCE_HIP_OK;
  const uint32_t n = c->n;
  if (ctl.next_nodes > enter) return BCE_HIP_OK;
  const uint32_t live = (uint32_t)ctl.next_nodes;
  const uint64_t all = 8ull * (n - 1);
  // (not in the ramp-up: an eighth of all nodes visited -- or a thousand rounds gone by with this few nodes alive,
  //  which is what an input that is one long run or one table looks like from the start)
  if (live == 0 || live > enter || (ctl.nodes_total < all / 8 && c->round < 1024u && !c->dbg_tail_round)) return BCE_HIP_OK;
  // tagged symbols: the tail is mostly pass-through, a fraction of the nodes that are left is plenty; if it is
  // not (chains in which every byte codes a symbol or two), the walk is repeated with four times the room
  const uint64_t left = all > ctl.nodes_total ? all - ctl.nodes_total : 0;
  uint32_t local_from = c->dbg_local_from ? c->dbg_local_from : KD_LOCAL_FROM;
  if (const char *e = getenv("BCE_HIP_LOCAL_FROM")) local_from = (uint32_t)strtoul(e, nullptr, 10);
  const bool local = live > local_from && !c->dbg_no_local;
  uint64_t cap_scale = 1;
retry:
  // (the bushy part the local rounds take codes more symbols per node than the chains of the deep tail)
  // (+ the blocks of LB_SBLK slots the workgroups of the local rounds reserve and leave partly unused)
  uint64_t cap64 = (left / (local ? 2 : 8) + (1u << 20)) * cap_scale + (local ? (uint64_t)LB_SBLK * 2 * ((live + LB_IN - 1) / LB_IN) : 0);
  if (cap64 < (4u << 20)) cap64 = 4u << 20;
  if (cap64 > (512u << 20)) cap64 = 512u << 20;
  if (cap64 > left + 64) cap64 = left + 64;                  // a node codes at most one symbol
  if (local && cap64 < (uint64_t)LB_SBLK * ((live + LB_IN - 1) / LB_IN + 1)) cap64 = (uint64_t)LB_SBLK * ((live + LB_IN - 1) / LB_IN + 1);   // one block per workgroup at least
  const uint32_t cap = ((uint32_t)cap64 + 63u) & ~63u;       // keeps the carved arrays 8-byte aligned
  const uint32_t qmax = local ? KD_LOCAL_QUEUE : KD_QUEUE;
  const uint32_t qcap = (uint32_t)(left + 64 < qmax ? left + 64 : qmax);      // queued nodes are distinct unvisited nodes
  const uint32_t wmax = qcap < KD_WALKERS ? qcap : KD_WALKERS;
  const uint32_t scap = local ? qcap : 0u;                   // spill queues of the local rounds
  // carve: tkey tesc ts trlo trhi | sort keys x2 vals x2 | DfsCtl | queue x2 | stacks | jobs | spill x2
  const size_t o_sort = (size_t)cap * 4 * 5, o_ctl = o_sort + (size_t)cap * 4 * 4, o_q = o_ctl + 512,
               o_stack = o_q + 2 * (size_t)qcap * sizeof(DNode), o_jobs = o_stack + (size_t)wmax * KD_STACK * sizeof(DNode),
               o_spill = o_jobs + (size_t)KD_JOBS * sizeof(StairJob);
  const size_t o_raw = o_spill + 2 * (size_t)scap * sizeof(DNode);
  BCE_TRY(ensure(c, c->dfs, o_raw + (c->scan_mode ? (size_t)cap * 4 : 0)));
  uint8_t *base = c->dfs.as<uint8_t>();
  DfsArgs a;
  a.k = k3_make_args(c, c->round, 0);
  a.text = c->text.as<uint8_t>();
  a.bwt = c->bwt.as<uint8_t>();
  a.sa = c->sa[c->sa_res].as<uint32_t>();
  a.isa = c->rank.as<uint32_t>();
  a.skip_ok = (!c->dbg_no_skip && c->k1_valid && c->text.p && c->rank.p && c->bwt.p) ? 1u : 0u;
  a.traw = c->scan_mode ? reinterpret_cast<uint32_t *>(base + o_raw) : nullptr;
  uint32_t *w = reinterpret_cast<uint32_t *>(base);
  a.tkey = w; a.tesc = w + cap; a.ts = w + 2 * (size_t)cap; a.trlo = w + 3 * (size_t)cap; a.trhi = w + 4 * (size_t)cap;
  uint32_t *sk[2] = {reinterpret_cast<uint32_t *>(base + o_sort), reinterpret_cast<uint32_t *>(base + o_sort) + cap};
  uint32_t *sv[2] = {reinterpret_cast<uint32_t *>(base + o_sort) + 2 * (size_t)cap, reinterpret_cast<uint32_t *>(base + o_sort) + 3 * (size_t)cap};
  a.dctl = reinterpret_cast<DfsCtl *>(base + o_ctl);
  DNode *queue[2] = {reinterpret_cast<DNode *>(base + o_q), reinterpret_cast<DNode *>(base + o_q) + qcap};
  DNode *spillq[2] = {reinterpret_cast<DNode *>(base + o_spill), reinterpret_cast<DNode *>(base + o_spill) + scap};
  a.stacks = reinterpret_cast<DNode *>(base + o_stack);
  a.jobs = reinterpret_cast<StairJob *>(base + o_jobs);
  a.round0 = c->round;
  a.symcap = cap;
  a.out_cap = qcap;
  a.spill = spillq[0]; a.spill_cap = scap;
  a.lb_budget = KD_LOCAL_BUDGET;
  if (const char *e = getenv("BCE_HIP_LOCAL_BUDGET")) a.lb_budget = (uint32_t)strtoul(e, nullptr, 10);
  if (c->dbg_local_budget) a.lb_budget = c->dbg_local_budget;
  a.lpw = KD_T;
  // chains leave the local rounds early only on request: per node a walker is ~100x slower than a workgroup round, a skip
  // pays from a few hundred bytes (measured: 32 levels -> 1.1 M of 2 M nodes went to the walkers, K3 34 -> 228 ms)
  a.lb_quiet = 0xFFFFFFFFu;
  if (const char *e = getenv("BCE_HIP_LOCAL_QUIET")) a.lb_quiet = (uint32_t)strtoul(e, nullptr, 10);
  if (a.lb_budget < 1u) a.lb_budget = 1u;
  if (a.lb_budget > 65536u) a.lb_budget = 65536u;
  a.budget = c->dbg_dfs_budget ? c->dbg_dfs_budget : K3_DFS_PASS;
  a.lane_budget = c->dbg_dfs_budget ? c->dbg_dfs_budget : K3_DFS_LANE_PASS;
  if (const char *e = getenv("BCE_HIP_DFS_LANE_PASS")) a.lane_budget = (uint32_t)strtoul(e, nullptr, 10);
  if (const char *e = getenv("BCE_HIP_DFS_PASS")) a.budget = (uint32_t)strtoul(e, nullptr, 10);
  a.dbg = getenv("BCE_HIP_DFS_DEBUG") ? 1u : 0u;
  a.no_spine = getenv("BCE_HIP_NO_SPINE") ? 1u : 0u;
  static_assert(sizeof(DfsCtl) <= 512, "the carve reserves 512 bytes");
  BCE_HIP_TRY(c, hipMemsetAsync(a.dctl, 0, sizeof(DfsCtl), c->stream));
  DfsCtl h;
  uint32_t count = live, passes = 0, jobs_done = 0;
  double t_pass = 0.0;
  if (a.dbg) { BCE_HIP_TRY(c, hipStreamSynchronize(c->stream)); t_pass = now_s(); }
  const DNode *in = nullptr;
  bool walkers_next = true;
  if (local) {
    // ---- workgroup-local rounds: the planes' lists first, then what the workgroups handed back (nodes alive after
    //      lb_budget rounds, remainders of workgroups that ran low, overflow of the LDS lists), re-chunked, until few
    //      enough are left for the walkers or a pass no longer thins them out (regular chains: the walkers' closed forms) ----
    uint32_t lcount = live, lpass = 0;
    const DNode *lin = nullptr;
    for (;;) {
      a.in = lin; a.in_count = lcount; a.out = queue[0]; a.spill = spillq[lpass & 1];
      hipLaunchKernelGGL(k3_local_kernel, dim3((lcount + LB_IN - 1) / LB_IN), dim3(LB_T), 0, c->stream, a);
      BCE_TRY(read_back(c, &h, a.dctl, sizeof h));
      BCE_HIP_TRY(c, hipGetLastError());
      c->stats.k3_launches += 1.0;
      if (a.dbg) {
        const double now = now_s();
        fprintf(stderr, "local pass %u: %u nodes in %u workgroups, %.3f ms, %u chains for the walkers so far, %u nodes handed back, %u symbol slots, %llu nodes done\n", lpass,
                lcount, (lcount + LB_IN - 1) / LB_IN, (now - t_pass) * 1e3, h.queued, h.spilled, h.nsym, (unsigned long long)h.nodes);
        t_pass = now;
      }
      ++lpass;
      if (h.err) break;
      const bool last = h.spilled <= local_from || lpass >= 32u || (uint64_t)h.spilled * 4 > (uint64_t)lcount * 3;
      if (last) {
        if ((uint64_t)h.queued + h.spilled > qcap) { h.err = 4; break; }
        if (h.spilled)
          BCE_HIP_TRY(c, hipMemcpyAsync(queue[0] + h.queued, spillq[(lpass - 1) & 1], (size_t)h.spilled * sizeof(DNode), hipMemcpyDeviceToDevice, c->stream));
        h.queued += h.spilled;
        break;
      }
      lin = spillq[(lpass - 1) & 1];
      lcount = h.spilled;
      BCE_HIP_TRY(c, hipMemsetAsync(&a.dctl->spilled, 0, 4, c->stream));
    }
    if (!h.err) {
      in = queue[0]; count = h.queued; passes = 1;
      BCE_HIP_TRY(c, hipMemsetAsync(&a.dctl->queued, 0, 4, c->stream));
      walkers_next = count > 0;
    } else walkers_next = false;
  }
  while (walkers_next) {
    a.in = in; a.in_count = count; a.out = queue[passes & 1];
    const uint32_t walkers = count < wmax ? count : wmax;
    static const uint32_t uni_max = getenv("BCE_HIP_UNI_MAX") ? (uint32_t)strtoul(getenv("BCE_HIP_UNI_MAX"), nullptr, 10) : KD_UNI_MAX;
    if (passes > 0 && count <= uni_max && !getenv("BCE_HIP_DFS_NO_UNI"))   // few walkers left: one WAVE each
      hipLaunchKernelGGL(k3_dfs_kernel<true>, dim3(count), dim3((count <= KD_HELP_MAXW && !getenv("BCE_HIP_DFS_NO_HELP")) ? KD_T * KD_HELP : KD_T), 0, c->stream, a);
    else {
      // (fewer walkers per wave -- BCE_HIP_DFS_LPW -- was measured on 62 K walkers: 64 lanes 16.7 ms, 32: 18.3, 16: 20.9, 4: 35.8:
      //  a pass lasts as long as its longest walk, not as the comparisons a wave serves one after the other)
      uint32_t lpw = KD_T;
      if (const char *e = getenv("BCE_HIP_DFS_LPW")) { lpw = (uint32_t)strtoul(e, nullptr, 10); if (lpw < 1u || lpw > (uint32_t)KD_T) lpw = KD_T; }
      a.lpw = lpw;
      hipLaunchKernelGGL(k3_dfs_kernel<false>, dim3((walkers + lpw - 1) / lpw), dim3(KD_T), 0, c->stream, a);
    }
    // the long staircase stretches the walkers queued in this pass: their events, over the whole grid
    hipLaunchKernelGGL(kd_jobs_kernel, dim3(1024), dim3(256), 0, c->stream, a, jobs_done);
    BCE_TRY(read_back(c, &h, a.dctl, sizeof h));
    BCE_HIP_TRY(c, hipGetLastError());
    jobs_done = h.njobs < KD_JOBS ? h.njobs : KD_JOBS;
    if (a.dbg) {
      const uint32_t uni_max_dbg = getenv("BCE_HIP_UNI_MAX") ? (uint32_t)strtoul(getenv("BCE_HIP_UNI_MAX"), nullptr, 10) : KD_UNI_MAX;
      const double now = now_s();
      fprintf(stderr, "dfs pass %u: %u walkers%s, %.3f ms, %u queued, %u symbols so far\n", passes, count,
              (passes > 0 && count <= uni_max_dbg) ? " (waves)" : "", (now - t_pass) * 1e3, h.queued, h.nsym);
      fprintf(stderr, "   staircase jobs so far: %u\n", jobs_done);
      fprintf(stderr, "   slowest chain-skip comparison so far: %llu K cycles, x = %llu, kk = %llu\n", h.dbg_slow >> 32, (h.dbg_slow >> 20) & 0xFFF, h.dbg_slow & 0xFFFFF);
      fprintf(stderr, "   slowest wave of this pass: %.2f M cycles, of which chain-skip comparisons %.2f M, staircases %.2f M\n",
              (double)(h.dbg_maxwave >> 40) * 4096e-6, (double)((h.dbg_maxwave >> 20) & 0xFFFFF) * 4096e-6, (double)(h.dbg_maxwave & 0xFFFFF) * 4096e-6);
      BCE_HIP_TRY(c, hipMemsetAsync(&a.dctl->dbg_maxwave, 0, 8, c->stream));
      if (h.dbg_gen[0] || h.dbg_gen[2])
        fprintf(stderr, "   wave-walkers so far: %llu general nodes at %.0f cycles each, %llu pass-through-loop nodes at %.0f cycles each\n", (unsigned long long)h.dbg_gen[0],
                h.dbg_gen[0] ? (double)h.dbg_gen[1] / h.dbg_gen[0] : 0.0, (unsigned long long)h.dbg_gen[2], h.dbg_gen[2] ? (double)h.dbg_gen[3] / h.dbg_gen[2] : 0.0);
      if (h.dbg_spine[0]) fprintf(stderr, "   spine bursts so far: %u tried, %u done, %u byte levels; ended by: small node %u, no c-row in A %u / B %u / E %u, run over %u; %u scans, %u with x1 < 4; K cycles finding the levels %u, walking their planes %u; stretches ended by B %u, by E %u, by both %u, by nothing %u\n", h.dbg_spine[0], h.dbg_spine[1], h.dbg_spine[2],
                                  h.dbg_why[0], h.dbg_why[1], h.dbg_why[2], h.dbg_why[3], h.dbg_why[5], h.dbg_why[6], h.dbg_why[7], h.dbg_why[4], h.dbg_spine[3], h.dbg_cont[0], h.dbg_cont[1], h.dbg_cont[2], h.dbg_cont[3]);
      fprintf(stderr, "   wave cycles so far: walk %.0f M, chain-skip comparisons %.0f M, staircases %.0f M (%llu looks)\n", h.dbg_cyc[0] * 1e-6,
              h.dbg_cyc[1] * 1e-6, h.dbg_cyc[2] * 1e-6, (unsigned long long)h.dbg_cyc[3]);
      t_pass = now;
    }
    c->stats.k3_launches += 1.0;
    ++passes;
    if (h.err || h.queued == 0) break;
    in = queue[(passes - 1) & 1];
    count = h.queued;
    BCE_HIP_TRY(c, hipMemsetAsync(&a.dctl->queued, 0, 4, c->stream));
  }
  if (a.dbg) {
    fprintf(stderr, "dfs: live %u passes %u err %u nsym %u (cap %u) nodes %llu maxround %llu maxvisited %u skips %u skipbytes %llu stairs %u + %u of several regions (%u symbols)\n", live,
            passes, h.err, h.nsym, cap, (unsigned long long)h.nodes, (unsigned long long)h.maxround, h.dbg_maxvis, h.dbg_skips,
            (unsigned long long)h.dbg_skipbytes, h.dbg_stairs & 0xFFFFu, h.dbg_stairs >> 16, h.dbg_stairsyms);
    for (int i = 0; i < 28; ++i) if (h.dbg_hist[i]) fprintf(stderr, "  x in [2^%d, 2^%d): %u nodes\n", i, i + 1, h.dbg_hist[i]);
    fprintf(stderr, "  several-region staircases, K cycles: decomposition %u, periodicity %u, sides + events %u; decompositions by galloping %u, by scanning all rows %u\n", h.dbg_hist[28], h.dbg_hist[29], h.dbg_hist[30], h.dbg_hist[26], h.dbg_hist[27]);
  }
  if (h.err == 2 && cap64 < left + 64 && cap64 < (512u << 20)) { cap_scale *= 4; goto retry; }   // nothing was modified: once more with more room
  if (h.err) return BCE_HIP_OK;                     // fall back to the rounds; nothing was modified
  const uint32_t m = h.nsym;                         // reserved slots (real symbols + holes)
  uint32_t mv = 0;                                   // real symbols
  for (int p = 0; p < 8; ++p) mv += h.cntp[p];
  if (mv > c->sym_cap) BCE_TRY(k3_grow_symbols(c, (uint64_t)mv + 1024));
  if (m) {
    // stream order inside a coder = (round, s); planes are separated: LSD sort by s, round low, (plane, round high),
    // each over the bits that can be set only (s < n; rounds <= maxround; a hole is (plane 7, round high 0xFF))
    const uint32_t g = (m + 255) / 256 < 2048 ? (m + 255) / 256 : 2048;
    const bool wide_round = h.maxround >= 0xFFFFFFFFull;
    const uint32_t sbits = ceil_log2(n), rbits = wide_round ? 32u : ceil_log2((uint32_t)h.maxround + 1u), hfirst = wide_round ? 0u : 7u;
    int r = 0;
    hipLaunchKernelGGL(kd_iota_kernel, dim3(g), dim3(256), 0, c->stream, a.ts, m, sk[0], sv[0]);
    BCE_TRY(radix_sort_pairs(c, sk, sv, m, 0, sbits ? sbits : 1u, &r, 9));
    uint32_t *k2[2] = {sk[r ^ 1], sk[r]}, *v2[2] = {sv[r], sv[r ^ 1]};
    hipLaunchKernelGGL(kd_gather_kernel, dim3(g), dim3(256), 0, c->stream, a.trlo, v2[0], m, k2[0]);
    BCE_TRY(radix_sort_pairs(c, k2, v2, m, 0, rbits ? rbits : 1u, &r, 9));
    uint32_t *k3[2] = {k2[r ^ 1], k2[r]}, *v3[2] = {v2[r], v2[r ^ 1]};
    hipLaunchKernelGGL(kd_gather_kernel, dim3(g), dim3(256), 0, c->stream, a.trhi, v3[0], m, k3[0]);
    BCE_TRY(radix_sort_pairs(c, k3, v3, m, hfirst, 11u - hfirst, &r, 9));
    if (c->scan_mode)
      hipLaunchKernelGGL(kd_place_raw_kernel, dim3(g), dim3(256), 0, c->stream, a.traw, v3[r], mv, c->scanrec.as<uint32_t>());
    else
      hipLaunchKernelGGL(kd_place_kernel, dim3(g), dim3(256), 0, c->stream, a.tkey, a.tesc, v3[r], mv,   // the holes sorted last
                         c->skey[0].as<uint32_t>(), c->sesc.as<uint32_t>());
  }
  // control block: the enumeration is finished
  EnumCtl *d = c->ctl.as<EnumCtl>();
  const uint64_t symtot = mv, nodes = ctl.nodes_total + h.nodes;
  const uint32_t done_round = (uint32_t)(h.maxround + 1 > 0xFFFFFFFEull ? 0xFFFFFFFEull : h.maxround + 1);
  BCE_HIP_TRY(c, hipMemcpyAsync(&d->sym_total, &symtot, 8, hipMemcpyHostToDevice, c->stream));
  BCE_HIP_TRY(c, hipMemcpyAsync(&d->nodes_total, &nodes, 8, hipMemcpyHostToDevice, c->stream));
  BCE_HIP_TRY(c, hipMemcpyAsync(&d->done_round, &done_round, 4, hipMemcpyHostToDevice, c->stream));
  BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->stats.spine_levels = h.spine_levels;
  for (int p = 0; p < 8; ++p) c->run_log[p].clear();
  uint64_t start = 0;
  for (int p = 0; p < 8; ++p) {
    if (h.cntp[p]) c->run_log[p].push_back(RunEntry{start, h.cntp[p], c->round});
    start += h.cntp[p];
  }
  *done = true;
  return BCE_HIP_OK;
}

}  // namespace bce
