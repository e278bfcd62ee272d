// bce_core.h -- per-element arithmetic shared by the HIP kernels (device) and the CPU unit tests
// (host, tests/ only).  Nothing here loops over the input: the product runs these functions inside
// kernels only.
//
// Data layout decisions (MI355X-first, not the reference's):
//  * rank directory: 16-byte granules {cum, w0, w1, w2} = running popcount + 96 payload bits, so one
//    rank query is ONE global_load_dwordx4 (the reference's 32+32-bit words, bce.cpp:138-151, are a
//    CPU choice; planes never reach the archive).
//  * node = absolute (s, x0, x1) u32 triple (the reference stores gamma-coded deltas, bce.cpp:226-356).
//  * symbol record = two u32 words (key word, escape word) in two arrays (see pack_symbol).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BCE_HD __host__ __device__ __forceinline__
#else
#define BCE_HD inline
#endif

namespace bce {

constexpr int kMaxK = 31;              // AdaptiveCoder<31>::max, bce.cpp:1381
constexpr uint32_t kGranuleBits = 96;  // payload bits per rank granule

struct Granule { uint32_t cum, w0, w1, w2; };
struct Node { uint32_t s, x0, x1; };

BCE_HD uint32_t popc32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__popc(x);
#else
  return (uint32_t)__builtin_popcount(x);
#endif
}

// floor(s / 96) for any 32-bit s: s/3 = (s * 0xAAAAAAAB) >> 33, then / 32.
BCE_HD uint32_t div96(uint32_t s) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umulhi(s, 0xAAAAAAABu) >> 6;          // one v_mul_hi_u32 (the 64-bit product form compiles to v_mad_u64_u32)
#else
  return (uint32_t)(((uint64_t)s * 0xAAAAAAABull) >> 38);
#endif
}

// rank1 inside one granule for offset o in [0, 96): b = the bits below o & 31; the word o falls in is masked with b, the
// words before it count whole, the words after it not at all.  ~13 VALU instructions on the device (two compares, one
// v_bfm, four selects, three ands, three popcounts with accumulate) -- the rank is the inner loop of every K3 kernel.
// (Selecting the WORD by o instead of the masks makes the compiler index the granule through scratch memory.)
BCE_HD uint32_t granule_rank1(const Granule &g, uint32_t o) {
  const bool lt32 = o < 32u, lt64 = o < 64u;
  const uint32_t b = (1u << (o & 31u)) - 1u;
  const uint32_t m0 = lt32 ? b : 0xFFFFFFFFu;
  const uint32_t m1 = lt32 ? 0u : (lt64 ? b : 0xFFFFFFFFu);
  const uint32_t m2 = lt64 ? 0u : b;
  return g.cum + popc32(g.w0 & m0) + popc32(g.w1 & m1) + popc32(g.w2 & m2);
}

// Per-plane model geometry derived from one config row (AdaptiveCoder::init, bce.cpp:700-705).
struct PlaneCfg {
  uint8_t bits[32];      // context bits per k
  uint32_t off[32];      // byte offset of k's counter block inside the plane's stat array (off_ & 0xFFFFFF)
  uint32_t ctxoff[32];   // first slot id of k (slot = ctxoff[k] + ctx)
  uint32_t stat_bytes;   // total counter bytes (= `start`, bce.cpp:705)
  uint32_t nslots;
};

BCE_HD void plane_cfg_init(PlaneCfg &c, const uint8_t row[32]) {
  uint32_t start = 0, slots = 0;
  for (int k = 0; k < 32; ++k) { c.bits[k] = row[k]; c.off[k] = 0; c.ctxoff[k] = 0; }
  for (int k = 2; k <= kMaxK; ++k) {
    c.off[k] = start;
    c.ctxoff[k] = slots;
    start += (uint32_t)k << (row[k] * 2);
    slots += 1u << (row[k] * 2);
  }
  c.stat_bytes = start;
  c.nslots = slots;
}

// Result of one node of BCE::code (bce.cpp:1261-1351), encode mode.
struct StepOut {
  uint32_t has0, has1, hassym;
  Node c0, c1;                 // children, absolute positions in plane (p+1)%8
  uint32_t sym, k, ctx1, ctx2, ctxs;   // coder_.set(sym, k, c1, c2, cs) arguments (bce.cpp:1302)
};

// One node of BCE::code (bce.cpp:1261-1351) in two halves, so a kernel can issue the rank gathers of many
// nodes together: node_pre needs rank1(s) and rank1(s+x); node_post needs rank1(s+x0) only when kind == 3.
struct NodePre {
  uint32_t s1, n1x, n0x, mn, mx;
  uint32_t kind;   // 0: all-0 pass-through, 1: all-1 pass-through, 2: mixed with forced split, 3: mixed, coded
};

BCE_HD void node_pre(const Node &nd, uint32_t r_s, uint32_t r_e, NodePre &pr) {
  const uint32_t x = nd.x0 + nd.x1;
  pr.s1 = r_s;                                        // :1265
  pr.n1x = r_e - r_s;                                 // _1x, :1271
  pr.n0x = x - pr.n1x;                                // _0x, :1281
  pr.mn = pr.mx = 0;
  if (pr.n1x == 0) { pr.kind = 0; return; }           // :1274
  if (pr.n0x == 0) { pr.kind = 1; return; }           // :1282
  uint32_t mn = nd.x0 - pr.n1x, mx = pr.n1x - nd.x1;  // :1290-1294
  mn = ((int32_t)mn < 0) ? 0u : mn;
  mx = ((int32_t)mx < 0) ? 0u : mx;
  mx = nd.x0 - mx;
  pr.mn = mn; pr.mx = mx;
  pr.kind = (mx != mn) ? 3u : 2u;                     // :1299
}

// zeros_p = rank0_p(n): child1 lists are indexed from C[p+1] = zeros(plane p) (bce.cpp:1128,1259).
BCE_HD void node_post(const Node &nd, uint32_t zeros_p, const NodePre &pr, uint32_t r_m, StepOut &o) {
  const uint32_t s = nd.s, x0 = nd.x0, x1 = nd.x1;
  const uint32_t s1 = pr.s1, s0 = s - s1;             // :1272
  o.has0 = o.has1 = o.hassym = 0;
  if (pr.kind == 0) {                                 // :1274-1279
    o.has0 = 1; o.c0.s = s0; o.c0.x0 = x0; o.c0.x1 = x1;
    return;
  }
  if (pr.kind == 1) {                                 // :1282-1287
    o.has1 = 1; o.c1.s = zeros_p + s1; o.c1.x0 = x0; o.c1.x1 = x1;
    return;
  }
  uint32_t n0x0 = pr.mn;                              // :1297
  if (pr.kind == 3) {                                 // :1299-1302
    n0x0 = (s + x0 - r_m) - s0;                       // rank0(s + x0) - s0
    o.hassym = 1; o.sym = n0x0 - pr.mn; o.k = pr.mx - pr.mn + 1; o.ctx1 = pr.n0x; o.ctx2 = x1; o.ctxs = x0 + x1;
  }
  const uint32_t n0x1 = pr.n0x - n0x0;                // :1337
  if (n0x0 && n0x1) { o.has0 = 1; o.c0.s = s0; o.c0.x0 = n0x0; o.c0.x1 = n0x1; }
  const uint32_t n1x1 = x1 - n0x1;                    // :1343-1344
  const uint32_t n1x0 = pr.n1x - n1x1;
  if (n1x0 && n1x1) { o.has1 = 1; o.c1.s = zeros_p + s1; o.c1.x0 = n1x0; o.c1.x1 = n1x1; }
}

// Branch-free form of the same node body for the kernels.  The two pass-through cases of the reference
// (bce.cpp:1274-1287) are special cases of the general formulas: with _1x == 0, min = max = x0 forces
// _0x0 = x0 and the "1" child vanishes; with _0x == 0, min = max = 0 forces _0x0 = 0 and the "0" child
// vanishes.  So: need_mid = (max != min) says whether rank1(s + x0) is needed at all, and node_flat
// produces children and the coder arguments with one select.
struct NodeFlat {
  uint32_t s1, s0, n1x, n0x, mn, mx;
  uint32_t need_mid;
};
BCE_HD void node_flat_pre(const Node &nd, uint32_t r_s, uint32_t r_e, NodeFlat &f) {
  const uint32_t x = nd.x0 + nd.x1;
  f.s1 = r_s; f.s0 = nd.s - r_s;
  f.n1x = r_e - r_s; f.n0x = x - f.n1x;
  const int32_t a = (int32_t)(nd.x0 - f.n1x), b = (int32_t)(f.n1x - nd.x1);
  f.mn = a < 0 ? 0u : (uint32_t)a;
  f.mx = nd.x0 - (b < 0 ? 0u : (uint32_t)b);
  f.need_mid = f.mx != f.mn;
}
// r_m = rank1(s + x0) (ignored unless need_mid).  Children: has0/has1 + (s, x0, x1); symbol args when need_mid.
BCE_HD void node_flat_post(const Node &nd, uint32_t zeros_p, const NodeFlat &f, uint32_t r_m, uint32_t &has0,
                           Node &c0, uint32_t &has1, Node &c1, uint32_t &sym, uint32_t &k) {
  const uint32_t n0x0 = f.need_mid ? (nd.s + nd.x0 - r_m) - f.s0 : f.mn;   // rank0(s + x0) - s0   (:1301)
  const uint32_t n0x1 = f.n0x - n0x0;                                       // :1337
  const uint32_t n1x1 = nd.x1 - n0x1, n1x0 = f.n1x - n1x1;                  // :1343-1344
  has0 = (n0x0 != 0u) & (n0x1 != 0u);
  has1 = (n1x0 != 0u) & (n1x1 != 0u);
  c0.s = f.s0; c0.x0 = n0x0; c0.x1 = n0x1;
  c1.s = zeros_p + f.s1; c1.x0 = n1x0; c1.x1 = n1x1;
  sym = n0x0 - f.mn; k = f.mx - f.mn + 1u;                                  // :1302
}

// rank1(pos) is supplied by the caller.  The sequential form used by the CPU unit tests.
template <class Rank1>
BCE_HD void node_step(const Node &nd, uint32_t zeros_p, Rank1 rank1, StepOut &o) {
  NodePre pr;
  node_pre(nd, rank1(nd.s), rank1(nd.s + nd.x0 + nd.x1), pr);
  node_post(nd, zeros_p, pr, pr.kind == 3 ? rank1(nd.s + nd.x0) : 0u, o);
}

// floor(a / b) for quotients known to be < 2^24 (here < 32): one float reciprocal and an exact integer
// correction instead of the generic ~30-instruction u32 division.  Exact: the float estimate is within +-1.
// Host and device run the SAME arithmetic (rn conversions, a correctly rounded reciprocal, one rn multiply),
// so tests/test_core_cpu.py checks on the CPU exactly what the kernels compute.
BCE_HD uint32_t small_quotient(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  const float r = __frcp_rn((float)b);
#else
  const float r = 1.0f / (float)b;
#endif
  uint32_t q = (uint32_t)((float)a * r);
  // q may be off by one either way (and a*rcp can round up to 2^k): fix with exact arithmetic
  uint64_t prod = (uint64_t)q * b;
  if (prod > a) { --q; prod -= b; }
  if ((uint64_t)a - prod >= b) ++q;
  return q;
}

// AdaptiveCoder::get_context's slot number (bce.cpp:671-677): ctx = (((c1 << bits) / cs) << bits) | ((c2 << bits) / cs)
// in uint32 arithmetic, i.e. c1 << bits WRAPS for c1 >= 2^(32-bits) (inputs of 2^27 bytes and more).
// Both quotients stay < 2^bits <= 32 (c1, c2 < cs; after a wrap cs > c1 >= 2^(32-bits) bounds it too).
BCE_HD uint32_t context_index(uint32_t bits, uint32_t c1, uint32_t c2, uint32_t cs) {
  return (small_quotient((uint32_t)(c1 << bits), cs) << bits) | small_quotient((uint32_t)(c2 << bits), cs);
}

// Symbol record = two u32 words kept in two arrays (SoA):
//   key word: [4:0] sym  [9:5] k  [25:10] slot  [28:26] plane      (sorted by K4 on bits 10..28)
//   esc word: [26:0] escbits  [31:27] nesc                         (goes to the host range coder as is)
// The k > 31 escape (bce.cpp:507-510) is unrolled here: nesc uniform bits (LSB first in escbits),
// then the residual (sym, k <= 31).  slot = ctxoff[k] + ctx with ctx as in get_context
// (bce.cpp:671-677), including its uint32 wrap-around.
constexpr int kSymRunShift = 10;        // key >> 10 = plane<<16 | slot : one model slot = one sorted run
constexpr int kSymRunBits = 19;

BCE_HD void pack_symbol(const PlaneCfg &cfg, uint32_t plane, uint32_t sym, uint32_t k, uint32_t c1, uint32_t c2,
                        uint32_t cs, uint32_t &key_word, uint32_t &esc_word) {
  uint32_t nesc = 0, esc = 0;
  while (k > (uint32_t)kMaxK) {
    esc |= (sym & 1u) << nesc;
    ++nesc;
    k = (k + (~sym & 1u)) >> 1;
    sym >>= 1;
  }
  const uint32_t bits = cfg.bits[k];
  const uint32_t ctx = context_index(bits, c1, c2, cs);
  const uint32_t slot = cfg.ctxoff[k] + ctx;
  key_word = sym | (k << 5) | (slot << 10) | (plane << 26);
  esc_word = esc | (nesc << 27);
}
// `bce -s`: what ScanCoder<31>::set (bce.cpp:737-744) keeps of a symbol, in ONE word (the kernels used to hand the host the
// raw 20-byte tuple): the k > 31 escape loop with ScanCoder's OWN halving (k >> 1) + (~s & 1) (quirk Q2: not AdaptiveCoder's)
// is run here -- it is integer arithmetic -- and so are the two 8-bit quotients of the map key, uint32 wrap included (:743).
//   [4:0] sym  [9:5] k (2..31)  [17:10] (c1 << 8) / cs  [25:18] (c2 << 8) / cs  [30:26] escapes (each one log(2) of cost)
BCE_HD uint32_t scan_pack(uint32_t sym, uint32_t k, uint32_t c1, uint32_t c2, uint32_t cs) {
  uint32_t nesc = 0;
  while (k > (uint32_t)kMaxK) {
    ++nesc;
    const uint32_t s0 = sym;
    sym = s0 >> 1;
    k = (k >> 1) + ((~s0) & 1u);
  }
  const uint32_t q1 = (uint32_t)(c1 << 8) / cs, q2 = (uint32_t)(c2 << 8) / cs;      // both < 256: c < cs, a wrap only shrinks the numerator
  return sym | (k << 5) | (q1 << 10) | (q2 << 18) | (nesc << 26);
}
BCE_HD uint32_t key_sym(uint32_t k) { return k & 31u; }
BCE_HD uint32_t key_k(uint32_t k) { return (k >> 5) & 31u; }
BCE_HD uint32_t key_slot(uint32_t k) { return (k >> 10) & 0xFFFFu; }
BCE_HD uint32_t key_plane(uint32_t k) { return (k >> 26) & 7u; }
BCE_HD uint32_t esc_bits(uint32_t e) { return e & 0x7FFFFFFu; }
BCE_HD uint32_t esc_n(uint32_t e) { return e >> 27; }

// Record handed to the host range coder, one u64 per symbol:
//   [12:0] cum   [20:13] freq - 1   [33:21] total   [61:34] escape bits with a sentinel: bits | 1 << nesc
// (cum, total < 8192: k <= 31 counters <= 254 each, + k; freq <= 255; nesc <= 27.)
BCE_HD uint64_t pack_model_out(uint32_t cum, uint32_t freq, uint32_t total, uint32_t esc_word) {
  const uint32_t sent = esc_bits(esc_word) | (1u << esc_n(esc_word));
  return (uint64_t)cum | ((uint64_t)(freq - 1u) << 13) | ((uint64_t)total << 21) | ((uint64_t)sent << 34);
}
BCE_HD uint32_t out_cum(uint64_t o) { return (uint32_t)(o & 0x1FFFu); }
BCE_HD uint32_t out_freq(uint64_t o) { return (uint32_t)((o >> 13) & 0xFFu) + 1u; }
BCE_HD uint32_t out_total(uint64_t o) { return (uint32_t)((o >> 21) & 0x1FFFu); }
BCE_HD uint32_t out_esc_sentinel(uint64_t o) { return (uint32_t)(o >> 34); }     // 1 = no escape bits

// One adaptive-model step on a slot's k byte counters (bce.cpp:512-518,529,531-533), the sequential
// definition the K4 kernel is tested against.
BCE_HD uint64_t model_step(uint8_t *ctr, uint32_t k, uint32_t s, uint32_t esc_word) {
  uint32_t l = 0;
  for (uint32_t i = 0; i < s; ++i) l += ctr[i];
  const uint32_t cum = l + s;
  for (uint32_t i = s; i < k; ++i) l += ctr[i];
  const uint32_t total = l + k;
  const uint32_t freq = (uint32_t)ctr[s] + 1u;
  if (++ctr[s] == 0xFF)
    for (uint32_t i = 0; i < k; ++i) ctr[i] >>= 1;
  return pack_model_out(cum, freq, total, esc_word);
}

}  // namespace bce
