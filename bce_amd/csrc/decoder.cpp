// decoder.cpp -- host C++ decoder for `.bce` archives (`bce -d`), SURVEY section 8f "next #1".
// Mirrors BCE::decode (bce.cpp:1169-1233), BCE::code mode 0 (bce.cpp:1236-1374), AdaptiveCoder::get
// (bce.cpp:555-608, shift_in :663-669), VCoder::getv (bce.cpp:372-377) and unbwt::bytewise
// (bce.cpp:1043-1102).  Not part of the accelerated -c path: plain host code, no HIP.
//
// Own design where the reference's is a CPU bit trick:
//  * Rank::set / finalize (bce.cpp:153-194) pack "rank1(x) = v" facts lazily into the 32+32-bit words.  Here
//    each plane keeps a dense array of rank1 at every boundary learnt so far (UNKNOWN elsewhere): set() is a
//    store, get() a load, and the planes are materialised at the end by filling the gaps between known
//    boundaries (a gap is all-0 or all-1, else the archive is inconsistent).  This also decodes the periodic
//    inputs on which the reference decoder returns zeros.
//  * inverse_bw_transform (libdivsufsort, bce.cpp:1091) is restated from its contract: BWT column without the
//    sentinel row, primary index idx (here always 1: the encoder rotated the input to its minimal rotation).
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <new>
#include <stdio.h>
#include <vector>

#include "../../include/bce_hip.h"
#include "bce_core.h"
#include "host_coder.h"

#include "decoder_core.h"

namespace {

using bce::Decoder;
using bce::kUnknown;

struct Triple { uint32_t s, x0, x1; };

}  // namespace

// Decode an archive produced by `bce -c`.  out == NULL: only report the decoded size in *out_len.
static int decompress_body(const uint8_t *archive, size_t len, uint8_t *out, size_t cap, size_t *out_len);
extern "C" int bce_hip_decompress(const uint8_t *archive, size_t len, uint8_t *out, size_t cap, size_t *out_len) {
  try { return decompress_body(archive, len, out, cap, out_len); }          // no C++ exception may cross the C ABI
  catch (const std::bad_alloc &) { return BCE_HIP_E_NOMEM; }
  catch (...) { return BCE_HIP_E_INTERNAL; }
}
static int decompress_body(const uint8_t *archive, size_t len, uint8_t *out, size_t cap, size_t *out_len) {
  if (!archive || !out_len) return BCE_HIP_E_ARG;
  bce::ArchiveHead hd;
  if (bce::parse_archive(archive, len, hd, /*header_only=*/true) != 0) return BCE_HIP_E_ARG;
  *out_len = hd.n;
  if (!out) return BCE_HIP_OK;
  if (cap < hd.n) return BCE_HIP_E_OVERFLOW;
  if (bce::parse_archive(archive, len, hd, false) != 0) return BCE_HIP_E_ARG;
  const uint32_t n = hd.n, offset = hd.offset;
  std::vector<Decoder> &dec = hd.dec;
  const uint32_t *C = hd.C;
  // boundary ranks: r1[p][x] = rank1_p(x) once known
  std::vector<std::vector<uint32_t>> r1(8);
  for (int p = 0; p < 8; ++p) { r1[p].assign((size_t)n + 1, kUnknown); r1[p][0] = 0; }
  for (int i = 0; i < 8; ++i) r1[(i + 7) & 7][n] = n - C[i];
  const bool timing = getenv("BCE_DEC_TIMING") != nullptr;
  auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tp0 = tnow();
  std::vector<Triple> cur[8][2], nxt[8][2];
  for (int i = 0; i < 8; ++i)
    if (C[i] && n - C[i]) cur[i][0].push_back(Triple{0u, C[i], n - C[i]});   // :1214-1216
  bool bad = false;
  for (bool again = true; again && !bad;) {                       // BCE::code, mode 0  (:1246-1371)
    for (int i = 0; i < 8 && !bad; ++i) {
      std::vector<uint32_t> &R = r1[i];
      const uint32_t zi = C[(i + 1) & 7];                         // zeros of plane i: child1 lists start there
      for (int j = 0; j < 2 && !bad; ++j)
        for (const Triple &nd : cur[i][j]) {
          const uint32_t s = nd.s, x0 = nd.x0, x1 = nd.x1, x = x0 + x1;
          if ((uint64_t)s + x > n || R[s] == kUnknown || R[s + x] == kUnknown) { bad = true; break; }
          const uint32_t s1 = R[s], n1x = R[s + x] - s1, s0 = s - s1;
          if (n1x > x) { bad = true; break; }
          uint32_t n1x0;
          if (!n1x) {                                             // :1274-1279
            nxt[(i + 1) & 7][0].push_back(Triple{s0, x0, x1});
            n1x0 = 0;
          } else if (n1x == x) {                                  // :1282-1287
            nxt[(i + 1) & 7][1].push_back(Triple{zi + s1, x0, x1});
            n1x0 = x0;
          } else {
            const uint32_t n0x = x - n1x;
            uint32_t mn = x0 - n1x, mx = n1x - x1;                // :1290-1294
            mn = ((int32_t)mn < 0) ? 0u : mn;
            mx = ((int32_t)mx < 0) ? 0u : mx;
            mx = x0 - mx;
            uint32_t n0x0 = mn;
            if (mx != mn) n0x0 = mn + dec[i].get_adaptive(mx - mn + 1, n0x, x1, x);   // :1304
            if (n0x0 > mx) { bad = true; break; }
            const uint32_t n0x1 = n0x - n0x0;
            if (n0x0 && n0x1) nxt[(i + 1) & 7][0].push_back(Triple{s0, n0x0, n0x1});
            const uint32_t n1x1 = x1 - n0x1;
            n1x0 = n1x - n1x1;
            if (n1x0 && n1x1) nxt[(i + 1) & 7][1].push_back(Triple{zi + s1, n1x0, n1x1});
          }
          R[s + x0] = s1 + n1x0;                                  // ranks[i].set(s + _x0, s1 + _1x0)  :1277,1285,1350
        }
    }
    again = false;
    for (int i = 0; i < 8; ++i)
      for (int j = 0; j < 2; ++j) { cur[i][j].swap(nxt[i][j]); nxt[i][j].clear(); if (!cur[i][j].empty()) again = true; }
  }
  if (bad) return BCE_HIP_E_INTERNAL;
  if (timing) { fprintf(stderr, "decode: code loop %.2f s\n", tnow() - tp0); tp0 = tnow(); }
  // materialise the planes: between two known boundaries all bits are equal
  std::vector<std::vector<uint8_t>> bit(8);
  for (int p = 0; p < 8; ++p) {
    bit[p].assign(n, 0);
    const std::vector<uint32_t> &R = r1[p];
    uint32_t a = 0;
    for (uint32_t b = 1; b <= n; ++b) {
      if (R[b] == kUnknown) continue;
      const uint32_t ones = R[b] - R[a], wdt = b - a;
      if (ones == wdt) memset(bit[p].data() + a, 1, wdt);
      else if (ones != 0) return BCE_HIP_E_INTERNAL;             // a mixed gap that was never split
      a = b;
    }
    std::vector<uint32_t>().swap(r1[p]);
  }
  // unbwt::bytewise (:1066-1085): walk the 8 planes with the cursor heap D (one chunk, a = 0)
  std::vector<uint8_t> bwt(n);
  {
    uint32_t D[256];
    memset(D, 0, sizeof D);
    std::vector<uint32_t> z(n + 1);                               // rank0 prefix of the current plane
    for (int j = 0; j < 7; ++j) {
      z[0] = 0;
      for (uint32_t q = 0; q < n; ++q) z[q + 1] = z[q] + (bit[j][q] ? 0u : 1u);
      const uint32_t zeros = z[n];
      for (int v = 0; v < (1 << j); ++v) {                        // :1071-1077
        const uint32_t e = D[(1 << j) | v];
        D[(2 << j) | v] = z[e];                                   // ranks[j].get<0>(e)
        D[(3 << j) | v] = zeros + (e - z[e]);                     // C[j] + ranks[j].get<1>(e)
      }
    }
    for (uint32_t i = 0; i < n; ++i) {                            // :1079-1084
      uint32_t chr = 0;
      for (int j = 0; j < 8; ++j) chr |= (uint32_t)bit[j][D[(1u << j) | chr]++] << j;
      bwt[i] = (uint8_t)chr;
    }
  }
  for (int p = 0; p < 8; ++p) std::vector<uint8_t>().swap(bit[p]);
  if (timing) { fprintf(stderr, "decode: planes + unbwt %.2f s\n", tnow() - tp0); tp0 = tnow(); }
  // inverse_bw_transform(out, out, nullptr, n, 1) + rotate (:1091-1093).  The encoder's BWT is the BWT of all
  // cyclic rotations with row 0 = the minimal rotation R (that is what idx = 1 says in libdivsufsort's
  // sentinel convention), so R is recovered by a cyclic LF walk from row 0; for periodic inputs the walk runs
  // round one cycle of the LF permutation several times, which yields the period repeated -- also correct.
  {
    uint32_t cnt[256];
    memset(cnt, 0, sizeof cnt);
    std::vector<uint32_t> occ(n);
    for (uint32_t r = 0; r < n; ++r) occ[r] = cnt[bwt[r]]++;
    uint32_t Cc[256], acc = 0;
    for (int c = 0; c < 256; ++c) { Cc[c] = acc; acc += cnt[c]; }
    std::vector<uint8_t> text(n);
    uint32_t row = 0;
    for (uint32_t i = n; i-- > 0;) {
      const uint8_t c = bwt[row];
      text[i] = c;
      row = Cc[c] + occ[row];
    }
    // std::rotate(out.begin(), out.end() - offset, out.end())  (:1093)
    const uint32_t off = offset % n;
    memcpy(out, text.data() + (n - off), off);
    memcpy(out + off, text.data(), n - off);
  }
  if (timing) fprintf(stderr, "decode: inverse BWT %.2f s\n", tnow() - tp0);
  return BCE_HIP_OK;
}
