// decoder_replay.cpp -- the host range decoder + adaptive model (decoder_core.h) on the REAL query stream of an archive:
// decodes the archive once with the plain host loop of decoder.cpp (recording every plane's queries in stream order), then
// replays each plane's recorded queries through Decoder::answer_batch -- what the GPU-assisted decoder's eight host threads
// run -- and prints ns per symbol per plane.  CPU only.
//   g++ -O2 -std=c++17 -I bce_amd/csrc tools/decoder_replay.cpp bce_amd/csrc/host_coder.cpp -o /tmp/decreplay -lpthread
//   /tmp/decreplay archive.bce [repeats]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>
#include "../bce_amd/csrc/decoder_core.h"
using namespace bce;
struct Triple { uint32_t s, x0, x1; };
int main(int argc, char **argv) {
  if (argc < 2) { printf("usage: decreplay archive.bce [repeats]\n"); return 1; }
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  std::ifstream f(argv[1], std::ios::binary | std::ios::ate);
  std::vector<uint8_t> a((size_t)f.tellg());
  f.seekg(0); f.read((char *)a.data(), (std::streamsize)a.size());
  ArchiveHead hd;
  if (parse_archive(a.data(), a.size(), hd, false) != 0) { printf("bad archive\n"); return 1; }
  const uint32_t n = hd.n;
  const uint32_t *C = hd.C;
  std::vector<Decoder> fresh = hd.dec;                     // the decoders right behind their preambles: the replay starts here
  std::vector<std::vector<uint32_t>> r1(8);
  for (int p = 0; p < 8; ++p) { r1[p].assign((size_t)n + 1, kUnknown); r1[p][0] = 0; }
  for (int i = 0; i < 8; ++i) r1[(i + 7) & 7][n] = n - C[i];
  std::vector<uint32_t> Q[8], ANS[8];
  std::vector<Decoder::Esc> E[8];
  std::vector<Triple> cur[8][2], nxt[8][2];
  for (int i = 0; i < 8; ++i) if (C[i] && n - C[i]) cur[i][0].push_back(Triple{0u, C[i], n - C[i]});
  for (bool again = true; again;) {
    for (int i = 0; i < 8; ++i) {
      std::vector<uint32_t> &R = r1[i];
      const uint32_t zi = C[(i + 1) & 7];
      for (int j = 0; j < 2; ++j)
        for (const Triple &nd : cur[i][j]) {
          const uint32_t s = nd.s, x0 = nd.x0, x1 = nd.x1, x = x0 + x1;
          const uint32_t s1 = R[s], n1x = R[s + x] - s1, s0 = s - s1;
          uint32_t n1x0;
          if (!n1x) { nxt[(i + 1) & 7][0].push_back(Triple{s0, x0, x1}); n1x0 = 0; }
          else if (n1x == x) { nxt[(i + 1) & 7][1].push_back(Triple{zi + s1, x0, x1}); n1x0 = x0; }
          else {
            const uint32_t n0x = x - n1x;
            uint32_t mn = x0 - n1x, mx = n1x - x1;
            mn = ((int32_t)mn < 0) ? 0u : mn; mx = ((int32_t)mx < 0) ? 0u : mx; mx = x0 - mx;
            uint32_t n0x0 = mn;
            if (mx != mn) {
              const uint32_t k = mx - mn + 1;
              const uint32_t ans = hd.dec[i].get_adaptive(k, n0x, x1, x);
              if (k > 31u) { Q[i].push_back(Decoder::kEscapeQuery); E[i].push_back(Decoder::Esc{k, n0x, x1, x}); }
              else { const uint32_t b = hd.dec[i].cfg.bits[k]; Q[i].push_back(k | ((((uint32_t)(n0x << b) / x) << b | ((uint32_t)(x1 << b) / x)) << 5)); }
              ANS[i].push_back(ans);
              n0x0 = mn + ans;
            }
            const uint32_t n0x1 = n0x - n0x0;
            if (n0x0 && n0x1) nxt[(i + 1) & 7][0].push_back(Triple{s0, n0x0, n0x1});
            const uint32_t n1x1 = x1 - n0x1; n1x0 = n1x - n1x1;
            if (n1x0 && n1x1) nxt[(i + 1) & 7][1].push_back(Triple{zi + s1, n1x0, n1x1});
          }
          R[s + x0] = s1 + n1x0;
        }
    }
    again = false;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 2; ++j) { cur[i][j].swap(nxt[i][j]); nxt[i][j].clear(); if (!cur[i][j].empty()) again = true; }
  }
  // cost of the rare classes: REPLACE_WIDE=1 / REPLACE_ESC=1 / REPLACE_MID=1 turn the k > 8 / escape / k 3..8 queries into binary ones
  // (the answers differ then; only the time is of interest)
  if (getenv("REPLACE_WIDE") || getenv("REPLACE_ESC") || getenv("REPLACE_MID"))
    for (int p = 0; p < 8; ++p) {
      std::vector<uint32_t> q2; std::vector<Decoder::Esc> e2;
      size_t ei = 0;
      for (uint32_t q : Q[p]) {
        if (q & Decoder::kEscapeQuery) { if (getenv("REPLACE_ESC")) q2.push_back(2u | (7u << 5)); else { q2.push_back(q); e2.push_back(E[p][ei]); } ++ei; continue; }
        const uint32_t k = q & 31u;
        if (k > 8 && getenv("REPLACE_WIDE")) q2.push_back(2u | ((q >> 5) & 255u) << 5);
        else if (k > 2 && k <= 8 && getenv("REPLACE_MID")) q2.push_back(2u | ((q >> 5) & 255u) << 5);
        else q2.push_back(q);
      }
      Q[p].swap(q2); E[p].swap(e2);
    }
  if (getenv("QSTAT")) {
    // the stream's shape: k histogram, how often a query meets the slot of the one before it, class changes
    for (int p = 0; p < 8; ++p) {
      const size_t cnt = Q[p].size();
      if (!cnt) continue;
      size_t h[7] = {0}, same = 0, change = 0; int prevc = -1; uint32_t prevq = ~0u;
      for (uint32_t q : Q[p]) {
        const uint32_t k = q & 31u;
        const int c = (q & Decoder::kEscapeQuery) ? 5 : k == 2 ? 0 : k == 3 ? 1 : k == 4 ? 2 : k <= 8 ? 3 : 4;
        ++h[c];
        if (q == prevq && c != 5) ++same;
        if (prevc >= 0 && (c == 0) != (prevc == 0)) ++change;
        prevc = c; prevq = q;
      }
      printf("plane %d: k=2 %.1f %%  k=3 %.1f %%  k=4 %.1f %%  k 5..8 %.1f %%  k 9..31 %.1f %%  escapes %.1f %%;  same slot as the query before: %.1f %%;  binary <-> other changes: %.1f %%\n", p,
             100.0 * h[0] / cnt, 100.0 * h[1] / cnt, 100.0 * h[2] / cnt, 100.0 * h[3] / cnt, 100.0 * h[4] / cnt, 100.0 * h[5] / cnt, 100.0 * same / cnt, 100.0 * change / cnt);
    }
  }
  uint64_t total = 0; double tsum = 0;
  for (int p = 0; p < 8; ++p) {
    const uint32_t cnt = (uint32_t)Q[p].size();
    if (!cnt) continue;
    std::vector<uint32_t> r(cnt);
    double best = 1e9;
    for (int it = 0; it < reps; ++it) {
      Decoder d = fresh[p];
      auto t0 = std::chrono::steady_clock::now();
      d.answer_batch(Q[p].data(), E[p].data(), r.data(), cnt);
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (dt < best) best = dt;
    }
    const bool ok = memcmp(r.data(), ANS[p].data(), (size_t)cnt * 4) == 0;
    size_t k2 = 0, k8 = 0, esc = 0;
    for (uint32_t q : Q[p]) { if (q & Decoder::kEscapeQuery) ++esc; else if ((q & 31u) == 2) ++k2; else if ((q & 31u) <= 8) ++k8; }
    printf("plane %d: %9u symbols  %.2f ns/symbol  (k=2 %.0f %%, k 3..8 %.0f %%, k 9..31 %.0f %%, escapes %.1f %%)  answers %s\n", p, cnt, best / cnt * 1e9,
           100.0 * k2 / cnt, 100.0 * k8 / cnt, 100.0 * (cnt - k2 - k8 - esc) / cnt, 100.0 * esc / cnt, ok ? "identical" : "DIFFERENT");
    total += cnt; tsum += best;
  }
  printf("all planes: %llu symbols, %.2f ns/symbol\n", (unsigned long long)total, tsum / total * 1e9);
  return 0;
}
