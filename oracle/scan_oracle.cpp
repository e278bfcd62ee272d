// scan_oracle.cpp -- CPU restatement of `bce -s` (ScanCoder<31>, bce.cpp:726-834).  TEST INFRASTRUCTURE ONLY
// (same rules as bce_oracle.c: used by tests/ and nothing else).
//
// C++ rather than C on purpose: the reference's result depends on the iteration order of
// std::unordered_map<uint32_t, std::vector<uint8_t>> (bce.cpp:756,765,825) and on the order of its double
// additions.  Using the same libstdc++ container with the same insertion sequence reproduces
// that order exactly.  The symbol stream comes from the C oracle's trace (bce_oracle.c), i.e. from the restated
// BCE::code loop; this file restates ScanCoder::set (:737-744), ScanCoder::flush (:751-800) and the way
// BCE::encode drives it (:1124-1149).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <array>
#include <unordered_map>
#include <vector>

extern "C" {
void bce_oracle_trace_begin(void);
void bce_oracle_trace_end(void);
void bce_oracle_trace_free(void);
size_t bce_oracle_trace_syms(const uint32_t **p);
int bce_oracle_encode_from_bwt(const uint8_t *bwt, uint32_t n, uint32_t file_offset, const uint8_t *config,
                               uint8_t **out, size_t *out_len, uint32_t C_out[8]);
int bce_oracle_bwt_stage(const uint8_t *in, uint32_t n, uint8_t *bwt_out, uint32_t *offset_out);
void bce_oracle_free(void *p);
}

namespace {
constexpr int MAX = 31;

struct ScanCoder {
  std::array<std::unordered_map<uint32_t, std::vector<uint8_t>>, MAX + 1> stat_;
  double z_ = 0;
  int i_;
  explicit ScanCoder(int i) : i_(i < 0 || i > 7 ? 8 : i) {}                    // :733

  void set(uint32_t s, uint32_t k, uint32_t c1, uint32_t c2, uint32_t cs) {   // :737-744
    while (k > (uint32_t)MAX) {
      z_ += log(2);
      const uint32_t s0 = s;
      s = s0 >> 1;
      k = (k >> 1) + ((~s0) & 1);
    }
    stat_[k][(((uint32_t)(c2 << 8) / cs) << 16) | ((uint32_t)(c1 << 8) / cs)].push_back((uint8_t)s);
  }

  void flush(uint8_t init[9][MAX + 1], double *result_bytes) {                // :751-800
    std::vector<uint16_t> s;
    for (uint32_t k = 2; k < (uint32_t)MAX; ++k) {
      double z_min = 0;
      for (auto &pair : stat_[k]) z_min += log(k) * pair.second.size();
      for (uint32_t j = 0; j <= 5; ++j) {
        s.clear();
        s.insert(s.begin(), k << (2 * j), 0);
        double z = 0;
        for (auto &pair : stat_[k]) {
          auto c = pair.first;
          uint16_t c1 = c >> 0;
          uint16_t c2 = c >> 16;
          c1 >>= 8 - j;
          c2 >>= 8 - j;
          c = (c1 << j) | c2;
          auto *ctx = &s[c * k];
          for (auto &sym : pair.second) {
            uint32_t l = k;
            for (uint32_t i = 0; i < k; ++i) l += ctx[i];
            z += log(static_cast<double>(l) / (1 + ctx[sym]));
            if (++ctx[sym] == 0xFF)
              for (uint32_t i = 0; i < k; ++i) ctx[i] >>= 1;
          }
        }
        if (z < z_min) { z_min = z; init[i_][k] = (uint8_t)j; }
      }
      z_ += z_min;
    }
    if (result_bytes) *result_bytes = z_ / log(256);
  }
};
}  // namespace

// main() -s branch (bce.cpp:1384-1402) minus file I/O: input -> 288-byte config; result_bytes[9] = the
// "Result size" lines.  ScanCoder::init_ is a zero-initialised static (:834), so untouched entries are 0.
extern "C" int bce_oracle_scan(const uint8_t *in, uint32_t n, uint8_t *config288, double *result_bytes9) {
  if (n == 0) return -1;
  std::vector<uint8_t> bwt(n);
  uint32_t off = 0;
  if (bce_oracle_bwt_stage(in, n, bwt.data(), &off) != 0) return -1;
  bce_oracle_trace_begin();
  uint8_t *arch = nullptr; size_t alen = 0; uint32_t C[8];
  // the enumeration does not depend on the coder policy: run the restated BCE::code once and replay its
  // coder_[i].set(...) calls (in call order per coder) into eight ScanCoders
  bce_oracle_encode_from_bwt(bwt.data(), n, off, nullptr, &arch, &alen, C);
  bce_oracle_free(arch);
  const uint32_t *syms = nullptr;
  const size_t ns = bce_oracle_trace_syms(&syms);
  std::vector<ScanCoder> coders;
  for (int i = 0; i < 8; ++i) coders.emplace_back(i);                           // :1124
  for (size_t t = 0; t < ns; ++t) {
    const uint32_t *r = syms + 6 * t;                                           // plane, s, k, c1, c2, cs
    coders[r[0]].set(r[1], r[2], r[3], r[4], r[5]);
  }
  bce_oracle_trace_end();
  bce_oracle_trace_free();
  uint8_t init[9][MAX + 1];
  memset(init, 0, sizeof init);
  for (int i = 0; i < 8; ++i) coders[i].flush(init, result_bytes9 ? result_bytes9 + i : nullptr);   // :1135-1138
  ScanCoder mainc(-1);                                                          // :1141-1149 (set(s,k) is a no-op)
  mainc.flush(init, result_bytes9 ? result_bytes9 + 8 : nullptr);
  memcpy(config288, init, 288);
  return 0;
}
