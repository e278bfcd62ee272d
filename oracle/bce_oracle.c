/*
 * bce_oracle.c -- CPU restatement of the reference `bce -c` path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the MI355X encoder.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may build, load or call it; the product path (bce_amd/) never does.
 *
 * It restates, in plain C, the algorithm of /root/reference/bce.cpp for the -c (compress) path.
 * Every function cites the reference lines it follows.  The one piece that is NOT in the reference
 * tree is libdivsufsort's divbwt() (bce.cpp:36,901; un-vendored, un-pinned, README.md:29-31): it is
 * restated here from its published contract (suffix sort of T[0..n) with an implicit smallest
 * sentinel, output U[0]=T[n-1], then T[SA[i]-1] skipping the suffix 0 whose 1-based position is the
 * returned primary index) on top of a textbook SA-IS suffix sorter (Nong/Zhang/Chan 2009).  The BWT
 * of a string is unique, so any correct suffix sorter gives byte-identical archives.
 *
 * Parity pin: the archive sha256 values recorded from the reference itself in SURVEY.md section 8c
 * (tests/test_oracle_golden.py).  The reference cannot be rebuilt in this image (divsufsort.h is
 * absent and stand-in headers are not allowed), so there is no oracle/_ref.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <stdio.h>

/* ------------------------------------------------------------------------------------------------
 * growable arrays
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint16_t *p; size_t n, cap; } vec16;
typedef struct { uint32_t *p; size_t n, cap; } vec32;

static void v16_push(vec16 *v, uint16_t x) {
  if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 1024; v->p = (uint16_t *)realloc(v->p, v->cap * 2); }
  v->p[v->n++] = x;
}
static void v32_push(vec32 *v, uint32_t x) {
  if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 1024; v->p = (uint32_t *)realloc(v->p, v->cap * 4); }
  v->p[v->n++] = x;
}
static void v32_push3(vec32 *v, uint32_t a, uint32_t b, uint32_t c) { v32_push(v, a); v32_push(v, b); v32_push(v, c); }

/* ------------------------------------------------------------------------------------------------
 * SA-IS suffix sorter (stands in for the suffix sort inside libdivsufsort's divbwt, bce.cpp:901).
 * s[0..n) over alphabet [0,K), s[n-1] == 0 is the unique smallest sentinel.
 * ---------------------------------------------------------------------------------------------- */
#define T_GET(i) ((t[(i) >> 3] >> ((i) & 7)) & 1)
#define T_SET(i, b) (t[(i) >> 3] = (uint8_t)((b) ? (t[(i) >> 3] | (1u << ((i) & 7))) : (t[(i) >> 3] & ~(1u << ((i) & 7)))))
#define IS_LMS(i) ((i) > 0 && T_GET(i) && !T_GET((i) - 1))

static void sa_buckets(const int32_t *s, int32_t *bkt, int32_t n, int32_t K, int end) {
  int32_t i, sum = 0;
  for (i = 0; i < K; ++i) bkt[i] = 0;
  for (i = 0; i < n; ++i) bkt[s[i]]++;
  for (i = 0; i < K; ++i) { sum += bkt[i]; bkt[i] = end ? sum : sum - bkt[i]; }
}
static void sa_induce_l(const uint8_t *t, int32_t *SA, const int32_t *s, int32_t *bkt, int32_t n, int32_t K) {
  int32_t i, j;
  sa_buckets(s, bkt, n, K, 0);
  for (i = 0; i < n; ++i) { j = SA[i] - 1; if (j >= 0 && !T_GET(j)) SA[bkt[s[j]]++] = j; }
}
static void sa_induce_s(const uint8_t *t, int32_t *SA, const int32_t *s, int32_t *bkt, int32_t n, int32_t K) {
  int32_t i, j;
  sa_buckets(s, bkt, n, K, 1);
  for (i = n - 1; i >= 0; --i) { j = SA[i] - 1; if (j >= 0 && T_GET(j)) SA[--bkt[s[j]]] = j; }
}
static void sais(const int32_t *s, int32_t *SA, int32_t n, int32_t K) {
  int32_t i, j;
  uint8_t *t = (uint8_t *)calloc((size_t)n / 8 + 1, 1);
  int32_t *bkt = (int32_t *)malloc(sizeof(int32_t) * (size_t)K);
  if (n == 1) { SA[0] = 0; free(t); free(bkt); return; }
  T_SET(n - 2, 0); T_SET(n - 1, 1);
  for (i = n - 3; i >= 0; --i) T_SET(i, (s[i] < s[i + 1] || (s[i] == s[i + 1] && T_GET(i + 1))) ? 1 : 0);
  /* stage 1: sort LMS substrings */
  sa_buckets(s, bkt, n, K, 1);
  for (i = 0; i < n; ++i) SA[i] = -1;
  for (i = 1; i < n; ++i) if (IS_LMS(i)) SA[--bkt[s[i]]] = i;
  sa_induce_l(t, SA, s, bkt, n, K);
  sa_induce_s(t, SA, s, bkt, n, K);
  int32_t n1 = 0;
  for (i = 0; i < n; ++i) if (IS_LMS(SA[i])) SA[n1++] = SA[i];
  for (i = n1; i < n; ++i) SA[i] = -1;
  int32_t name = 0, prev = -1;
  for (i = 0; i < n1; ++i) {
    int32_t pos = SA[i]; int diff = 0;
    for (int32_t d = 0; d < n; ++d) {
      if (prev == -1 || s[pos + d] != s[prev + d] || T_GET(pos + d) != T_GET(prev + d)) { diff = 1; break; }
      else if (d > 0 && (IS_LMS(pos + d) || IS_LMS(prev + d))) break;
    }
    if (diff) { name++; prev = pos; }
    SA[n1 + pos / 2] = name - 1;
  }
  for (i = n - 1, j = n - 1; i >= n1; --i) if (SA[i] >= 0) SA[j--] = SA[i];
  /* stage 2: solve the reduced problem */
  int32_t *SA1 = SA, *s1 = SA + n - n1;
  if (name < n1) sais(s1, SA1, n1, name);
  else for (i = 0; i < n1; ++i) SA1[s1[i]] = i;
  /* stage 3: induce the result */
  sa_buckets(s, bkt, n, K, 1);
  for (i = 1, j = 0; i < n; ++i) if (IS_LMS(i)) s1[j++] = i;
  for (i = 0; i < n1; ++i) SA1[i] = s1[SA1[i]];
  for (i = n1; i < n; ++i) SA[i] = -1;
  for (i = n1 - 1; i >= 0; --i) { j = SA[i]; SA[i] = -1; SA[--bkt[s[j]]] = j; }
  sa_induce_l(t, SA, s, bkt, n, K);
  sa_induce_s(t, SA, s, bkt, n, K);
  free(t); free(bkt);
}

/* libdivsufsort contract used at bce.cpp:901: divbwt(T, U, NULL, m) -> primary index.
 * n<=1 special cases as in libdivsufsort (returns n; U[0]=T[0] when n==1). */
static int32_t oracle_divbwt(const uint8_t *T, uint8_t *U, int32_t m) {
  if (m < 0) return -1;
  if (m <= 1) { if (m == 1) U[0] = T[0]; return m; }
  int32_t *s = (int32_t *)malloc(sizeof(int32_t) * ((size_t)m + 1));
  int32_t *SA = (int32_t *)malloc(sizeof(int32_t) * ((size_t)m + 1));
  for (int32_t i = 0; i < m; ++i) s[i] = (int32_t)T[i] + 1;
  s[m] = 0;
  sais(s, SA, m + 1, 257);
  /* SA[0] == m (the sentinel suffix).  Output convention: U[0] = T[m-1]; then T[SA[i]-1] for the m
   * real suffixes in order, skipping suffix 0, whose 1-based position is the primary index. */
  uint8_t *B = (uint8_t *)malloc((size_t)m);
  int32_t k = 0, pidx = 0;
  B[k++] = T[m - 1];
  for (int32_t i = 1; i <= m; ++i) {
    if (SA[i] == 0) pidx = i; else B[k++] = T[SA[i] - 1];
  }
  memcpy(U, B, (size_t)m);
  free(B); free(s); free(SA);
  return pidx;
}

/* test hooks for the libdivsufsort seam (tests/test_gpu_seam.py): the restated divbwt, and its inverse by the textbook
   LF walk (bce.cpp:1091 calls inverse_bw_transform(T, T, NULL, n, 1)) */
int32_t bce_oracle_divbwt(const uint8_t *T, uint8_t *U, int32_t n) { return oracle_divbwt(T, U, n); }
int bce_oracle_inverse_bwt(const uint8_t *T, uint8_t *U, int32_t n, int32_t idx) {
  if (n <= 0 || idx <= 0 || idx > n) return -1;
  /* rows of the BWT of T$: 0..n, row idx holds the sentinel; LF(row) = C[sym] + occurrences before the row */
  uint32_t C[258] = {0};
  uint32_t *lf = (uint32_t *)malloc(((size_t)n + 1) * sizeof(uint32_t));
  uint8_t *tmp = (uint8_t *)malloc((size_t)n);
  if (!lf || !tmp) { free(lf); free(tmp); return -2; }
  for (int32_t r = 0; r <= n; ++r) { uint32_t sy = r == idx ? 0u : (uint32_t)T[r < idx ? r : r - 1] + 1u; C[sy + 1]++; }
  for (int k = 1; k < 258; ++k) C[k] += C[k - 1];
  for (int32_t r = 0; r <= n; ++r) { uint32_t sy = r == idx ? 0u : (uint32_t)T[r < idx ? r : r - 1] + 1u; lf[r] = C[sy]++; }
  uint32_t row = 0;
  for (int32_t i = n - 1; i >= 0; --i) { tmp[i] = T[row < (uint32_t)idx ? row : row - 1]; row = lf[row]; }
  memcpy(U, tmp, (size_t)n);
  free(lf); free(tmp);
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * File::rotate (bce.cpp:858-894): index of the first minimal cyclic rotation; buffer rotated so
 * that the minimal rotation's first byte ends up LAST.  Returns offset_ (= i).
 * ---------------------------------------------------------------------------------------------- */
static void rotate_left(uint8_t *p, size_t n, size_t mid) { /* std::rotate(p, p+mid, p+n) */
  if (n == 0 || mid == 0 || mid == n) return;
  uint8_t *tmp = (uint8_t *)malloc(n);
  memcpy(tmp, p + mid, n - mid);
  memcpy(tmp + (n - mid), p, mid);
  memcpy(p, tmp, n);
  free(tmp);
}
uint32_t bce_oracle_rotate(uint8_t *map, uint32_t size) {
#define MOD(v) ({ uint64_t _v = (v); while (_v >= size) _v -= size; _v; })
  uint32_t i = 0, j = 1, k;
  while (j < size) {
    for (k = 0; map[MOD((uint64_t)i + k)] == map[MOD((uint64_t)j + k)] && k < size - 1; k++);
    if (map[MOD((uint64_t)i + k)] <= map[MOD((uint64_t)j + k)]) {
      j += k + 1;
    } else {
      i += k + 1;
      if (i < j) i = j++;
      else j = i + 1;
    }
  }
#undef MOD
  rotate_left(map, size, (size_t)i + 1);
  return i;
}

/* File::bwt (bce.cpp:896-910): divbwt on the first size-1 bytes, then the last byte is rotated
 * into slot pidx: std::rotate(begin + i, end - 1, end). */
void bce_oracle_bwt(uint8_t *map, uint32_t size) {
  uint32_t i = (uint32_t)oracle_divbwt(map, map, (int32_t)size - 1);
  uint8_t last = map[size - 1];
  memmove(map + i + 1, map + i, (size_t)(size - 1 - i));
  map[i] = last;
}

/* ------------------------------------------------------------------------------------------------
 * Rank (bce.cpp:130-151,196-219): u64 words, high 32 = payload bits, low 32 = running popcount.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint64_t *w; size_t nw; } Rank;
static void rank_init(Rank *r, uint32_t n) { r->nw = (size_t)n / 32 + 1; r->w = (uint64_t *)calloc(r->nw, 8); } /* :134-136 */
static void rank_set_bit(Rank *r, uint32_t off, uint64_t bit) { r->w[off / 32] |= bit << (off % 32); }             /* :200-202 */
static void rank_build(Rank *r) {                                                                                   /* :138-145 */
  uint32_t rank = 0;
  for (size_t i = 0; i < r->nw; ++i) { uint64_t b = r->w[i]; r->w[i] = (b << 32) | rank; rank += (uint32_t)__builtin_popcountll(b); }
}
static inline uint32_t rank_get1(const Rank *r, uint32_t index) {                                                  /* :147-151 */
  uint64_t rank = r->w[index / 32] & (~0ull >> (32 - index % 32));
  return (uint32_t)(rank + (uint64_t)__builtin_popcountll(rank >> 32));
}
static inline uint32_t rank_get0(const Rank *r, uint32_t index) { return index - rank_get1(r, index); }            /* :216-219 */

/* RankFile constructor body (bce.cpp:944-970): heap-indexed histogram, per-level exclusive scan,
 * 8n bit scatters, then Rank::build per plane. */
static void build_planes(const uint8_t *map, uint32_t size, Rank ranks[8]) {
  uint32_t C[256];
  for (int i = 0; i < 8; ++i) rank_init(&ranks[i], size);
  memset(C, 0, sizeof C);
  for (size_t i = 0; i < size; ++i) C[map[i] | 0x80]++;
  for (uint32_t i = 0x80; i < 0x100; ++i)
    for (uint32_t j = 1; j < 8; ++j)
      C[(((i << j) & 0xFF) | 0x80) >> j] += C[i];
  for (int i = 0; i < 8; ++i) {
    uint32_t sum = 0;
    for (int j = 1 << i; j < 1 << (i + 1); ++j) { uint32_t tmp = C[j]; C[j] = sum; sum += tmp; }
  }
  for (size_t i = 0; i < size; ++i) {
    uint32_t chr = map[i];
    for (int j = 0; j < 8; ++j) {
      uint32_t c = (chr & ((1u << j) - 1)) | (1u << j);
      rank_set_bit(&ranks[j], C[c]++, (chr >> j) & 1);
    }
  }
  for (int i = 0; i < 8; ++i) rank_build(&ranks[i]);
}

/* ------------------------------------------------------------------------------------------------
 * AdaptiveCoder<31> encode side (bce.cpp:484-553,610-615,655-661,671-710) + default init_ tables
 * (bce.cpp:713-724) + VCoder::setv (bce.cpp:364-370).
 * ---------------------------------------------------------------------------------------------- */
#define CMAX 31
static const uint8_t k_default_init[9][CMAX + 1] = {
  {0,0,5,5,5,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,0},
  {0,0,5,5,5,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,0},
  {0,0,5,5,5,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,3,3,3,3,0},
  {0,0,5,5,5,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,3,3,3,3,3,3,3,3,3,0},
  {0,0,5,5,4,4,4,4,4,4,4,4,3,3,3,3,3,3,3,3,3,3,3,3,3,3,3,3,3,3,3,0},
  {0,0,5,5,4,4,4,4,4,4,4,4,3,3,3,3,3,3,3,3,3,3,3,3,3,3,3,3,3,3,3,0},
  {0,0,5,4,4,4,4,4,4,3,3,3,3,3,3,3,3,3,3,3,3,3,3,3,3,3,3,3,3,3,3,0},
  {0,0,4,4,4,4,3,3,3,3,3,3,3,3,3,3,3,3,3,3,3,3,3,3,3,2,2,2,2,2,2,0},
  {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}
};
static uint8_t g_init[9][CMAX + 1];

/* optional trace sink: one entry per coder operation, in call order per coder */
typedef struct {
  int on;
  vec32 nodes;   /* per node: round, plane, abs_s, x0, x1 */
  vec32 syms;    /* per adaptive call (before escape expansion): plane, s, k, c1, c2, cs */
  vec32 ops;     /* per range-coder op: coder(-1 hdr => 8), cum, freq, total */
  uint32_t rounds;
} Trace;
static Trace g_tr;

typedef struct {
  uint64_t l_, h_;
  vec16 data_;
  uint32_t off_[CMAX + 1];
  uint8_t *stat_;
  uint32_t stat_size;
  int id;       /* 0..7 plane coder, 8 header coder (trace only) */
} __attribute__((aligned(128))) Coder;   /* own cache lines per coder: the OpenMP baseline should not lose to false sharing */

static void coder_shift_out(Coder *c) {                                   /* :655-661 */
  while (!((c->h_ ^ c->l_) >> 48)) {
    v16_push(&c->data_, (uint16_t)(c->h_ >> 48));
    c->l_ = (c->l_ << 16) + 0x0000;
    c->h_ = (c->h_ << 16) + 0xFFFF;
  }
}
/* the range-coder half shared by set(s,k) :538-553 and set(s,k,c1,c2,cs) :520-529,535 */
static void coder_encode(Coder *c, uint32_t cum, uint32_t freq, uint32_t total) {
  if (g_tr.on) { v32_push(&g_tr.ops, (uint32_t)c->id); v32_push3(&g_tr.ops, cum, freq, total); }
  if (c->h_ - c->l_ < total) {
    for (int i = 0; i < 4; ++i) v16_push(&c->data_, (uint16_t)(c->l_ >> (48 - 16 * i)));
    c->l_ = 0; c->h_ = ~0ull;
  }
  uint64_t step = (c->h_ - c->l_) / total;
  c->l_ += step * cum;
  c->h_ = c->l_ + step * freq - 1;
  coder_shift_out(c);
}
static void coder_set2(Coder *c, uint32_t s, uint32_t k) { coder_encode(c, s, 1, k); }   /* :538-553: l+=step*s; h=step+l-1 */

/* the slot number of get_context, :675: `auto ctx = (((c1 << bits) / cs) << bits) | ((c2 << bits) / cs);` with every operand
   uint32_t -- c1 << bits wraps for c1 >= 2^(32-bits), i.e. on inputs of 2^27 bytes and more */
static uint32_t context_index(uint32_t bits, uint32_t c1, uint32_t c2, uint32_t cs) {
  return (((uint32_t)(c1 << bits) / cs) << bits) | ((uint32_t)(c2 << bits) / cs);
}
/* test hook: the expression alone, for tests/test_core_cpu.py (wrap cases cannot be reached by small inputs) */
uint32_t bce_oracle_context_index(uint32_t bits, uint32_t c1, uint32_t c2, uint32_t cs) { return context_index(bits, c1, c2, cs); }

static uint8_t *coder_context(Coder *c, uint32_t k, uint32_t c1, uint32_t c2, uint32_t cs) { /* :671-677 (uint32 wrap kept) */
  uint32_t off = c->off_[k];
  uint32_t bits = off >> 24;
  uint32_t ctx = context_index(bits, c1, c2, cs);
  return c->stat_ + (off & 0x00FFFFFF) + ctx * k;
}
static void coder_set5(Coder *c, uint32_t s, uint32_t k, uint32_t c1, uint32_t c2, uint32_t cs) { /* :506-536 */
  while (k > CMAX) {                       /* tail recursion of :507-510 */
    coder_set2(c, s & 1, 2);
    k = (k + (~s & 1)) >> 1;
    s = s >> 1;
  }
  uint8_t *ctx = coder_context(c, k, c1, c2, cs);
  uint32_t l = 0;
  for (uint32_t i = 0; i < s; ++i) l += ctx[i];
  uint32_t n = l + s;
  for (uint32_t i = s; i < k; ++i) l += ctx[i];
  l += k;
  coder_encode(c, n, (uint32_t)ctx[s] + 1, l);   /* h_ = l_ + step*(ctx[s]+1) - 1, :529 */
  if (++ctx[s] == 0xFF)
    for (uint32_t i = 0; i < k; ++i) ctx[i] >>= 1;
}
static void coder_setv(Coder *c, uint32_t s) {                            /* :364-370 */
  while (s) { coder_set2(c, s & 1, 3); s >>= 1; }
  coder_set2(c, 2, 3);
}
static void coder_flush(Coder *c) {                                       /* :610-615 */
  coder_shift_out(c);
  uint32_t bits = (uint32_t)__builtin_clzll(c->l_ ^ c->h_) + 1;
  v16_push(&c->data_, (uint16_t)((c->h_ >> (64 - bits)) << (16 - bits)));
}
static void coder_init(Coder *c, int i) {                                 /* ctor :491-493 + init(1,i) :679-710 */
  memset(c, 0, sizeof *c);
  c->l_ = 0; c->h_ = ~0ull;
  if (0 > i || i > 7) i = 8;
  c->id = i;
  const uint8_t *bits = g_init[i];
  uint32_t last = 0;
  for (int b = 0; b < CMAX + 1; ++b) {
    uint32_t bit = bits[b];
    coder_set2(c, bit != last, 2);
    if (bit != last) coder_set2(c, bit, 6);
    last = bit;
  }
  uint32_t start = 0;
  for (int k = 2; k < CMAX + 1; ++k) {
    c->off_[k] = start | ((uint32_t)bits[k] << 24);
    start += (uint32_t)k << (bits[k] * 2);
  }
  c->stat_ = (uint8_t *)calloc(start ? start : 1, 1);
  c->stat_size = start;
}
static void coder_free(Coder *c) { free(c->data_.p); free(c->stat_); }

/* ------------------------------------------------------------------------------------------------
 * BCE::code, mode 1 (bce.cpp:1236-1374).  The four gamma-coded pArray queues per plane
 * (bce.cpp:226-356,1237) are plain arrays of (delta, x0, x1) here: pArray stores exactly the
 * values pushed, so the queue contents are the same.
 * ---------------------------------------------------------------------------------------------- */
static int g_threads = 1;   /* bce_oracle_set_threads: 1 = the reference built without OpenMP, 8 = with (CMakeLists.txt) */

static void bce_code(Coder coder_[8], const uint32_t C[8], Rank ranks[8], uint32_t n) {
  typedef struct { vec32 q[4]; } __attribute__((aligned(128))) PlaneQ;   /* the four queues of a plane (:1237) */
  PlaneQ QQ[8];
  memset(QQ, 0, sizeof QQ);
#define Q(i, j) QQ[i].q[j]
  for (int i = 0; i < 8; ++i)
    if (C[i] && n - C[i]) v32_push3(&Q(i, 0), 1, C[i], n - C[i]);               /* :1238-1240 */
  int again;
  uint32_t round = 0;
  do {
    again = 0;
    /* the reference's only parallel region: one OpenMP thread per plane, joined every round (:1250-1252) */
#pragma omp parallel for schedule(static, 1) num_threads(g_threads) if (g_threads > 1 && !g_tr.on)
    for (int i = 0; i < 8; ++i) {                                                /* :1252 */
      uint32_t offset[2] = {0, 0};
      for (int j = 0; j < 2; ++j) {                                              /* :1256 */
        const uint32_t *cur = Q(i, j).p, *end = Q(i, j).p + Q(i, j).n;
        uint32_t s = C[i] * (uint32_t)j;                                         /* :1259 */
        while (cur != end) {
          s += *cur++ - 1;                                                       /* :1264 */
          uint32_t s1 = rank_get1(&ranks[i], s);
          uint32_t _x0 = *cur++, _x1 = *cur++;
          uint32_t _x = _x0 + _x1;
          if (g_tr.on) { v32_push(&g_tr.nodes, round); v32_push(&g_tr.nodes, (uint32_t)i); v32_push3(&g_tr.nodes, s, _x0, _x1); }
          uint32_t _1x = rank_get1(&ranks[i], s + _x) - s1;                      /* :1271 */
          uint32_t s0 = s - s1;
          if (!_1x) {                                                            /* :1274-1279 */
            v32_push3(&Q(i, 2), s0 - offset[0] + 1, _x0, _x1);
            offset[0] = s0;
            continue;
          }
          uint32_t _0x = _x - _1x;
          if (!_0x) {                                                            /* :1282-1287 */
            v32_push3(&Q(i, 3), s1 - offset[1] + 1, _x0, _x1);
            offset[1] = s1;
            continue;
          }
          uint32_t mn = _x0 - _1x, mx = _1x - _x1;                               /* :1290-1294 */
          mn &= ~(uint32_t)((int32_t)mn >> 31);
          mx &= ~(uint32_t)((int32_t)mx >> 31);
          mx = _x0 - mx;
          uint32_t _0x0 = mn;
          if (mx != mn) {                                                        /* :1299-1302 */
            _0x0 = rank_get0(&ranks[i], s + _x0) - s0;
            if (g_tr.on) { v32_push(&g_tr.syms, (uint32_t)i); v32_push(&g_tr.syms, _0x0 - mn); v32_push(&g_tr.syms, mx - mn + 1); v32_push3(&g_tr.syms, _0x, _x1, _x); }
            coder_set5(&coder_[i], _0x0 - mn, mx - mn + 1, _0x, _x1, _x);
          }
          uint32_t _0x1 = _0x - _0x0;                                            /* :1337-1348 */
          if (_0x0 && _0x1) { v32_push3(&Q(i, 2), s0 - offset[0] + 1, _0x0, _0x1); offset[0] = s0; }
          uint32_t _1x1 = _x1 - _0x1;
          uint32_t _1x0 = _1x - _1x1;
          if (_1x0 && _1x1) { v32_push3(&Q(i, 3), s1 - offset[1] + 1, _1x0, _1x1); offset[1] = s1; }
        }
      }
    }
    for (int i = 0; i < 8; ++i) {                                                /* :1361-1370 */
      vec32 t;
      t = Q((i + 1) % 8, 0); Q((i + 1) % 8, 0) = Q(i, 2); Q(i, 2) = t;
      t = Q((i + 1) % 8, 1); Q((i + 1) % 8, 1) = Q(i, 3); Q(i, 3) = t;
      Q(i, 2).n = 0; Q(i, 3).n = 0;
      if (Q((i + 1) % 8, 0).n || Q((i + 1) % 8, 1).n) again = 1;
    }
    round++;
  } while (again);
  g_tr.rounds = round;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) free(Q(i, j).p);
#undef Q
}

/* BCE::encode (bce.cpp:1117-1167): 8 plane coders, header coder main(-1), concatenation. */
static void bce_encode(Rank ranks[8], uint32_t n, uint32_t file_offset, vec16 *out, uint32_t C_out[8]) {
  Coder coder_[8];
  uint32_t C[8];
  for (int i = 0; i < 8; ++i) coder_init(&coder_[i], i);                         /* :1124 */
  for (int i = 0; i < 8; ++i) {                                                  /* :1126-1130 */
    C[i] = rank_get0(&ranks[(i + 7) % 8], n);
    coder_set2(&coder_[i], C[i], n + 1);
    if (C_out) C_out[i] = C[i];
  }
  bce_code(coder_, C, ranks, n);                                                 /* :1132 */
  uint32_t size = 0;
  for (int i = 0; i < 8; ++i) { coder_flush(&coder_[i]); size += (uint32_t)coder_[i].data_.n; }   /* :1134-1138 */
  Coder mainc;                                                                   /* :1141-1149 */
  coder_init(&mainc, -1);
  coder_setv(&mainc, n);
  coder_set2(&mainc, file_offset, n + 1);
  coder_setv(&mainc, size);
  int s = (int)size;
  for (int i = 0; i < 7; ++i) {
    coder_set2(&mainc, (uint32_t)coder_[i].data_.n, (uint32_t)s + 1);
    s -= (int)coder_[i].data_.n;
  }
  coder_flush(&mainc);
  v16_push(out, (uint16_t)mainc.data_.n);                                        /* :1152-1157 */
  for (size_t k = 0; k < mainc.data_.n; ++k) v16_push(out, mainc.data_.p[k]);
  for (int i = 0; i < 8; ++i)
    for (size_t k = 0; k < coder_[i].data_.n; ++k) v16_push(out, coder_[i].data_.p[k]);
  coder_free(&mainc);
  for (int i = 0; i < 8; ++i) coder_free(&coder_[i]);
}

/* ------------------------------------------------------------------------------------------------
 * public entry points (ctypes-friendly)
 * ---------------------------------------------------------------------------------------------- */
static void load_cfg(const uint8_t *config) {        /* load_config :626-641 (file I/O is the caller's) */
  if (config) memcpy(g_init, config, sizeof g_init); else memcpy(g_init, k_default_init, sizeof g_init);
}

/* main() -c branch (bce.cpp:1403-1427) minus file I/O: in -> malloc'd archive bytes.
 * returns 0, or -1 for n == 0 (the reference crashes on empty input). */
/* The reference's -DM_TIME stage timers ("Rotate:" bce.cpp:864-866,886-891, "BWT:" :897-909, "Rank:" :941-943,974-979,
 * "Encode:" :1118-1120,1159-1164) for the last bce_oracle_compress: seconds of rotate, bwt, plane build, encode. */
static double g_stage_s[4];
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
void bce_oracle_stage_seconds(double out[4]) { memcpy(out, g_stage_s, sizeof g_stage_s); }

int bce_oracle_compress(const uint8_t *in, uint32_t n, const uint8_t *config, uint8_t **out, size_t *out_len) {
  if (n == 0) return -1;
  load_cfg(config);
  uint8_t *map = (uint8_t *)malloc(n);
  memcpy(map, in, n);
  double t0 = now_s();
  uint32_t off = bce_oracle_rotate(map, n);       /* RankFile ctor :935-936 */
  double t1 = now_s();
  bce_oracle_bwt(map, n);
  double t2 = now_s();
  Rank ranks[8];
  build_planes(map, n, ranks);
  double t3 = now_s();
  free(map);
  vec16 data = {0};
  bce_encode(ranks, n, off, &data, NULL);
  g_stage_s[0] = t1 - t0; g_stage_s[1] = t2 - t1; g_stage_s[2] = t3 - t2; g_stage_s[3] = now_s() - t3;
  for (int i = 0; i < 8; ++i) free(ranks[i].w);
  *out = (uint8_t *)data.p;                       /* native-endian u16 words, :1426 */
  *out_len = data.n * 2;
  return 0;
}
void bce_oracle_free(void *p) { free(p); }

/* Stage outputs for parity tests: BWT bytes + offset. */
int bce_oracle_bwt_stage(const uint8_t *in, uint32_t n, uint8_t *bwt_out, uint32_t *offset_out) {
  if (n == 0) return -1;
  memcpy(bwt_out, in, n);
  *offset_out = bce_oracle_rotate(bwt_out, n);
  bce_oracle_bwt(bwt_out, n);
  return 0;
}

/* Encode from a given BWT (skips rotate/bwt): used to test the GPU planes/enumeration/model
 * independently of the suffix sorter.  Also fills C[8] (= zeros of plane (i+7)%8). */
int bce_oracle_encode_from_bwt(const uint8_t *bwt, uint32_t n, uint32_t file_offset, const uint8_t *config,
                               uint8_t **out, size_t *out_len, uint32_t C_out[8]) {
  if (n == 0) return -1;
  load_cfg(config);
  Rank ranks[8];
  build_planes(bwt, n, ranks);
  vec16 data = {0};
  bce_encode(ranks, n, file_offset, &data, C_out);
  for (int i = 0; i < 8; ++i) free(ranks[i].w);
  *out = (uint8_t *)data.p;
  *out_len = data.n * 2;
  return 0;
}

/* Plane bits as the reference lays them out: plane j, position p -> bit.  out: 8 * n bytes of 0/1. */
int bce_oracle_plane_bits(const uint8_t *bwt, uint32_t n, uint8_t *out) {
  Rank ranks[8];
  build_planes(bwt, n, ranks);
  for (int j = 0; j < 8; ++j) {
    for (uint32_t p = 0; p < n; ++p) out[(size_t)j * n + p] = (uint8_t)((ranks[j].w[p / 32] >> (p % 32 + 32)) & 1);  /* Rank::bit :196-198 */
    free(ranks[j].w);
  }
  return 0;
}

/* Threads of the round loop for later calls (1..8); the archive does not depend on it. */
void bce_oracle_set_threads(int t) { g_threads = t < 1 ? 1 : (t > 8 ? 8 : t); }

/* Tracing: switch on, run a compress/encode call, read the arrays, switch off. */
void bce_oracle_trace_begin(void) {
  g_tr.on = 1; g_tr.nodes.n = g_tr.syms.n = g_tr.ops.n = 0; g_tr.rounds = 0;
}
void bce_oracle_trace_end(void) { g_tr.on = 0; }
void bce_oracle_trace_free(void) {
  free(g_tr.nodes.p); free(g_tr.syms.p); free(g_tr.ops.p);
  memset(&g_tr, 0, sizeof g_tr);
}
size_t bce_oracle_trace_nodes(const uint32_t **p) { *p = g_tr.nodes.p; return g_tr.nodes.n / 5; }
size_t bce_oracle_trace_syms(const uint32_t **p) { *p = g_tr.syms.p; return g_tr.syms.n / 6; }
size_t bce_oracle_trace_ops(const uint32_t **p) { *p = g_tr.ops.p; return g_tr.ops.n / 4; }
uint32_t bce_oracle_trace_rounds(void) { return g_tr.rounds; }

/* Synthetic generators of SURVEY.md section 8c (xorshift64*, synth-rand v1, synth-text v1). */
static uint64_t xs_next(uint64_t *st) {
  uint64_t x = *st;
  x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
  *st = x;
  return x * 0x2545F4914F6CDD1DULL;
}
void bce_oracle_synth_rand(uint64_t seed, uint8_t *out, size_t n) {
  uint64_t st = seed; size_t k = 0;
  while (k < n) { uint64_t v = xs_next(&st); for (int b = 0; b < 8 && k < n; ++b) out[k++] = (uint8_t)(v >> (8 * b)); }
}
void bce_oracle_synth_text(uint64_t seed, uint8_t *out, size_t n) {
  static const char letters[] = "etaoinshrdlcumwfgypbvkjxqz";
  enum { V = 4096 };
  uint64_t st = seed;
  uint8_t (*word)[10] = (uint8_t (*)[10])malloc(V * 10);
  uint8_t *wl = (uint8_t *)malloc(V);
  for (int w = 0; w < V; ++w) {
    int L = 2 + (int)(xs_next(&st) % 8);
    wl[w] = (uint8_t)L;
    for (int c = 0; c < L; ++c) {
      uint64_t a = xs_next(&st) % 26, b = xs_next(&st) % 26;
      word[w][c] = (uint8_t)letters[a < b ? a : b];
    }
  }
  size_t k = 0; uint64_t cnt = 0;
  while (k < n) {
    uint64_t r = xs_next(&st);
    uint64_t a = (r >> 32) % V, b = (r & 0xffffffffULL) % V;
    uint64_t w = (a * b) >> 12;
    for (int c = 0; c < wl[w] && k < n; ++c) out[k++] = word[w][c];
    cnt++;
    if (cnt % 13 == 0) { if (k < n) out[k++] = '.'; if (k < n) out[k++] = ' '; }
    else { if (k < n) out[k++] = ' '; }
  }
  free(word); free(wl);
}
