// common.h -- context object and helpers shared by the stage implementations of libbcehip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <new>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include <sys/mman.h>

#include "../../include/bce_hip.h"
#include "bce_core.h"
#include "host_coder.h"

namespace bce {

// grow-only device buffer
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Device-resident control block of the enumeration (K3).  Everything a round needs to know about
// the previous round lives here, so rounds can be queued back to back without host round trips.
struct EnumCtl {
  uint32_t cnt[2][8][2];     // [parity][plane][seg] node counts (seg0 = child0 list, seg1 = child1 list)
  uint64_t sym_total;        // symbol records in the buffer (since the last flush)
  uint64_t sym_cap;          // capacity of the symbol buffer
  uint64_t nodes_total;      // nodes visited so far
  uint32_t need_flush;       // a round was skipped because the symbol buffer could overflow
  uint32_t skip_round;       // first skipped round
  uint32_t done_round;       // first round whose successor is empty (0xFFFFFFFF while running)
  uint32_t overflow;         // node buffer overflow (fatal)
  uint64_t next_nodes;       // node total of the next round (after the last executed round): up to 8 (n/2) = 2^33
  uint64_t want_syms;        // symbol records of the skipped round (to grow the buffer when one round exceeds it)
  uint32_t ticket;           // k3_scan_kernel: arrival counter of its 8 plane blocks (last one runs the epilogue)
  uint32_t pad1;
  uint32_t ptot[8][3];       // per plane totals of the round: child0, child1, symbols
  uint64_t symbase[8];       // symbol-buffer base of each plane's records in this round
  uint32_t tail_rounds;      // rounds executed by the last k3_tail_kernel launch
  uint32_t small_bail;       // k3_small_kernel: the round (skip_round) does not fit its grid / tile table: run it wide
  uint32_t want_list;        // with `overflow`: the longest list the skipped round would write (the host makes that room: k3_grow_lists)
  uint32_t stalled;          // a single-launch round waited K3_SPIN_LIMIT polls for a predecessor tile's word: fatal, loud (see wait_word)
};

// Pinned host staging of one model flush + the batch descriptor handed to the coder threads.
struct FlushSlot {
  uint64_t *h_out = nullptr;     // one packed u64 per symbol (bce_core.h pack_model_out)
  size_t cap = 0;
  CoderBatch batch;
  hipEvent_t ev_start = nullptr, ev_copy = nullptr;   // K4 start (compute stream), outputs in h_out (copy stream)
  hipEvent_t ev_kend = nullptr;                       // K4's last kernel (compute stream): the kernels' own time, without the copy
  bool timed = false;            // the events of the last flush have not been added to stats.t_model yet
  std::shared_ptr<std::once_flag> once;   // the first coder thread to arrive waits for ev_copy, the others for it
  bool registered = false;       // h_out is a private mapping under hipHostRegister (k4_prepin), not hipHostMalloc's
  // staged ahead of its first use, on a thread of its own (k4_prepin): the buffer it prepares, adopted by the next flush
  std::thread pin_th;
  uint64_t *pin_p = nullptr;
  size_t pin_cap = 0;
  double pin_s = 0;
};

// ---- host memory registered with the runtime (hipHostRegister) ------------------------------------------------------------
// Rules, each closing one way in which a registered range can lose its pages under the GPU (round 4's
// "memory access fault ... write access to a read-only page" at an address inside the C library's heap):
//  1. A registered range is ALWAYS an anonymous private mapping of its own (huge_map) -- never memory carved out of the C
//     library's heap, whose pages go back to the kernel (brk trim), to other callers and to khugepaged as malloc sees fit.
//  2. MADV_DONTFORK: a fork() of the host program (Python's subprocess, system()) does not turn the pages copy-on-write under
//     the device's mapping (what the runtime's own pinned memory has).
//  3. Huge pages are taken at the FIRST TOUCH only: the caller touches every page (reg_map_settle), then the mapping is
//     MADV_NOHUGEPAGE -- the huge pages it got stay, but khugepaged never collapses 4 KB pages into a new huge page (= other
//     physical pages) while the range is registered.
//  4. hipHostUnregister does not wait for the device (hipHostFree does): before it, the owner waits for exactly the events /
//     streams whose work touches the range (the slot's copy event; the context's three streams) -- not hipDeviceSynchronize,
//     which would stall the other contexts of a pool.
// BCE_HIP_REG_MIN=bytes lowers the sizes from which buffers take this route (defaults: 8 MB for a flush slot, 256 MB for the
// decoder's boundary ranks) so that tests reach it with small inputs.
inline size_t reg_min_bytes(size_t dflt) {
  const char *e = getenv("BCE_HIP_REG_MIN");                          // (read per allocation: a test sets it for one context)
  return e ? (size_t)strtoull(e, nullptr, 10) : dflt;
}
// An anonymous mapping of its own, 2 MB-aligned, with huge pages asked for (3.2 GB are touched in a tenth of the time; the coder
// threads that read the staging buffers gain 1-2 %): `bytes` usable at the returned
// address; huge_unmap gives it back.  (The mapping is 2 MB longer than asked; the unaligned head and tail go back at once.)
inline void *huge_map(size_t bytes) {
  const size_t two_mb = (size_t)2 << 20, used = (bytes + 4095) & ~(size_t)4095;
  void *base = mmap(nullptr, used + two_mb, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (base == MAP_FAILED) return nullptr;
  char *q = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(base) + two_mb - 1) & ~(uintptr_t)(two_mb - 1));
  const size_t head = (size_t)(q - static_cast<char *>(base));          // (a multiple of the page size, below 2 MB)
  if (head) (void)munmap(base, head);                                    // what lies in front of the aligned address ...
  if (two_mb - head) (void)munmap(q + used, two_mb - head);              // ... and behind the end goes back at once
  (void)madvise(q, used, MADV_DONTFORK);                                 // rule 2
  static const bool no_thp = getenv("BCE_HIP_NO_THP") != nullptr;     // (diagnostic: what a host without free huge pages gives)
  if (!no_thp) (void)madvise(q, used, MADV_HUGEPAGE);
  return q;
}
// every page of the mapping has been touched: from here on its physical pages stay what they are (rule 3)
inline void reg_map_settle(void *q, size_t bytes) { (void)madvise(q, (bytes + 4095) & ~(size_t)4095, MADV_NOHUGEPAGE); }
inline void huge_unmap(void *q, size_t bytes) { if (q) (void)munmap(q, (bytes + 4095) & ~(size_t)4095); }
// bytes of the mapping k4_prepin makes for a slot of `cap` records
inline size_t slot_map_bytes(size_t cap) { return (cap * 8 + 16 + 4095) & ~(size_t)4095; }
// free a slot's host staging, whichever way it was pinned.  The only device work that touches a slot's staging is the
// device-to-host copy of its last flush, which ev_copy follows (rule 4; an event never recorded is complete).
inline void slot_free_host(FlushSlot &s, uint32_t *unmaps = nullptr) {
  if (s.h_out) {
    if (s.registered) {
      if (s.ev_copy) (void)hipEventSynchronize(s.ev_copy);
      (void)hipHostUnregister(s.h_out);
      huge_unmap(s.h_out, slot_map_bytes(s.cap));
      if (unmaps) ++*unmaps;
    } else (void)hipHostFree(s.h_out);
  }
  s.h_out = nullptr; s.cap = 0; s.registered = false;
}

struct RunEntry { uint64_t start; uint32_t count; uint32_t round; };  // symbols of (round, plane): [start, start+count)

}  // namespace bce

struct bce_hip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;             // device-to-host copies of the model outputs, overlapping the next rounds
  // K4 (model flushes) on a stream of its own, lowest priority: the next K3 rounds run beside it.  K3 writes its symbol
  // records into skey[0] / sesc, K4 reads the pair of the flush before (skey_alt / sesc_alt): the two are swapped at
  // every flush.  Off by default (BCE_HIP_OVERLAP=1 or debug knob 11 switch it on): see bce_hip_create.
  hipStream_t k4_stream = nullptr;
  hipEvent_t ev_k3_batch = nullptr;              // the rounds whose symbols a flush takes are done (main stream)
  hipEvent_t ev_k4_done[2] = {nullptr, nullptr}; // flush f's kernels have read their symbol buffers (k4 stream), f & 1
  bool overlap = false;
  bool gated = true;                             // bce_hip_set_gated: the enumerations of one device's contexts take turns (default on)
  bool gate_held = false;
  bool gate_lent = false;                        // given to a waiting context behind a model flush (api.hip: gate_lend / gate_regain)
  double gate_t0 = 0, gate_wait_s = 0, gate_held_s = 0;   // (BCE_HIP_GATE_TIMING: seconds this compression waited for / held the device gate)
  uint32_t flush_seq = 0;                        // flushes issued by this context (parity selects ev_k4_done)
  hipEvent_t copy_busy = nullptr;                // last copy out of `sout` (the next K4 must not overwrite it earlier)
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_k4 = nullptr;
  char err[256] = {0};

  uint32_t n = 0;
  int stage = 0;  // 0 empty, 1 loaded, 2 bwt, 3 planes, 4 encoded
  int phase = 0;  // what is running: 1 K1, 2 K2, 3 the enumeration / model (-c, -s), 4 a decode; 0 nothing.  ctx_trim() gives back the
                  // buffers the running phase does not use when the device runs out of memory (the context keeps every stage's
                  // buffers for its next input; at n ~ 2^31 they do not all fit together)
  uint32_t offset = 0;
  uint8_t config[9][32];
  bce::PlaneCfg cfg[8];
  uint32_t zeros[8] = {0};
  bool k1_unique = false;                        // K1 ended with all rotations distinct
  bool k1_valid = false;                         // sa[sa_res] / rank hold this input's suffix order (K1 ran; no injected BWT)
  int sa_res = 0;
  // debug knobs (bce_hip_debug_set): 0 = default
  uint32_t dbg_dfs_budget = 0, dbg_no_dfs = 0, dbg_no_tail = 0, dbg_no_skip = 0, dbg_no_small = 0, dbg_step_small = 0, dbg_no_fused = 0, dbg_no_local = 0, dbg_capp_div = 0, dbg_local_from = 0, dbg_local_budget = 0, dbg_tail_round = 0;
  uint64_t sym_cap_user = 0;
  bool sync_flush = false;                       // BCE_HIP_SYNC_FLUSH: flushes wait for their copy (profiling)
  bce_hip_progress_fn progress = nullptr;        // bce_hip_set_progress
  void *progress_user = nullptr;

  // device buffers (grow-only)
  bce::DevBuf text, bwt;                         // n bytes each
  bce::DevBuf sa[2], key[2], rank, k2, nrk, act[2];   // K1: 9 x 4n (act = active-set lists)
  bce::DevBuf khi[2], dl[4], kflag, actv[2];                   // K1: high key words of the first sort, then new rank / suffix per active element;
                                                      //     deferred (big-group) lists; per-element flags
  bce::DevBuf rs_hist, blk;                      // radix histograms [256][nb]; per-block scratch
  bce::DevBuf ptmp[2];                           // K2 byte ping-pong
  bce::DevBuf gran;                              // 8 planes x ngran x 16 B
  uint32_t ngran = 0;                            // granules per plane
  // node lists: one buffer per round parity, 8 planes x capL[parity] nodes each (a round reads one parity and writes the
  // other; the one about to be written holds nothing and can be replaced by a larger one without a copy: k3_grow_lists).
  // The decoder keeps both parities in nlist[0]: 2 x 8 planes x capP nodes.
  bce::DevBuf nlist[2];
  uint32_t capL[2] = {0, 0};
  uint32_t capP = 0;
  bce::DevBuf ctl, tilecnt, tileoff, runs;       // K3 control
  bce::DevBuf smwords;                           // k3_small_kernel: one published count word per tile
  size_t k3_groups = 0;
  uint32_t k3_count2_grid = 0;                   // blocks of k3_count2_kernel that are resident together
  bce::DevBuf k3tw, k3grp;                       // k3_count2_kernel: count word per tile; group words, tickets, offsets
  bce::DevBuf dfs;                               // depth-first tail: tagged symbols, sort scratch, walker stacks
  bce::DevBuf truns;                             // run table of the persistent tail kernel [K3_TAIL_MAXROUNDS][8]
  void *h_truns = nullptr;
  bce::DevBuf skey[2], sval[2], sout, sesc;       // K3->K4: symbol keys (skey[0]) + escape words (sesc); sort ping-pong; outputs
  bce::DevBuf skey_alt, sesc_alt, rs_hist_k4;     // the other pair of symbol buffers (see k4_stream); K4's radix histograms
  uint64_t sym_cap = 0;
  bool scan_mode = false;                        // `bce -s`: K3 emits scan_pack words (bce_core.h) into scanrec
  bce::DevBuf scanrec;
  bce::DevBuf stat, dcfg, k4w;                   // K4 counters, device copy of PlaneCfg[8], per-window work arrays
  uint32_t stat_off[8] = {0};

  // pinned host staging
  void *h_ctl = nullptr, *h_runs = nullptr;
  void *h_small = nullptr;                       // 4 KB of pinned host memory for read_back()
  void *h_big = nullptr;                         // pinned host memory for the decoder's host tail (the boundary ranks: 32 (n + 1) bytes), grow-only
  size_t h_big_cap = 0;
  uint64_t dec_cap_next = 0;                     // a decode that ran out of list room starts again with this many nodes per list
  bool dec_list_overflow = false;
  uint32_t dec_restarts = 0;
  uint32_t reg_maps = 0, reg_unmaps = 0;         // registered host mappings made / given back since the context was created
  bool h_big_registered = false;                 // h_big is big_host_alloc's private mapping under hipHostRegister
  void *dec_pin[3] = {nullptr, nullptr, nullptr};  // the decoder's pinned query / escape-record / answer buffers, kept from one decode to the next (grow-only)
  size_t dec_pin_cap[3] = {0, 0, 0};
  bce::FlushSlot slot[3];                        // flushes in flight: GPU fills one while the coders drain the others
  int slot_next = 0;

  // enumeration stepping state
  uint32_t round = 0;
  bool enum_active = false;
  std::vector<bce::RunEntry> run_log[8];         // runs since the last flush, per plane

  bce::HostCoder *coder = nullptr;
  bce_hip_stats stats;
  // where a cold context's time goes (BCE_CLI_TIMING=1 prints them): device allocations, pinned host allocations
  // k4_prepin's threads touch their pages at once and register them with the runtime when it is up (bce_hip_create_sized
  // starts them BEFORE the runtime's initialisation): 0 = wait, 1 = go, 2 = give up
  std::mutex stage_mu;
  std::condition_variable stage_cv;
  int stage_state = 1;
  double alloc_s = 0, pin_s = 0;
  uint64_t alloc_bytes = 0, pin_bytes = 0;
  uint32_t alloc_calls = 0, pin_calls = 0;
};

namespace bce {

inline double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// The decoder's large pinned host buffer (the boundary ranks of the host tail: 32 (n + 1) bytes, 3.2 GB at 10^8): from 256 MB on
// (reg_min_bytes), a private mapping (huge_map), touched by four threads, then registered with the runtime (hipHostRegister,
// mapped: the device address is the host's, or the buffer is given up).  Measured: 3.2 GB in 0.04 s against
// hipHostMalloc's 0.55-0.7 s, which also holds the kernels queued meanwhile up.  Only the copy engine and the host touch this
// buffer.  The small buffers KERNELS write while the host reads (queries, answers, the mailbox) are hipHostMalloc's.
// BCE_DEC_NO_HUGE=1: hipHostMalloc for all.  big_host_free waits for the owning context's streams first (rule 4 above).
void *big_host_alloc(size_t bytes, int device, bool *registered);
void big_host_free(bce_hip_ctx *c, void *p, size_t bytes, bool registered);

inline int set_err(bce_hip_ctx *c, hipError_t e, const char *what, int line) {
  snprintf(c->err, sizeof c->err, "%s:%d: %s", what, line, hipGetErrorString(e));
  return e == hipErrorOutOfMemory ? BCE_HIP_E_NOMEM : BCE_HIP_E_DEVICE;
}

#define BCE_HIP_TRY(ctx, expr)                                        \
  do {                                                                \
    hipError_t _e = (expr);                                           \
    if (_e != hipSuccess) return bce::set_err((ctx), _e, #expr, __LINE__); \
  } while (0)

#define BCE_TRY(expr)                 \
  do {                                \
    int _rc = (expr);                 \
    if (_rc != BCE_HIP_OK) return _rc; \
  } while (0)

size_t ctx_trim(bce_hip_ctx *c);                   // api.hip: bytes given back (0: nothing left to give)
// BCE_HIP_TEST_OOM=k (test hook): every k-th device allocation of the process is treated as "out of memory" at its first attempt,
// so that the give-back-and-retry path runs in every phase without a 288 GB input.
inline bool test_oom_now() {
  static const char *e = getenv("BCE_HIP_TEST_OOM");
  static const unsigned long k = e ? strtoul(e, nullptr, 10) : 0;
  static std::atomic<unsigned long> count{0};
  return k && (count.fetch_add(1) + 1) % k == 0;
}
inline int ensure(bce_hip_ctx *c, DevBuf &b, size_t bytes) {
  if (bytes <= b.cap) return BCE_HIP_OK;
  if (b.p) {
    // The old buffer may still be read or written by work this context has queued (a model flush and its copy run
    // behind the host): nothing of it may be in flight when the memory goes back to the allocator, because another
    // context's hipMalloc -- another host thread -- can be handed it at once.
    for (hipStream_t st : {c->stream, c->k4_stream, c->copy_stream}) if (st) BCE_HIP_TRY(c, hipStreamSynchronize(st));
    (void)hipFree(b.p);
    b.p = nullptr; b.cap = 0;
  }
  const double t0 = now_s();
  hipError_t e = hipMalloc(&b.p, bytes);
  const bool pretend = e == hipSuccess && test_oom_now();
  if (pretend) { (void)hipFree(b.p); e = hipErrorOutOfMemory; }
  if (e == hipErrorOutOfMemory) {
    // the buffers of the stages that are not running go back, then once more
    (void)hipGetLastError();
    b.p = nullptr;
    if (ctx_trim(c) || pretend) e = hipMalloc(&b.p, bytes);
  }
  if (e != hipSuccess) { b.p = nullptr; BCE_HIP_TRY(c, e); }
  c->alloc_s += now_s() - t0; c->alloc_bytes += bytes; c->alloc_calls++;
  if (now_s() - t0 > 0.02 && getenv("BCE_ALLOC_TRACE")) fprintf(stderr, "alloc: hipMalloc of %.1f MB took %.3f s\n", bytes / 1e6, now_s() - t0);
  b.cap = bytes;
  return BCE_HIP_OK;
}
// pinned host memory of the context (timed like the device allocations)
inline int pin_alloc(bce_hip_ctx *c, void **p, size_t bytes) {
  const double t0 = now_s();
  BCE_HIP_TRY(c, hipHostMalloc(p, bytes, hipHostMallocDefault));
  c->pin_s += now_s() - t0; c->pin_bytes += bytes; c->pin_calls++;
  return BCE_HIP_OK;
}
inline void release(DevBuf &b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr; b.cap = 0;
}

// A few bytes from the device, now: through pinned memory of the context's own (a copy into pageable memory -- a stack
// variable -- goes through the runtime's staging buffers, which all host threads and contexts share).
inline int read_back(bce_hip_ctx *c, void *dst, const void *dev_src, size_t bytes) {
  if (bytes > 4096) return BCE_HIP_E_ARG;
  if (!c->h_small) BCE_TRY(pin_alloc(c, &c->h_small, 4096));
  BCE_HIP_TRY(c, hipMemcpyAsync(c->h_small, dev_src, bytes, hipMemcpyDeviceToHost, c->stream));
  BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
  memcpy(dst, c->h_small, bytes);
  return BCE_HIP_OK;
}

// C ABI boundary: no C++ exception may leave an entry point (a host allocation that fails in a std::vector, a thread that
// cannot be started): the heavy entry points run their body through this.
template <class F>
inline int bce_guarded(bce_hip_ctx *c, F &&body) {
  try { return body(); }
  catch (const std::bad_alloc &) { if (c) snprintf(c->err, sizeof c->err, "host allocation failed"); return BCE_HIP_E_NOMEM; }
  catch (const std::exception &e) { if (c) snprintf(c->err, sizeof c->err, "%s", e.what()); return BCE_HIP_E_INTERNAL; }
  catch (...) { if (c) snprintf(c->err, sizeof c->err, "unknown exception"); return BCE_HIP_E_INTERNAL; }
}

// roctx ranges around the kernel families (the reference's M_TIME stage lines, bce.cpp:886-891,974-979,1159-1164, as
// profiler markers: rocprofv3 --marker-trace shows K1 / K2 / K3 batches / K4 flushes on the host timeline).  The marker
// library is looked up at run time -- librocprofiler-sdk-roctx.so, else libroctx64.so -- and absent = no-op.
struct RoctxRange {
  explicit RoctxRange(const char *name);
  ~RoctxRange();
  RoctxRange(const RoctxRange &) = delete;
  RoctxRange &operator=(const RoctxRange &) = delete;
 private:
  bool on_;
};

constexpr uint32_t K3_TAIL_CAP = 1024;        // nodes the tail kernel holds in LDS (all 8 planes together)
constexpr uint32_t K3_SMALL_MAXTILES = 2048;  // tile table of the one-launch round kernel
#ifndef K3_SMALL_NODES_VALUE
#define K3_SMALL_NODES_VALUE 2000000u   // (1/2 M: 19.8 ms of K3 on the natural corpus, 1 M: 19.7, 1.5 M: 19.4, 2 M: 19.3; its tile table holds 2048.
                                        //  With a table of 4096: 3 M 19.5 vs 19.7 in the same run, 4 M 20.5; text 13.0 / 13.1 vs 12.9)
#endif
constexpr uint32_t K3_SMALL_NODES = K3_SMALL_NODES_VALUE;   // rounds up to this many nodes use it
constexpr uint32_t K3_TAIL_ENTER = 512;       // the host switches to the tail kernel at or below this many nodes
constexpr uint32_t K3_TAIL_MAXROUNDS = 65536; // rounds per tail launch (bounded by its run table)

inline uint32_t ceil_log2(uint32_t v) { uint32_t b = 0; while ((1ull << b) < v) ++b; return b; }

// ---- stage implementations (one .hip file each) ----
int k1_bwt(bce_hip_ctx *c);                         // k1_bwt.hip
int k1_divbwt(bce_hip_ctx *c, const uint8_t *T_host, uint8_t *U_host, uint32_t n, uint32_t *pidx);   // the libdivsufsort seam
int kd_inverse_bw_transform(bce_hip_ctx *c, const uint8_t *T_host, uint8_t *U_host, uint32_t n, uint32_t idx);   // kd_decode.hip
int k2_build_planes(bce_hip_ctx *c);                // k2_planes.hip
int k2_get_plane_bits(bce_hip_ctx *c, int plane, uint8_t *out);
int k2_rank1(bce_hip_ctx *c, int plane, const uint32_t *idx, uint32_t count, uint32_t *out);
int k3_begin(bce_hip_ctx *c);                       // k3_enumerate.hip
int k3_rounds(bce_hip_ctx *c, uint32_t count, uint64_t nodes_hint);
int k3_rounds_small(bce_hip_ctx *c, uint32_t count, uint64_t cur_nodes, bool growing);   // one launch per round, narrow rounds
int k3_round_masked(bce_hip_ctx *c, uint32_t mask, bool repeat);   // one round, three launches, symbols of the planes in `mask` only
int k3_clear_need_flush(bce_hip_ctx *c);
int k3_clear_small_bail(bce_hip_ctx *c);   // queue `count` rounds from c->round (no sync)
int k3_dfs_tail(bce_hip_ctx *c, const EnumCtl &ctl, uint32_t enter, bool *done);   // k3_dfs.hip: finish the enumeration depth-first
int k3_tail(bce_hip_ctx *c, uint32_t max_rounds = K3_TAIL_MAXROUNDS);   // queue the persistent narrow-round kernel from c->round (no sync)
int k3_fetch_tail_runs(bce_hip_ctx *c, uint32_t rounds);
int k3_sync_ctl(bce_hip_ctx *c, EnumCtl *out);      // copy the control block back (syncs the stream)
int k3_fetch_runs(bce_hip_ctx *c, uint32_t first_round, uint32_t count);  // append run-table rows to run_log
int k3_get_nodes(bce_hip_ctx *c, int plane, uint32_t *out, uint32_t cap, uint32_t *count);
int k3_reset_symbols(bce_hip_ctx *c);               // after a flush: sym_total = 0, need_flush = 0
int k3_grow_symbols(bce_hip_ctx *c, uint64_t cap);  // enlarge the (empty) symbol buffer, clear need_flush
int k4_prepare(bce_hip_ctx *c);                     // k4_model.hip: counters to zero, cfg upload
int k3_grow_lists(bce_hip_ctx *c, const EnumCtl &ctl);
uint64_t k3_symbol_capacity(const bce_hip_ctx *c, uint32_t n);   // records between two model flushes for an input of n bytes
void k4_prepin(bce_hip_ctx *c, uint32_t n);         // start pinning the flush slots' host staging for an input of n bytes (threads)
void k4_prepin_join(bce_hip_ctx *c, bool drop);     // wait for those threads (drop: free what they pinned and nobody adopted)
int k4_flush(bce_hip_ctx *c, uint64_t nsym, FlushSlot &slot);         // synchronous: outputs are in slot.h_out on return
int k4_flush_async(bce_hip_ctx *c, uint64_t nsym, FlushSlot &slot);   // outputs are in slot.h_out once slot.ev_copy has fired   // sort + replay + D2H into the slot (synchronous)

// radix sort (radix_sort.hip): stable LSD sort of (key,val) u32 pairs on key bits [first_bit, first_bit+bits).
// Result is left in key[res]/val[res]; returns res (0 or 1) through *res.
int radix_sort_pairs(bce_hip_ctx *c, uint32_t *key[2], uint32_t *val[2], uint32_t n, uint32_t first_bit, uint32_t bits, int *res,
                     uint32_t max_digit_bits = 8);
int radix_sort_pairs_on(bce_hip_ctx *c, hipStream_t stream, DevBuf &hist, uint32_t *key[2], uint32_t *val[2], uint32_t n,
                        uint32_t first_bit, uint32_t bits, int *res, uint32_t max_digit_bits = 8);
// the same with 64-bit keys in two u32 arrays (hi:lo) on key bits [0, bits), bits <= 64 (K1's first sort)
int radix_sort_wide(bce_hip_ctx *c, uint32_t *lo[2], uint32_t *hi[2], uint32_t *val[2], uint32_t n, uint32_t bits, int *res,
                    uint32_t max_digit_bits = 10);
bool k4_in_flight(bce_hip_ctx *c);                  // a model flush may still be running beside the main stream

}  // namespace bce
