// api.hip -- the C ABI of libbcehip.so (include/bce_hip.h): context lifetime, stage entry points and
// the host-side driver of BCE::encode (bce.cpp:1117-1167).
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <new>
#include <utility>

#include "common.h"
#include "host_coder.h"
#include "scan_coder.h"

#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <sys/file.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

using namespace bce;

#ifndef K3_BATCH_MID
#define K3_BATCH_MID 8u   // rounds queued per host sync while 1 M < nodes <= 4 M
#endif

namespace bce {
namespace {
struct RoctxApi {
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
  RoctxApi() {
    for (const char *lib : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
      void *h = dlopen(lib, RTLD_NOW | RTLD_LOCAL);
      if (!h) continue;
      push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
      pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
      if (push && pop) return;
      push = nullptr; pop = nullptr;
    }
  }
};
const RoctxApi &roctx() { static const RoctxApi api; return api; }
}  // namespace
void *big_host_alloc(size_t bytes, int device, bool *registered) {
  *registered = false;
  void *q = nullptr;
  // (only for the one large buffer; always a mapping of its own: common.h, "host memory registered with the runtime")
  if (!getenv("BCE_DEC_NO_HUGE") && bytes >= reg_min_bytes((size_t)256 << 20)) {
    q = huge_map(bytes);
  }
  if (q) {
    // first touch on a few threads, then the pages are registered where they are
    constexpr unsigned nt = 4;
    auto touch = [q, bytes](unsigned t) {
      const size_t lo = bytes / nt * t, hi = t + 1 == nt ? bytes : bytes / nt * (t + 1);
      for (size_t o = lo; o < hi; o += 4096) static_cast<volatile uint8_t *>(q)[o] = 0;
    };
    std::vector<std::thread> th;
    try {
      for (unsigned t = 1; t < nt; ++t) th.emplace_back(touch, t);
    } catch (...) {}
    const unsigned started = (unsigned)th.size() + 1;
    touch(0);
    for (unsigned t = started; t < nt; ++t) touch(t);            // (threads that did not start)
    for (auto &x : th) x.join();
    reg_map_settle(q, bytes);
    // registered memory is mapped and host-coherent; the device address must be the host's (the copies are given either)
    void *dp = nullptr;
    if (hipSetDevice(device) == hipSuccess && hipHostRegister(q, bytes, hipHostRegisterMapped) == hipSuccess) {
      if (hipHostGetDevicePointer(&dp, q, 0) == hipSuccess && dp == q) { *registered = true; return q; }
      (void)hipHostUnregister(q);
    }
    (void)hipGetLastError();
    huge_unmap(q, bytes);
    q = nullptr;
  }
  if (hipSetDevice(device) != hipSuccess || hipHostMalloc(&q, bytes, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return q;
}
void big_host_free(bce_hip_ctx *c, void *p, size_t bytes, bool registered) {
  if (!p) return;
  if (registered) {
    // hipHostFree waits for the device before it lets go of the memory; unregistering does not.  The only work that touches
    // this buffer are copies on its context's own streams: those are waited for, no other context is stalled.
    if (c) {
      (void)hipSetDevice(c->device);
      for (hipStream_t st : {c->stream, c->k4_stream, c->copy_stream}) if (st) (void)hipStreamSynchronize(st);
    } else (void)hipDeviceSynchronize();
    (void)hipHostUnregister(p);
    huge_unmap(p, bytes);
    if (c) c->reg_unmaps++;
  } else (void)hipHostFree(p);
}

// Out of device memory while phase `c->phase` runs: what only OTHER phases use goes back (the context keeps every stage's buffers
// from one input to the next -- K1's ~45 n bytes of sort scratch beside node lists of up to 72 n bytes do not fit at n ~ 2^31).
// Nothing that is given back can be in use: K1 and K2 end with their work complete, the enumeration starts after them, and a
// decode owns the whole context.
size_t ctx_trim(bce_hip_ctx *c) {
  std::vector<DevBuf *> give;
  if (c->phase == 3) {
    // the enumeration and the model keep the text, the BWT, the suffix array and its inverse (the depth-first tail), the planes
    give = {&c->sa[c->sa_res ^ 1], &c->key[0], &c->key[1], &c->k2, &c->nrk, &c->act[0], &c->act[1], &c->khi[0], &c->khi[1],
            &c->dl[0], &c->dl[1], &c->dl[2], &c->dl[3], &c->kflag, &c->actv[0], &c->actv[1], &c->ptmp[0], &c->ptmp[1]};
  } else if (c->phase == 1 || c->phase == 2) {
    give = {&c->nlist[0], &c->nlist[1], &c->dfs, &c->tilecnt, &c->tileoff, &c->k3tw, &c->k3grp, &c->skey[0], &c->skey[1], &c->sval[0],
            &c->sval[1], &c->sout, &c->sesc, &c->skey_alt, &c->sesc_alt, &c->scanrec, &c->k4w};
    if (c->phase == 1) give.push_back(&c->gran);
  } else if (c->phase == 4) {
    // a decode uses the suffix arrays, the low key words, the rank array, the record arrays and nlist[0] as its scratch; the rest
    // of the encoder's buffers it never touches
    give = {&c->nlist[1], &c->sout, &c->k4w, &c->scanrec, &c->skey_alt, &c->sesc_alt, &c->khi[0], &c->khi[1], &c->k2, &c->nrk, &c->act[0], &c->act[1],
            &c->actv[0], &c->actv[1], &c->kflag, &c->dl[0], &c->dl[1], &c->dl[2], &c->dl[3], &c->ptmp[0], &c->ptmp[1], &c->sval[0], &c->sval[1],
            &c->k3tw, &c->k3grp};
  }
  size_t freed = 0;
  bool synced = false;
  for (DevBuf *b : give) {
    if (!b->p) continue;
    if (!synced) { for (hipStream_t st : {c->stream, c->k4_stream, c->copy_stream}) if (st) (void)hipStreamSynchronize(st); synced = true; }
    freed += b->cap;
    release(*b);
  }
  if (freed && getenv("BCE_ALLOC_TRACE")) fprintf(stderr, "alloc: out of device memory in phase %d: %.1f GB of other phases' buffers given back\n", c->phase, freed / 1e9);
  return freed;
}

RoctxRange::RoctxRange(const char *name) : on_(roctx().push != nullptr) { if (on_) roctx().push(name); }
RoctxRange::~RoctxRange() { if (on_) roctx().pop(); }
}  // namespace bce

namespace {

int check_stage(bce_hip_ctx *c, int need) {
  if (!c) return BCE_HIP_E_ARG;
  if (c->stage < need) { snprintf(c->err, sizeof c->err, "stage %d required, have %d", need, c->stage); return BCE_HIP_E_STATE; }
  return BCE_HIP_OK;
}

// The device gate (bce_hip_set_gated).  A compression is a GPU phase (K1..K4) followed by a host phase in which the
// coder threads finish the last batches; other contexts can use the GPU meanwhile, and K1 / K2 of one context run
// beside the enumeration of another.  What must NOT run side by side is the enumeration (K3) of two contexts: its
// single-launch rounds spin on predecessor tiles and count on their grid becoming resident as a whole
// (k3_enumerate.hip); two such grids can fill the XCDs with blocks that wait for blocks of their own kernel that
// found no room.  So gated contexts of one device hold the gate from the first round of bce_hip_encode to its last
// model flush, and lend it to the next context whenever they would wait for their own coder threads.
// (Measured on MI355X, 10^8-byte inputs, two contexts: text 1096 -> 1256 MB/s, natural corpus 1065 -> 1200, binary corpus
//  921 -> 1095 against a gate around the whole GPU phase; one context at a time: 758 / 887 / 725.)
// Between PROCESSES (two `bce -c` on one GPU, a pool beside another program's context) the same rule is kept by an
// advisory file lock per device, keyed by the PCI address (/dev/shm/bce_hip_gate_<bdf>, flock): taken after the
// in-process gate, given back with it.  A box without /dev/shm (or without permission) runs without the file lock.
struct DeviceGate { std::mutex mu; std::condition_variable cv; bool busy = false; int waiting = 0; int fd = -2; };
DeviceGate &device_gate(int device) {
  static DeviceGate gates[64];
  return gates[device >= 0 && device < 64 ? device : 0];
}
int gate_file(DeviceGate &g, int device) {          // (called with the in-process gate held: one thread at a time)
  if (g.fd != -2) return g.fd;
  g.fd = -1;
  if (getenv("BCE_HIP_NO_FILE_GATE")) return g.fd;
  char bdf[64] = {0};
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf - 1, device) != hipSuccess) { (void)hipGetLastError(); snprintf(bdf, sizeof bdf, "dev%d", device); }
  for (char *q = bdf; *q; ++q) if (*q == ':' || *q == '.' || *q == '/') *q = '_';
  char path[160];
  snprintf(path, sizeof path, "/dev/shm/bce_hip_gate_%s", bdf);
  const mode_t um = umask(0);
  g.fd = open(path, O_RDWR | O_CREAT | O_CLOEXEC, 0666);
  umask(um);
  return g.fd;
}
void gate_acquire(bce_hip_ctx *c) {
  if (!c->gated || c->gate_held) return;
  DeviceGate &g = device_gate(c->device);
  const double tw0 = now_s();
  {
    std::unique_lock<std::mutex> lk(g.mu);
    ++g.waiting;
    g.cv.wait(lk, [&] { return !g.busy; });
    --g.waiting;
    g.busy = true;
  }
  const int fd = gate_file(g, c->device);
  if (fd >= 0) while (flock(fd, LOCK_EX) != 0 && errno == EINTR) {}
  c->gate_held = true;
  c->gate_t0 = now_s();
  c->gate_wait_s += c->gate_t0 - tw0;
}
void gate_release(bce_hip_ctx *c) {
  if (!c->gate_held) return;
  DeviceGate &g = device_gate(c->device);
  if (g.fd >= 0) (void)flock(g.fd, LOCK_UN);
  { std::lock_guard<std::mutex> lk(g.mu); g.busy = false; }
  c->gate_held = false;
  c->gate_held_s += now_s() - c->gate_t0;
  g.cv.notify_one();
}
// Another context of this process waits for the gate (contexts of other processes, behind the file lock, are not seen).
bool gate_contended(bce_hip_ctx *c) {
  DeviceGate &g = device_gate(c->device);
  std::lock_guard<std::mutex> lk(g.mu);
  return g.waiting > 0;
}
// The gate is for the enumeration's kernels (the one-launch rounds spin on tiles of their own grid); the model's kernels wait
// for nothing, so a context whose rounds have all ended (every batch ends with a sync on the control block) and whose model
// flush is queued LENDS the gate to a context that waits for it, and takes it back -- once its flush's kernels are through,
// so that it does not hold the gate over kernels that do not need it -- before it queues its next round (gate_regain).
// Measured, three contexts on text: the gate was held 42-71 ms per input (K3 13-20 ms + the flushes' 18 ms of kernels + what the
// contexts cost each other) and the step of the stream was 60 ms.  BCE_HIP_NO_GATE_LEND=1: as before.
void gate_lend(bce_hip_ctx *c) {
  static const bool no_lend = getenv("BCE_HIP_NO_GATE_LEND") != nullptr;
  if (no_lend || !c->gate_held || !gate_contended(c)) return;
  gate_release(c);
  c->gate_lent = true;
}
int gate_regain(bce_hip_ctx *c) {
  if (!c->gate_lent) return BCE_HIP_OK;
  c->gate_lent = false;
  BCE_HIP_TRY(c, hipEventSynchronize(c->ev_k4));
  gate_acquire(c);
  BCE_HIP_TRY(c, hipSetDevice(c->device));
  return BCE_HIP_OK;
}
// a stage that failed gives the gate back: the caller will not get to the point where encode does
int gate_on_error(bce_hip_ctx *c, int status) {
  if (c && status != BCE_HIP_OK) gate_release(c);
  return status;
}

// A bounded wait for a tagged word ran out (wait_word, k3_enumerate.hip): the round is void, later kernels no-op.  Reported
// before anything of the control block is used -- a stalled round sets overflow (so that later launches return at once) but
// not skip_round, and "rounds executed" computed from a stale skip_round once sized a copy out of the run table.
int k3_stalled(bce_hip_ctx *c) {
  snprintf(c->err, sizeof c->err, "k3: a single-launch round waited too long for a predecessor tile (dispatch order not as assumed)");
  return BCE_HIP_E_INTERNAL;
}
int k3_bad_skip(bce_hip_ctx *c, const EnumCtl &ctl, uint32_t first, uint32_t batch) {
  snprintf(c->err, sizeof c->err, "k3: control block names round %u as skipped, outside the batch %u..%u", ctl.skip_round, first, first + batch);
  return BCE_HIP_E_INTERNAL;
}

int set_input_body(bce_hip_ctx *c, const void *src, uint32_t n, hipMemcpyKind kind);
int set_input(bce_hip_ctx *c, const void *src, uint32_t n, hipMemcpyKind kind) {
  if (!c || !src || n == 0 || n >= 0x80000000u) return BCE_HIP_E_ARG;   // n < 2^31 (saidx_t, getv: SURVEY section 5)
  if (getenv("BCE_HIP_GATE_ALL")) gate_acquire(c);      // (experiment: the whole GPU phase exclusive, not only the enumeration)
  return gate_on_error(c, set_input_body(c, src, n, kind));
}
int set_input_body(bce_hip_ctx *c, const void *src, uint32_t n, hipMemcpyKind kind) {
  BCE_HIP_TRY(c, hipSetDevice(c->device));
  const double t0 = now_s();
  BCE_TRY(ensure(c, c->text, n));
  BCE_HIP_TRY(c, hipMemcpyAsync(c->text.p, src, n, kind, c->stream));
  BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->n = n;
  c->stage = 1;
  c->enum_active = false;
  memset(&c->stats, 0, sizeof c->stats);
  c->stats.n = n;
  c->stats.t_load = now_s() - t0;
  return BCE_HIP_OK;
}

// GPU time of a finished flush (K4 kernels + copy), from its events
void account_slot(bce_hip_ctx *c, FlushSlot &slot) {
  if (!slot.timed) return;
  float ms = 0;
  if (hipEventElapsedTime(&ms, slot.ev_start, slot.ev_copy) == hipSuccess) c->stats.t_model += ms * 1e-3;
  if (slot.ev_kend && hipEventElapsedTime(&ms, slot.ev_start, slot.ev_kend) == hipSuccess) c->stats.t_model_kernels += ms * 1e-3;
  slot.timed = false;
}

// Flush the model (K4) for the symbols buffered so far and hand the result to the coder threads.  Nothing here
// waits for the GPU: the kernels are queued on the compute stream, the device-to-host copy on the copy stream,
// and the coder threads wait for the copy's event before they touch the batch; the next rounds overlap both.
int flush_symbols(bce_hip_ctx *c, uint64_t nsym) {
  if (nsym) {
    FlushSlot &slot = c->slot[c->slot_next];
    c->slot_next = (c->slot_next + 1) % 3;
    double t0 = now_s();
    // the slot's previous batch must be fully coded (so its copy is done); a gated context gives the GPU away while it
    // waits (no round of its own is in flight here: every batch of rounds ends with a sync on the control block)
    const bool handover = c->gate_held && slot.batch.pending.load() != 0;
    if (handover) gate_release(c);
    c->coder->wait(&slot.batch);
    if (handover) { gate_acquire(c); BCE_HIP_TRY(c, hipSetDevice(c->device)); }
    c->stats.t_coder += now_s() - t0;
    account_slot(c, slot);
    const uint32_t seq0 = c->flush_seq;
    RoctxRange range("bce K4 model flush");
    BCE_TRY(k4_flush_async(c, nsym, slot));
    if (c->flush_seq != seq0) {
      // the flush runs beside the main stream and reads skey[0] / sesc: the next rounds write the other pair, which
      // the flush BEFORE this one was the last to read
      std::swap(c->skey[0], c->skey_alt);
      std::swap(c->sesc, c->sesc_alt);
      if (seq0 > 0) BCE_HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_k4_done[(seq0 - 1u) & 1u], 0));
    }
    // BCE_HIP_SYNC_FLUSH=1 (profiling): wait for the copy before queuing more rounds.  rocprofv3's kernel trace
    // serialises the copy stream's blit kernel with the compute stream and charges the 2.4 ms to the K3 kernel behind it.
    if (c->sync_flush) BCE_HIP_TRY(c, hipEventSynchronize(slot.ev_copy));
    slot.batch.out = slot.h_out;
    for (int p = 0; p < 8; ++p) {
      slot.batch.runs[p].clear();
      slot.batch.runs[p].reserve(c->run_log[p].size());
      for (const RunEntry &e : c->run_log[p]) slot.batch.runs[p].push_back(SymRun{e.start, e.count, e.round});
    }
    slot.once = std::make_shared<std::once_flag>();
    {
      std::shared_ptr<std::once_flag> once = slot.once;
      hipEvent_t ev = slot.ev_copy;
      slot.batch.wait_ready = [once, ev]() { std::call_once(*once, [ev]() { (void)hipEventSynchronize(ev); }); };
    }
    c->coder->submit(&slot.batch);
    c->stats.flushes++;
    c->stats.symbols += nsym;
    gate_lend(c);
  }
  return k3_reset_symbols(c);
}

}  // namespace

extern "C" {

static int create_body(bce_hip_ctx **out, int device, uint64_t expected);
int bce_hip_create(bce_hip_ctx **out, int device) { return bce_hip_create_sized(out, device, 0); }
int bce_hip_create_sized(bce_hip_ctx **out, int device, uint64_t expected_input_bytes) {
  if (!out) return BCE_HIP_E_ARG;
  *out = nullptr;
  // (HostCoder starts 8 threads: std::system_error must not cross the C ABI)
  return bce_guarded(nullptr, [&] { return create_body(out, device, expected_input_bytes); });
}
static void stage_release(bce_hip_ctx *c, int state) {
  { std::lock_guard<std::mutex> lk(c->stage_mu); c->stage_state = state; }
  c->stage_cv.notify_all();
}
static int create_body(bce_hip_ctx **out, int device, uint64_t expected) {
  bce_hip_ctx *c = new (std::nothrow) bce_hip_ctx();
  if (!c) return BCE_HIP_E_NOMEM;
  c->device = device;
  memcpy(c->config, kDefaultConfig, sizeof c->config);
  memset(&c->stats, 0, sizeof c->stats);
  // a one-shot caller that knows its input's size: the flush slots' host staging is allocated and touched on threads of
  // its own while this thread brings the runtime up (k4_prepin); the threads register it once the runtime is there
  if (expected > 0 && expected < 0x80000000ull) { c->stage_state = 0; k4_prepin(c, (uint32_t)expected); }
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) {
    stage_release(c, 2);
    k4_prepin_join(c, true);
    delete c;
    return BCE_HIP_E_DEVICE;
  }
  stage_release(c, 1);
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess || hipEventCreate(&c->ev_k4) != hipSuccess ||
      hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) {
    k4_prepin_join(c, true);
    delete c;
    return BCE_HIP_E_DEVICE;
  }
  c->sync_flush = getenv("BCE_HIP_SYNC_FLUSH") != nullptr;
  if (getenv("BCE_HIP_NO_FUSED")) c->dbg_no_fused = 1;             // three launches per wide round (as debug knob 6)
  // (measured on MI355X: K3 and K4 each fill the chip, so running them side by side moves no end-to-end number -- natural
  //  corpus 114.4 vs 113.8 ms, binary 192 vs 190 -- while the K3 kernels take 14 -> 18 ms on text: off unless asked for)
  c->overlap = getenv("BCE_HIP_OVERLAP") != nullptr;
  {
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);      // K4 yields to K3 wherever the two compete
    if (hipStreamCreateWithPriority(&c->k4_stream, hipStreamNonBlocking, least) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_k3_batch, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_k4_done[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_k4_done[1], hipEventDisableTiming) != hipSuccess) {
      bce_hip_destroy(c);
      return BCE_HIP_E_DEVICE;
    }
  }
  try { c->coder = new HostCoder(); }
  catch (...) { c->coder = nullptr; }
  if (!c->coder) { bce_hip_destroy(c); return BCE_HIP_E_NOMEM; }
  *out = c;
  return BCE_HIP_OK;
}

void bce_hip_destroy(bce_hip_ctx *c) {
  if (!c) return;
  gate_release(c);
  (void)hipSetDevice(c->device);
  if (c->coder) c->coder->drain();
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->k4_stream) (void)hipStreamSynchronize(c->k4_stream);
  if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
  DevBuf *bufs[] = {&c->text, &c->bwt, &c->sa[0], &c->sa[1], &c->key[0], &c->key[1], &c->rank, &c->k2, &c->nrk, &c->act[0], &c->act[1], &c->khi[0], &c->khi[1], &c->dl[0], &c->dl[1], &c->dl[2], &c->dl[3], &c->kflag, &c->actv[0], &c->actv[1],
                    &c->rs_hist, &c->blk, &c->ptmp[0], &c->ptmp[1], &c->gran, &c->nlist[0], &c->nlist[1], &c->ctl, &c->tilecnt,
                    &c->tileoff, &c->runs, &c->smwords, &c->k3tw, &c->k3grp, &c->truns, &c->skey[0], &c->skey[1], &c->sval[0], &c->sval[1], &c->sout,
                    &c->sesc, &c->stat, &c->dcfg, &c->k4w, &c->scanrec, &c->dfs, &c->skey_alt, &c->sesc_alt, &c->rs_hist_k4};
  for (DevBuf *b : bufs) release(*b);
  k4_prepin_join(c, true);
  if (c->h_ctl) (void)hipHostFree(c->h_ctl);
  if (c->h_small) (void)hipHostFree(c->h_small);
  if (c->h_big) big_host_free(c, c->h_big, c->h_big_cap, c->h_big_registered);
  for (void *q : c->dec_pin) if (q) (void)hipHostFree(q);
  if (c->h_runs) (void)hipHostFree(c->h_runs);
  if (c->h_truns) (void)hipHostFree(c->h_truns);
  if (c->coder) c->coder->drain();
  for (FlushSlot &sl : c->slot) {
    slot_free_host(sl, &c->reg_unmaps);
    if (sl.ev_start) (void)hipEventDestroy(sl.ev_start);
    if (sl.ev_copy) (void)hipEventDestroy(sl.ev_copy);
    if (sl.ev_kend) (void)hipEventDestroy(sl.ev_kend);
  }
  if (c->ev_k4) (void)hipEventDestroy(c->ev_k4);
  if (c->ev_k3_batch) (void)hipEventDestroy(c->ev_k3_batch);
  for (hipEvent_t e : c->ev_k4_done) if (e) (void)hipEventDestroy(e);
  if (c->k4_stream) (void)hipStreamDestroy(c->k4_stream);
  if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c->coder;
  delete c;
}

const char *bce_hip_strerror(int status) {
  switch (status) {
    case BCE_HIP_OK: return "ok";
    case BCE_HIP_E_ARG: return "bad argument";
    case BCE_HIP_E_DEVICE: return "HIP device/runtime error";
    case BCE_HIP_E_NOMEM: return "out of memory";
    case BCE_HIP_E_STATE: return "stage called out of order";
    case BCE_HIP_E_OVERFLOW: return "buffer capacity exceeded";
    case BCE_HIP_E_INTERNAL: return "internal consistency check failed";
    default: return "unknown status";
  }
}
const char *bce_hip_last_error(const bce_hip_ctx *c) { return c ? c->err : ""; }

int bce_hip_set_config(bce_hip_ctx *c, const uint8_t *config288) {
  if (!c) return BCE_HIP_E_ARG;
  if (config288) {
    for (uint32_t i = 0; i < BCE_HIP_CONFIG_BYTES; ++i)
      if (config288[i] > 5) {
        // context bits above 5 cannot be serialised by the preamble (set(bit, 6), bce.cpp:689)
        snprintf(c->err, sizeof c->err, "config entry %u = %u > 5", i, config288[i]);
        return BCE_HIP_E_ARG;
      }
    memcpy(c->config, config288, BCE_HIP_CONFIG_BYTES);
  } else {
    memcpy(c->config, kDefaultConfig, sizeof c->config);
  }
  return BCE_HIP_OK;
}

int bce_hip_debug_set(bce_hip_ctx *c, int knob, uint32_t value) {
  if (!c) return BCE_HIP_E_ARG;
  switch (knob) {
    case 0: c->dbg_dfs_budget = value; break;
    case 1: c->dbg_no_dfs = value; break;
    case 4: c->dbg_no_small = value; break;
    case 5: c->dbg_step_small = value; break;
    case 6: c->dbg_no_fused = value; break;
    case 7: c->dbg_no_local = value; break;
    case 8: c->dbg_local_from = value; break;
    case 9: c->dbg_local_budget = value; break;
    case 10: c->dbg_tail_round = value; break;
    case 11: c->overlap = value != 0; break;
    case 12: c->dbg_capp_div = value; break;
    case 2: c->dbg_no_tail = value; break;
    case 3: c->dbg_no_skip = value; break;
    default: return BCE_HIP_E_ARG;
  }
  return BCE_HIP_OK;
}

int bce_hip_set_progress(bce_hip_ctx *c, bce_hip_progress_fn fn, void *user) {
  if (!c) return BCE_HIP_E_ARG;
  c->progress = fn;
  c->progress_user = user;
  return BCE_HIP_OK;
}

int bce_hip_set_gated(bce_hip_ctx *c, int on) {
  if (!c) return BCE_HIP_E_ARG;
  gate_release(c);                               // either way a gate this context still holds is given back
  c->gated = on != 0;
  return BCE_HIP_OK;
}

int bce_hip_set_symbol_capacity(bce_hip_ctx *c, uint64_t records) {
  if (!c || records >= (1ull << 31)) return BCE_HIP_E_ARG;
  c->sym_cap_user = records;
  return BCE_HIP_OK;
}

int bce_hip_load_host(bce_hip_ctx *c, const uint8_t *in, uint32_t n) { return set_input(c, in, n, hipMemcpyHostToDevice); }
int bce_hip_load_device(bce_hip_ctx *c, const void *d_in, uint32_t n) { return set_input(c, d_in, n, hipMemcpyDeviceToDevice); }

static int bwt_body(bce_hip_ctx *c, uint32_t *offset);
int bce_hip_bwt(bce_hip_ctx *c, uint32_t *offset) { return gate_on_error(c, bwt_body(c, offset)); }
static int bwt_body(bce_hip_ctx *c, uint32_t *offset) {
  BCE_TRY(check_stage(c, 1));
  BCE_HIP_TRY(c, hipSetDevice(c->device));
  const double t0 = now_s();
  RoctxRange range("bce K1 rotation sort + BWT");
  c->phase = 1;
  struct PhaseEnd { bce_hip_ctx *c; ~PhaseEnd() { c->phase = 0; } } phase_end{c};
  BCE_TRY(k1_bwt(c));
  c->stats.t_bwt = now_s() - t0;
  c->stage = 2;
  if (offset) *offset = c->offset;
  return BCE_HIP_OK;
}

int bce_hip_set_bwt(bce_hip_ctx *c, const uint8_t *bwt, uint32_t n, uint32_t offset) {
  if (!c || !bwt || n == 0 || n >= 0x80000000u || offset >= n) return BCE_HIP_E_ARG;
  BCE_HIP_TRY(c, hipSetDevice(c->device));
  BCE_TRY(ensure(c, c->bwt, n));
  BCE_HIP_TRY(c, hipMemcpy(c->bwt.p, bwt, n, hipMemcpyHostToDevice));
  c->n = n; c->offset = offset; c->stage = 2; c->enum_active = false;
  c->k1_unique = false; c->k1_valid = false;     // no suffix array behind an injected BWT
  memset(&c->stats, 0, sizeof c->stats);
  c->stats.n = n;
  return BCE_HIP_OK;
}

int bce_hip_get_bwt(bce_hip_ctx *c, uint8_t *out) {
  BCE_TRY(check_stage(c, 2));
  if (!out) return BCE_HIP_E_ARG;
  BCE_HIP_TRY(c, hipSetDevice(c->device));
  BCE_HIP_TRY(c, hipMemcpy(out, c->bwt.p, c->n, hipMemcpyDeviceToHost));
  return BCE_HIP_OK;
}

int bce_hip_divbwt(bce_hip_ctx *c, const uint8_t *in, uint8_t *out, uint32_t n, uint32_t *primary) {
  if (!c || !in || !out || !primary || n == 0) return BCE_HIP_E_ARG;
  return bce_guarded(c, [&]() -> int {
    BCE_HIP_TRY(c, hipSetDevice(c->device));
    c->coder->drain();
    c->phase = 1;
    struct PhaseEnd { bce_hip_ctx *c; ~PhaseEnd() { c->phase = 0; } } phase_end{c};
    return k1_divbwt(c, in, out, n, primary);
  });
}

int bce_hip_inverse_bwt(bce_hip_ctx *c, const uint8_t *in, uint8_t *out, uint32_t n, uint32_t primary) {
  if (!c || !in || !out || n == 0) return BCE_HIP_E_ARG;
  return bce_guarded(c, [&]() -> int {
    BCE_HIP_TRY(c, hipSetDevice(c->device));
    c->coder->drain();
    return kd_inverse_bw_transform(c, in, out, n, primary);
  });
}

static int planes_body(bce_hip_ctx *c, uint32_t zeros[8]);
int bce_hip_build_planes(bce_hip_ctx *c, uint32_t zeros[8]) { return gate_on_error(c, planes_body(c, zeros)); }
static int planes_body(bce_hip_ctx *c, uint32_t zeros[8]) {
  BCE_TRY(check_stage(c, 2));
  BCE_HIP_TRY(c, hipSetDevice(c->device));
  const double t0 = now_s();
  RoctxRange range("bce K2 planes + rank directory");
  c->phase = 2;
  struct PhaseEnd { bce_hip_ctx *c; ~PhaseEnd() { c->phase = 0; } } phase_end{c};
  BCE_TRY(k2_build_planes(c));
  c->stats.t_planes = now_s() - t0;
  c->stage = 3;
  if (zeros) memcpy(zeros, c->zeros, sizeof c->zeros);
  return BCE_HIP_OK;
}

int bce_hip_get_plane_bits(bce_hip_ctx *c, int plane, uint8_t *out) {
  BCE_TRY(check_stage(c, 3));
  if (!out || plane < 0 || plane > 7) return BCE_HIP_E_ARG;
  BCE_HIP_TRY(c, hipSetDevice(c->device));
  return k2_get_plane_bits(c, plane, out);
}

int bce_hip_rank1(bce_hip_ctx *c, int plane, const uint32_t *idx, uint32_t count, uint32_t *out) {
  BCE_TRY(check_stage(c, 3));
  if (!idx || !out || plane < 0 || plane > 7) return BCE_HIP_E_ARG;
  for (uint32_t i = 0; i < count; ++i) if (idx[i] > c->n) return BCE_HIP_E_ARG;
  BCE_HIP_TRY(c, hipSetDevice(c->device));
  return k2_rank1(c, plane, idx, count, out);
}

// ---- stepping interface ---------------------------------------------------------------------------
int bce_hip_enum_begin(bce_hip_ctx *c) {
  BCE_TRY(check_stage(c, 3));
  c->phase = 3;
  BCE_HIP_TRY(c, hipSetDevice(c->device));
  BCE_TRY(k4_prepare(c));
  BCE_TRY(k3_begin(c));
  return BCE_HIP_OK;
}

int bce_hip_enum_nodes(bce_hip_ctx *c, int plane, uint32_t *out, uint32_t cap_nodes, uint32_t *count) {
  if (!c || !c->enum_active || !out || !count || plane < 0 || plane > 7) return BCE_HIP_E_ARG;
  BCE_HIP_TRY(c, hipSetDevice(c->device));
  return k3_get_nodes(c, plane, out, cap_nodes, count);
}

int bce_hip_enum_round(bce_hip_ctx *c, uint64_t *next_nodes) {
  if (!c || !c->enum_active) return BCE_HIP_E_STATE;
  BCE_HIP_TRY(c, hipSetDevice(c->device));
  const uint32_t first = c->round;
  if (c->dbg_step_small) BCE_TRY(k3_rounds_small(c, 1, K3_SMALL_NODES, false));   // test hook: step with the one-launch kernel
  else BCE_TRY(k3_rounds(c, 1, 0));
  EnumCtl ctl;
  BCE_TRY(k3_sync_ctl(c, &ctl));
  if (ctl.stalled) return k3_stalled(c);
  if (ctl.overflow) {                              // larger lists, the same round again (as bce_hip_encode does)
    BCE_TRY(k3_grow_lists(c, ctl));
    if (c->dbg_step_small) BCE_TRY(k3_rounds_small(c, 1, K3_SMALL_NODES, false));
    else BCE_TRY(k3_rounds(c, 1, 0));
    BCE_TRY(k3_sync_ctl(c, &ctl));
    if (ctl.overflow) return BCE_HIP_E_OVERFLOW;
  }
  if (ctl.need_flush) return BCE_HIP_E_OVERFLOW;   // the stepping interface never flushes
  BCE_TRY(k3_fetch_runs(c, first, 1));
  c->round = first + 1;
  if (next_nodes) *next_nodes = ctl.next_nodes;
  return BCE_HIP_OK;
}

int bce_hip_enum_symbols(bce_hip_ctx *c, uint32_t *out, uint64_t cap_records, uint64_t *count) {
  if (!c || !c->enum_active || !out || !count) return BCE_HIP_E_ARG;
  BCE_HIP_TRY(c, hipSetDevice(c->device));
  EnumCtl ctl;
  BCE_TRY(k3_sync_ctl(c, &ctl));
  *count = ctl.sym_total;
  if (ctl.sym_total > cap_records) return BCE_HIP_E_OVERFLOW;
  std::vector<uint32_t> kw(ctl.sym_total), ew(ctl.sym_total);
  if (ctl.sym_total) {
    BCE_HIP_TRY(c, hipMemcpy(kw.data(), c->skey[0].p, ctl.sym_total * 4, hipMemcpyDeviceToHost));
    BCE_HIP_TRY(c, hipMemcpy(ew.data(), c->sesc.p, ctl.sym_total * 4, hipMemcpyDeviceToHost));
  }
  for (uint64_t i = 0; i < ctl.sym_total; ++i) {
    out[6 * i + 0] = key_plane(kw[i]); out[6 * i + 1] = key_sym(kw[i]); out[6 * i + 2] = key_k(kw[i]);
    out[6 * i + 3] = esc_n(ew[i]); out[6 * i + 4] = esc_bits(ew[i]); out[6 * i + 5] = key_slot(kw[i]);
  }
  return BCE_HIP_OK;
}

int bce_hip_enum_model(bce_hip_ctx *c, uint32_t *out, uint64_t cap_records, uint64_t *count) {
  if (!c || !c->enum_active || !out || !count) return BCE_HIP_E_ARG;
  BCE_HIP_TRY(c, hipSetDevice(c->device));
  EnumCtl ctl;
  BCE_TRY(k3_sync_ctl(c, &ctl));
  *count = ctl.sym_total;
  if (ctl.sym_total > cap_records) return BCE_HIP_E_OVERFLOW;
  c->coder->drain();
  BCE_TRY(k4_flush(c, ctl.sym_total, c->slot[0]));
  for (uint64_t i = 0; i < ctl.sym_total; ++i) {
    const uint64_t o = c->slot[0].h_out[i];
    out[3 * i + 0] = out_cum(o); out[3 * i + 1] = out_freq(o); out[3 * i + 2] = out_total(o);
  }
  BCE_TRY(k3_reset_symbols(c));
  return BCE_HIP_OK;
}

// ---- BCE::encode ------------------------------------------------------------------------------------
static int encode_body(bce_hip_ctx *c);
static int enumerate_body(bce_hip_ctx *c, EnumCtl &ctl, const std::function<int(uint64_t)> &sink);
static int split_round(bce_hip_ctx *c, EnumCtl &ctl, const std::function<int(uint64_t)> &sink, bool *executed);
int bce_hip_encode(bce_hip_ctx *c) { return gate_on_error(c, bce_guarded(c, [&] { return encode_body(c); })); }
static int encode_body(bce_hip_ctx *c) {
  BCE_TRY(check_stage(c, 3));
  c->phase = 3;
  struct PhaseEnd { bce_hip_ctx *c; ~PhaseEnd() { c->phase = 0; } } phase_end{c};
  c->gate_wait_s = 0; c->gate_held_s = 0; c->gate_lent = false;
  const double t_enc0 = now_s();
  gate_acquire(c);
  BCE_HIP_TRY(c, hipSetDevice(c->device));
  const uint32_t n = c->n;
  k4_prepin(c, n);                                 // (the staged interface: the one-shot entry points have started it at the load)
  BCE_TRY(k4_prepare(c));
  BCE_TRY(k3_begin(c));
  uint32_t C[8];
  for (int i = 0; i < 8; ++i) C[i] = c->zeros[(i + 7) & 7];   // bce.cpp:1128
  c->coder->begin(c->config, C, n);
  c->stats.symbols = 0; c->stats.flushes = 0; c->stats.t_model = 0; c->stats.t_model_kernels = 0; c->stats.t_coder = 0;

  EnumCtl ctl;
  BCE_TRY(enumerate_body(c, ctl, [&](uint64_t nsym) { return flush_symbols(c, nsym); }));
  gate_release(c);                               // the GPU phase is over (the last flush and its copy are queued): next context
  c->gate_lent = false;
  static const bool gate_timing = getenv("BCE_HIP_GATE_TIMING") != nullptr;
  const double t_gpu_done = now_s();
  {
    const double tw = now_s();
    c->coder->drain();                           // coding of the last batches (the exposed part)
    c->stats.t_coder += now_s() - tw;
    for (FlushSlot &sl : c->slot) account_slot(c, sl);
  }
  if (c->coder->failed()) { snprintf(c->err, sizeof c->err, "host allocation failed in a range-coder thread"); return BCE_HIP_E_NOMEM; }
  c->stats.t_coder_busy = c->coder->busy_seconds();
  c->stats.rounds = ctl.done_round;
  c->stats.nodes = ctl.nodes_total;
  c->coder->finish(c->config, n, c->offset);     // the archive is laid out by bce_hip_archive_copy, straight into the caller's buffer
  c->enum_active = false;
  c->stage = 4;
  c->stats.t_enum = c->stats.k3_ms * 1e-3;      // GPU time of the enumeration; K4, copies and coding overlap it and each other
  if (gate_timing)
    fprintf(stderr, "gate: ctx %p waited %.1f ms, held %.1f ms of %.1f ms from encode's start to its last flush queued (K3 %.1f ms, model %.1f ms, %llu flushes); coders' tail %.1f ms\n",
            (void *)c, c->gate_wait_s * 1e3, c->gate_held_s * 1e3, (t_gpu_done - t_enc0) * 1e3, c->stats.k3_ms, c->stats.t_model * 1e3,
            (unsigned long long)c->stats.flushes, (now_s() - t_gpu_done) * 1e3);
  return BCE_HIP_OK;
}

// The round loop of BCE::code (bce.cpp:1246-1371) as the host drives it: wide rounds in batches, one-launch rounds, the
// LDS tail kernel, the depth-first tail.  `sink(nsym)` takes the symbol records buffered so far (the model flush of -c,
// the ScanCoders of -s) and leaves the buffer empty.
// A round whose symbols do not fit ONE model flush (K4 takes fewer than 2^31 records; BCE_HIP_SPLIT_SYMS lowers the limit for
// tests).  The planes' streams are independent, so the round is run once per GROUP of planes -- every pass classifies all
// nodes and writes all children (the same values to the same places: the parents' lists are not touched), records only its
// group's symbols (K3Args::pmask) and is flushed on its own; a plane's records of one round never exceed n/2 < 2^30.
// *executed: the round is done (ctl = the control block after its last pass, its symbols flushed); false: its lists were too
// small, they have been grown and the caller runs the round again.
static int split_round(bce_hip_ctx *c, EnumCtl &ctl, const std::function<int(uint64_t)> &sink, bool *executed) {
  *executed = false;
  // (K3's event time covers the passes' launches only: the flushes between them wait for coder threads)
  auto pass = [&](uint32_t mask, bool repeat) -> int {
    BCE_HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    BCE_TRY(k3_clear_need_flush(c));
    BCE_TRY(k3_round_masked(c, mask, repeat));
    BCE_HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    BCE_TRY(k3_sync_ctl(c, &ctl));
    float ms = 0;
    if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) c->stats.k3_ms += ms;
    return BCE_HIP_OK;
  };
  // pass 0, all planes, three launches: either the round fits after all (the one-launch rounds only know an upper bound), or
  // the per-plane counts are in the control block afterwards
  BCE_TRY(pass(0xFFu, false));
  if (ctl.stalled) return k3_stalled(c);
  if (ctl.overflow) { BCE_TRY(k3_grow_lists(c, ctl)); ctl.overflow = 0; return BCE_HIP_OK; }
  if (!ctl.need_flush) {
    BCE_TRY(k3_fetch_runs(c, c->round, 1));
    BCE_TRY(sink(ctl.sym_total));
    ctl.sym_total = 0;
    *executed = true;
    return BCE_HIP_OK;
  }
  uint64_t cnt[8], maxp = 0;
  for (int q = 0; q < 8; ++q) { cnt[q] = ctl.ptot[q][2]; if (cnt[q] > maxp) maxp = cnt[q]; }
  if (maxp + 1024 > c->sym_cap) BCE_TRY(k3_grow_symbols(c, maxp + 1024));      // (< 2^30 + 1025)
  bool first = true;
  for (int q = 0; q < 8;) {
    uint32_t mask = 0;
    uint64_t acc = 0;
    while (q < 8 && (mask == 0 || acc + cnt[q] <= c->sym_cap)) { mask |= 1u << q; acc += cnt[q]; ++q; }
    BCE_TRY(pass(mask, !first));
    if (ctl.stalled) return k3_stalled(c);
    if (ctl.need_flush || ctl.overflow) { snprintf(c->err, sizeof c->err, "k3: pass of planes %#x of round %u does not fit (%llu symbols, room for %llu)", mask, c->round, (unsigned long long)acc, (unsigned long long)c->sym_cap); return BCE_HIP_E_INTERNAL; }
    BCE_TRY(k3_fetch_runs(c, c->round, 1));
    BCE_TRY(sink(ctl.sym_total));
    BCE_TRY(gate_regain(c));
    ctl.sym_total = 0;
    first = false;
  }
  c->stats.split_rounds += 1.0;
  *executed = true;
  return BCE_HIP_OK;
}

static int enumerate_body(bce_hip_ctx *c, EnumCtl &ctl, const std::function<int(uint64_t)> &sink) {
  const uint32_t n = c->n;
  uint64_t split_limit = 1ull << 31;
  if (const char *e = getenv("BCE_HIP_SPLIT_SYMS")) { const uint64_t v = strtoull(e, nullptr, 10); if (v && v < split_limit) split_limit = v; }
  uint64_t cur_nodes = 0;
  for (int i = 0; i < 8; ++i) { const uint32_t Ci = c->zeros[(i + 7) & 7]; cur_nodes += (Ci && n - Ci) ? 1 : 0; }
  bool decaying = false;
  bool have_ctl = false, wide_once = false;
  const uint32_t early_max = 4;                   // early small flushes: 1M, 2M, 4M, 8M records (0..5 measured: +3 % on text, neutral on random data)
  // depth-first tail: first attempt when the live set is small, a second one (if the first ran out of room) when tiny
  // (first attempt: from 1 M live nodes down (512 K .. 8 M measured: 1 M is best on text, source code and executables) the tail starts with workgroup-local rounds, k3_local_kernel; without them
  //  the walkers alone take over at 65 536)
  uint32_t kDfsEnter[2] = {c->dbg_no_local ? 65536u : (1u << 20), 2048u};
  if (const char *e = getenv("BCE_HIP_DFS_ENTER")) kDfsEnter[0] = (uint32_t)strtoul(e, nullptr, 10);
  int dfs_try = 0;
  // A round that finds the symbol buffer too small is a round thrown away: its count kernel has run in full, the rest of
  // the batch's launches return at once (8 launches of ~5 us) and the round runs again after the flush -- ~90 us per
  // flush, 1.2 ms of the 13 ms of text.  The host knows how many symbols the last rounds emitted and how much room is
  // left, so it flushes BEFORE a round that is unlikely to fit and never queues more rounds than the room allows
  // (an estimate: a round that does not fit after all still takes the old way).
  uint64_t est_syms = 0;                           // symbols of a coming round, from the last rounds (0: unknown)
  uint64_t recent_syms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t recent_at = 0;
  static const bool no_predict = getenv("BCE_HIP_NO_FLUSH_PREDICT") != nullptr;
  auto rounds_that_fit = [&](uint32_t batch) -> uint32_t {  // 0: flush first
    if (no_predict || !have_ctl || !est_syms || ctl.need_flush) return batch;
    const uint64_t room = c->sym_cap > ctl.sym_total ? c->sym_cap - ctl.sym_total : 0;
    uint64_t need = decaying ? est_syms + (est_syms >> 4) + 1024 : 2 * est_syms + 1024;   // (ramp-up: a round emits twice the last one's)
    uint64_t acc = 0;
    uint32_t fit = 0;
    while (fit < batch && acc + need <= room) { acc += need; if (!decaying) need *= 2; ++fit; }
    if (!fit && !ctl.sym_total) return batch;              // (an empty buffer that is still too small grows the old way)
    return fit;
  };
  for (;;) {
    BCE_TRY(gate_regain(c));                       // (lent to another context behind the last model flush)
    if (have_ctl && decaying && !ctl.need_flush && ctl.done_round == 0xFFFFFFFFu) {
      // few live nodes and almost everything visited: finish depth-first (k3_dfs.hip).  The walkers' symbols
      // come after everything emitted so far, so flush that first.
      const uint64_t all = 8ull * (n - 1);
      if (dfs_try < 2 && ctl.next_nodes && ctl.next_nodes <= kDfsEnter[dfs_try] && (ctl.nodes_total >= all / 8 || c->round >= 1024u || c->dbg_tail_round)) {
        BCE_TRY(sink(ctl.sym_total));
        BCE_TRY(gate_regain(c));
        ctl.sym_total = 0;
        bool dfs_done = false;
        BCE_HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
        RoctxRange range("bce K3 tail (local rounds + walkers)");
        BCE_TRY(k3_dfs_tail(c, ctl, kDfsEnter[dfs_try], &dfs_done));
        BCE_HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
        BCE_HIP_TRY(c, hipEventSynchronize(c->ev1));
        { float ms = 0; if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) c->stats.k3_ms += ms; }
        if (dfs_done) {
          BCE_TRY(k3_sync_ctl(c, &ctl));
          BCE_TRY(sink(ctl.sym_total));
          break;
        }
        ++dfs_try;                                // out of room (symbols / queue): carry on with rounds
      }
    }
    const uint32_t first = c->round;
    bool runs_fetched = false;
    uint32_t executed = 0;
    RoctxRange range("bce K3 rounds");
    BCE_HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    if ((!c->dbg_no_tail || cur_nodes == 0) && cur_nodes <= K3_TAIL_ENTER) {       // (no node at all, e.g. one byte repeated: only this kernel says "done" then)
      // narrow phase: the persistent single-workgroup kernel loops over rounds on the device
      // (while the depth-first tail may still take over, come back after 1024 rounds: an input that is one long run
      //  or one table has this few nodes from the first round on, and its millions of rounds belong to the walkers)
      BCE_TRY(k3_tail(c, (dfs_try < 2 && !c->dbg_no_dfs) ? 1024u : K3_TAIL_MAXROUNDS));
      BCE_HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
      BCE_TRY(k3_sync_ctl(c, &ctl));
      if (ctl.stalled) return k3_stalled(c);        // (before skip_round is read: a stalled round does not set it)
      executed = ctl.tail_rounds;
      BCE_TRY(k3_fetch_tail_runs(c, executed));
    } else if (!c->dbg_no_small && !wide_once && cur_nodes <= K3_SMALL_NODES) {
      // narrow rounds: one launch per round (k3_small_kernel); while the count still doubles, queue only as many
      // rounds as stay within its range
      uint32_t batch = 64;
      if (!decaying) {
        batch = 1;
        while (batch < 16 && (cur_nodes << (batch + 1)) <= 2ull * K3_SMALL_NODES) ++batch;
        if (have_ctl && c->stats.flushes < early_max) {          // see the wide branch: stop at the next early-flush size
          const uint64_t want = (uint64_t)1 << (20 + c->stats.flushes);
          uint64_t est = ctl.sym_total, nn = cur_nodes;
          uint32_t b = 0;
          while (b < batch && est < want) { est += nn; nn *= 2; ++b; }
          batch = b ? b : 1u;
        }
      }
      {
        const uint32_t fit = rounds_that_fit(batch);
        if (!fit) { BCE_TRY(sink(ctl.sym_total)); ctl.sym_total = 0; continue; }
        batch = fit;
      }
      BCE_TRY(k3_rounds_small(c, batch, cur_nodes, !decaying));
      BCE_HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
      BCE_TRY(k3_sync_ctl(c, &ctl));
      if (ctl.stalled) return k3_stalled(c);        // (before skip_round is read: a stalled round does not set it)
      executed = (ctl.need_flush || ctl.small_bail || ctl.overflow) ? ctl.skip_round - first : batch;
      if (executed > batch) return k3_bad_skip(c, ctl, first, batch);
      BCE_TRY(k3_fetch_runs(c, first, executed));
      runs_fetched = true;
      if (ctl.small_bail) { BCE_TRY(k3_clear_small_bail(c)); wide_once = true; }
    } else {
      // wide rounds: sync often (the round dominates); medium rounds: queue many per sync
      wide_once = false;
      uint32_t batch = cur_nodes > (1u << 22) ? 4u : (cur_nodes > (1u << 20) ? K3_BATCH_MID : (cur_nodes > (1u << 14) ? 16u : 64u));
      // ramp-up: the node count doubles per round; once a round or two reach the next early-flush size, stop there so
      // that the coders get their first (small) batches as early as possible
      if (!decaying && have_ctl && c->stats.flushes < early_max) {
        const uint64_t want = (uint64_t)1 << (20 + c->stats.flushes);
        uint64_t est = ctl.sym_total, nn = cur_nodes;
        uint32_t b = 0;
        while (b < batch && est < want) { est += nn; nn *= 2; ++b; }
        batch = b ? b : 1u;
      }
      {
        const uint32_t fit = rounds_that_fit(batch);
        if (!fit) { BCE_TRY(sink(ctl.sym_total)); ctl.sym_total = 0; continue; }
        batch = fit;
      }
      BCE_TRY(k3_rounds(c, batch, decaying ? cur_nodes : 0));
      BCE_HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
      BCE_TRY(k3_sync_ctl(c, &ctl));
      if (ctl.stalled) return k3_stalled(c);        // (before skip_round is read: a stalled round does not set it)
      executed = (ctl.need_flush || ctl.overflow) ? ctl.skip_round - first : batch;
      if (executed > batch) return k3_bad_skip(c, ctl, first, batch);
      BCE_TRY(k3_fetch_runs(c, first, executed));
      runs_fetched = true;
    }
    { float ms = 0; if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) c->stats.k3_ms += ms; }
    c->round = first + executed;
    if (executed && runs_fetched) {
      // what the last eight rounds emitted, each (h_runs holds this batch's run table: one entry per round and plane): the
      // bulk of the nodes moves from plane to plane with a period of eight rounds, and the symbols per round swing with
      // it -- the largest of a period is what the next round may bring.  Ramp-up: the last round alone (the next doubles).
      const RunEntry *re = reinterpret_cast<const RunEntry *>(c->h_runs);
      for (uint32_t i = 0; i < executed; ++i) {
        uint64_t t = 0;
        for (int p = 0; p < 8; ++p) t += re[(size_t)i * 8 + p].count;
        recent_syms[recent_at++ & 7u] = t;
      }
      est_syms = 0;
      if (decaying) { for (uint64_t v : recent_syms) est_syms = v > est_syms ? v : est_syms; }
      else est_syms = recent_syms[(recent_at - 1u) & 7u];
    }
    if (ctl.overflow) {
      // the round c->round does not fit the node lists (nothing of it was written): larger lists, the same round again
      BCE_TRY(k3_grow_lists(c, ctl));
      ctl.overflow = 0;
      have_ctl = true;
      cur_nodes = 0;
      for (int p = 0; p < 8; ++p) cur_nodes += (uint64_t)ctl.cnt[c->round & 1u][p][0] + ctl.cnt[c->round & 1u][p][1];
      continue;
    }
    if (c->progress) c->progress(ctl.nodes_total, 8ull * n, c->progress_user);
    {
      static const bool plane_trace = getenv("BCE_HIP_PLANE_TRACE") != nullptr;
      if (plane_trace) {
        fprintf(stderr, "round %u: %llu nodes next, k3 %.2f ms so far; per plane:", c->round, (unsigned long long)ctl.next_nodes, c->stats.k3_ms);
        for (int p = 0; p < 8; ++p) fprintf(stderr, " %u", ctl.cnt[c->round & 1u][p][0] + ctl.cnt[c->round & 1u][p][1]);
        fprintf(stderr, "\n");
      }
    }
    decaying = ctl.next_nodes <= cur_nodes && c->round > 16;   // past the ramp-up: the node count no longer doubles
    if (c->dbg_tail_round && c->round >= c->dbg_tail_round) decaying = true;   // test knob 10: the tail starts while the count still grows
    have_ctl = true;
    cur_nodes = ctl.next_nodes;
    const bool done = ctl.done_round != 0xFFFFFFFFu;
    if (ctl.need_flush) {
      if (ctl.sym_total == 0) {
        // one round alone exceeds the symbol buffer: enlarge it and run the round again -- or, when the round emits more than
        // one model flush can take (2^31 records: a high-entropy input of more than ~10^9 bytes), plane group by plane group
        uint64_t want = ctl.want_syms + (ctl.want_syms >> 3) + 1024;
        if (want >= split_limit) {
          bool executed_round = false;
          BCE_TRY(split_round(c, ctl, sink, &executed_round));
          if (!executed_round) continue;           // (the lists were too small for it: grown, the round comes again)
          c->round += 1;
          est_syms = 0;
          if (c->progress) c->progress(ctl.nodes_total, 8ull * n, c->progress_user);
          decaying = ctl.next_nodes <= cur_nodes && c->round > 16;
          cur_nodes = ctl.next_nodes;
          if (ctl.done_round != 0xFFFFFFFFu) break;
          continue;
        }
        BCE_TRY(k3_grow_symbols(c, want));
        continue;
      }
      BCE_TRY(sink(ctl.sym_total));
      ctl.sym_total = 0;                         // (the host copy: what the next batch starts from)
      continue;
    }
    if (done) {
      BCE_TRY(sink(ctl.sym_total));
      break;
    }
    // The host coders are the critical path from the first batch on: hand them small batches early (1M, 2M, 4M, 8M
    // records) instead of waiting for the symbol buffer to fill, so that they are never idle while the GPU works on
    // the next 16M.
    if (c->stats.flushes < early_max && ctl.sym_total >= ((uint64_t)1 << (20 + c->stats.flushes))) {
      BCE_TRY(sink(ctl.sym_total));
      ctl.sym_total = 0;                         // the host copy is consulted again at the top of the loop
    }
  }
  return BCE_HIP_OK;
}

// `bce -s`: BCE<ScanCoder<31>, unbwt::noop>::encode + save_config (bce.cpp:1384-1402, 726-834).  The enumeration
// runs on the GPU in scan mode (raw symbol tuples instead of model records); the eight ScanCoders consume them on
// the host in stream order, then pick the context bits.
static int scan_body(bce_hip_ctx *c, uint8_t *config288, double *result_bytes);
int bce_hip_scan(bce_hip_ctx *c, uint8_t config288[BCE_HIP_CONFIG_BYTES], double result_bytes[9]) {
  const int r = bce_guarded(c, [&] { return scan_body(c, config288, result_bytes); });
  if (c) gate_release(c);                        // (-s keeps the gate to its end: its host part is short)
  return r;
}
static int scan_body(bce_hip_ctx *c, uint8_t *config288, double *result_bytes) {
  BCE_TRY(check_stage(c, 3));
  if (!config288) return BCE_HIP_E_ARG;
  const double t_begin = now_s();
  gate_acquire(c);
  BCE_HIP_TRY(c, hipSetDevice(c->device));
  c->coder->drain();
  c->scan_mode = true;
  c->phase = 3;
  struct Reset { bce_hip_ctx *c; ~Reset() { c->scan_mode = false; c->phase = 0; } } reset{c};
  BCE_TRY(k4_prepare(c));
  BCE_TRY(k3_begin(c));
  c->stats.t_coder = 0;
  ScanSet coders;                                          // planes 0-7 + the header coder, on a pool of host threads
  // the records come over through pinned memory (a copy into a fresh std::vector ran at a fifth of the bus)
  struct PinnedWords {
    uint32_t *p = nullptr; size_t cap = 0;
    ~PinnedWords() { if (p) (void)hipHostFree(p); }
    int ensure(size_t words) {
      if (words <= cap) return 0;
      if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
      if (hipHostMalloc(reinterpret_cast<void **>(&p), words * 4, hipHostMallocDefault) != hipSuccess) { p = nullptr; return 1; }
      cap = words;
      return 0;
    }
  } host;
  double t_copy = 0, t_record = 0, t_pin = 0;
  auto consume = [&](uint64_t nsym) -> int {
    if (nsym) {
      const double tp0 = now_s();
      if (host.ensure((size_t)nsym > (size_t)c->sym_cap ? (size_t)nsym : (size_t)c->sym_cap)) return BCE_HIP_E_NOMEM;
      const double tc0 = now_s();
      t_pin += tc0 - tp0;
      BCE_HIP_TRY(c, hipMemcpyAsync(host.p, c->scanrec.p, (size_t)nsym * 4, hipMemcpyDeviceToHost, c->stream));
      BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
      t_copy += now_s() - tc0;
      std::vector<ScanSpan> spans[8];
      for (int p = 0; p < 8; ++p)
        for (const RunEntry &e : c->run_log[p]) spans[p].push_back(ScanSpan{e.start, e.count});
      const double t0 = now_s();
      coders.consume(host.p, spans);
      c->stats.t_coder += now_s() - t0;
      t_record += now_s() - t0;
    }
    return k3_reset_symbols(c);
  };
  EnumCtl ctl;
  c->stats.flushes = 0; c->stats.symbols = 0;
  BCE_TRY(enumerate_body(c, ctl, [&](uint64_t nsym) { c->stats.flushes++; c->stats.symbols += nsym; return consume(nsym); }));
  c->stats.rounds = ctl.done_round;
  c->stats.nodes = ctl.nodes_total;
  c->enum_active = false;
  uint8_t init[9][32];
  memset(init, 0, sizeof init);                            // ScanCoder::init_ is a zero-initialised static (:834)
  double res[9];
  const double tf0 = now_s();
  coders.flush(init, res);                                 // coder_[i].flush() :1135-1138, then main(-1).flush() :1141-1149
  c->stats.t_coder += now_s() - tf0;
  if (getenv("BCE_HIP_SCAN_DEBUG"))
    fprintf(stderr, "scan: %llu symbols in %llu batches: pinned buffer %.3f s, device-to-host copies %.3f s, recording %.3f s, optimisation %.3f s on %u host threads; K3 %.1f ms; %.3f s since entry\n",
            (unsigned long long)c->stats.symbols, (unsigned long long)c->stats.flushes, t_pin, t_copy, t_record, now_s() - tf0, coders.threads(), c->stats.k3_ms, now_s() - t_begin);
  if (result_bytes) memcpy(result_bytes, res, sizeof res);
  memcpy(config288, init, BCE_HIP_CONFIG_BYTES);
  if (getenv("BCE_HIP_SCAN_DEBUG")) {
    const double t0 = now_s();
    coders.release();
    const double t1 = now_s();
    host.ensure(0); if (host.p) { (void)hipHostFree(host.p); host.p = nullptr; host.cap = 0; }
    fprintf(stderr, "scan: giving back the recorded streams %.3f s, the pinned buffer %.3f s\n", t1 - t0, now_s() - t1);
  }
  return BCE_HIP_OK;
}

int bce_hip_set_plane_mask(bce_hip_ctx *c, uint32_t mask) {
  if (!c) return BCE_HIP_E_ARG;
  return bce_guarded(c, [&] { c->coder->drain(); c->coder->plane_mask = mask & 0xFFu; return (int)BCE_HIP_OK; });
}

int bce_hip_plane_stream_size(bce_hip_ctx *c, int plane, size_t *words) {
  BCE_TRY(check_stage(c, 4));
  if (!words || plane < 0 || plane > 7) return BCE_HIP_E_ARG;
  *words = c->coder->plane[plane].data().size();
  return BCE_HIP_OK;
}

int bce_hip_plane_stream_copy(bce_hip_ctx *c, int plane, uint16_t *out, size_t cap_words) {
  BCE_TRY(check_stage(c, 4));
  if (!out || plane < 0 || plane > 7) return BCE_HIP_E_ARG;
  const std::vector<uint16_t> &d = c->coder->plane[plane].data();
  if (cap_words < d.size()) return BCE_HIP_E_OVERFLOW;
  memcpy(out, d.data(), d.size() * 2);
  return BCE_HIP_OK;
}

int bce_hip_plane_stream_set(bce_hip_ctx *c, int plane, const uint16_t *words, size_t count) {
  BCE_TRY(check_stage(c, 4));
  if ((!words && count) || plane < 0 || plane > 7) return BCE_HIP_E_ARG;
  return bce_guarded(c, [&] {
    c->coder->set_stream(plane, words, count);
    c->coder->rebuild_header(c->config, c->n, c->offset);
    return (int)BCE_HIP_OK;
  });
}

int bce_hip_archive_size(bce_hip_ctx *c, size_t *bytes) {
  BCE_TRY(check_stage(c, 4));
  if (!bytes) return BCE_HIP_E_ARG;
  *bytes = c->coder->archive_words() * 2;
  return BCE_HIP_OK;
}

int bce_hip_archive_copy(bce_hip_ctx *c, uint8_t *out, size_t cap) {
  BCE_TRY(check_stage(c, 4));
  if (!out) return BCE_HIP_E_ARG;
  if (cap < c->coder->archive_words() * 2) return BCE_HIP_E_OVERFLOW;
  return bce_guarded(c, [&] { c->coder->assemble(reinterpret_cast<uint16_t *>(out)); return (int)BCE_HIP_OK; });
}

static int compress_loaded(bce_hip_ctx *c, uint8_t *out, size_t cap, size_t *out_len) {
  const double t0 = now_s();
  BCE_TRY(bce_hip_bwt(c, nullptr));
  BCE_TRY(bce_hip_build_planes(c, nullptr));
  BCE_TRY(bce_hip_encode(c));
  c->stats.t_total = now_s() - t0 + c->stats.t_load;
  if (getenv("BCE_CLI_TIMING"))
    fprintf(stderr, "lib: load %.3f bwt %.3f planes %.3f encode %.3f s (K3 %.1f ms, coder busiest %.3f s, waited for coders %.3f s); device allocations %u calls %.1f MB %.3f s, pinned %u calls %.1f MB %.3f s\n",
            c->stats.t_load, c->stats.t_bwt, c->stats.t_planes, c->stats.t_total - c->stats.t_load - c->stats.t_bwt - c->stats.t_planes, c->stats.k3_ms,
            c->stats.t_coder_busy, c->stats.t_coder, c->alloc_calls, c->alloc_bytes / 1e6, c->alloc_s, c->pin_calls, c->pin_bytes / 1e6, c->pin_s);
  if (out_len) *out_len = c->coder->archive_words() * 2;
  if (out) return bce_hip_archive_copy(c, out, cap);
  return BCE_HIP_OK;
}

int bce_hip_compress(bce_hip_ctx *c, const uint8_t *in, uint32_t n, uint8_t *out, size_t cap, size_t *out_len) {
  if (c && in && n && n < 0x80000000u) k4_prepin(c, n);      // the flush slots' pinned staging, beside the copy and K1 (cold contexts)
  BCE_TRY(bce_hip_load_host(c, in, n));
  return compress_loaded(c, out, cap, out_len);
}
int bce_hip_compress_device(bce_hip_ctx *c, const void *d_in, uint32_t n, uint8_t *out, size_t cap, size_t *out_len) {
  if (c && d_in && n && n < 0x80000000u) k4_prepin(c, n);
  BCE_TRY(bce_hip_load_device(c, d_in, n));
  return compress_loaded(c, out, cap, out_len);
}

int bce_hip_get_stats(const bce_hip_ctx *c, bce_hip_stats *out) {
  if (!c || !out) return BCE_HIP_E_ARG;
  *out = c->stats;
  out->reg_maps = c->reg_maps; out->reg_unmaps = c->reg_unmaps; out->dec_restarts = c->dec_restarts;
  return BCE_HIP_OK;
}

}  // extern "C"
