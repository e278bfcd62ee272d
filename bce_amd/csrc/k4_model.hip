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
//   2. replay: ONE WAVE PER RUN, 64 events per iteration.  Between two halvings a counter is
//      base + (number of earlier events with that symbol), so for a 64-event chunk
//        freq  = C[s] + #{earlier lanes with the same symbol}            (+1)
//        cum   = sum_{i<s} C[i] + #{earlier lanes with a smaller symbol} (+s)
//        total = sum_i C[i] + lane                                       (+k)
//      with C held one counter per lane (lane i = counter i), the two counts taken from 5 ballots
//      (bit-sliced less-than / equal masks) and the prefix of C from a 5-step wave scan.  The first
//      lane whose counter reaches 0xFF ends the chunk: events up to it are committed, all counters are
//      halved, and the next chunk starts right after it.
//   3. (cum, freq, total) go back to the record's ORIGINAL index; the host range coders read them in
//      stream order.
// Counter state persists in HBM between flushes, so flushes can be arbitrarily small.
#include "common.h"

namespace bce {

constexpr int K4_T = 256;

struct K4Args {
  const uint32_t *keys, *vals;   // sorted
  uint8_t *stat;
  const PlaneCfg *cfg;
  uint64_t *out;
  uint32_t stat_off[8];
  uint32_t nsym;
};

__global__ __launch_bounds__(K4_T) void k4_iota_kernel(uint32_t nsym, uint32_t *__restrict__ vals) {
  for (uint64_t i = (uint64_t)blockIdx.x * K4_T + threadIdx.x; i < nsym; i += (uint64_t)gridDim.x * K4_T)
    vals[i] = (uint32_t)i;
}

// replay one slot run [start, ...) with the whole wave; hkey = key word of its first record
__device__ __forceinline__ void k4_run(const K4Args &a, uint64_t start, uint32_t hkey, volatile uint32_t *cb,
                                       uint32_t lane, uint64_t ltm) {
  const uint32_t runid = hkey >> kSymRunShift;
  const uint32_t k = key_k(hkey), p = runid >> 16, slot = runid & 0xFFFFu;
  const PlaneCfg &cfg = a.cfg[p];
  uint8_t *ctr = a.stat + a.stat_off[p] + cfg.off[k] + (slot - cfg.ctxoff[k]) * k;
  uint32_t C = lane < k ? (uint32_t)ctr[lane] : 0u;
  uint64_t pos = start;
  uint64_t j = pos + lane;
  uint32_t kk = j < a.nsym ? a.keys[j] : 0xFFFFFFFFu;
  for (;;) {
    const bool valid = (kk >> kSymRunShift) == runid;     // the run is contiguous: valid lanes form a prefix
    const uint64_t vm = __ballot(valid);
    if (!vm) break;
    const uint32_t idx = valid ? a.vals[j] : 0u;
    // speculative load of the next chunk (correct unless a halving cuts this one short)
    const uint64_t jn = j + 64;
    const uint32_t kn = jn < a.nsym ? a.keys[jn] : 0xFFFFFFFFu;
    const uint32_t s = kk & 31u;
    // exclusive prefix of the counters over lanes 0..31 and their total
    uint32_t inc = C;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
      const uint32_t t = __shfl_up(inc, o);
      if (lane >= (uint32_t)o) inc += t;
    }
    const uint32_t T = __shfl(inc, 31);
    const uint32_t Ps = __shfl(inc - C, (int)s);
    const uint32_t Cs = __shfl(C, (int)s);
    // lanes with an equal / a smaller symbol (valid lanes only), bit-sliced from the MSB down
    uint64_t eq = vm, lt = 0;
#pragma unroll
    for (int b = 4; b >= 0; --b) {
      const bool bit = (s >> b) & 1u;
      const uint64_t Bb = __ballot(bit);
      if (bit) { lt |= eq & ~Bb; eq &= Bb; } else { eq &= ~Bb; }
    }
    const uint32_t eqb = (uint32_t)__popcll(eq & ltm), ltb = (uint32_t)__popcll(lt & ltm);
    const uint32_t before = Cs + eqb;                       // counter value this event sees
    const uint64_t hm = __ballot(valid && before + 1u == 0xFFu);
    const uint32_t nc = hm ? (uint32_t)__ffsll((long long)hm) : (uint32_t)__popcll(vm);   // committed events
    const bool commit = lane < nc;
    if (commit) {
      const uint32_t cum = Ps + ltb + s, total = T + lane + k, freq = before + 1u;
      a.out[idx] = (uint64_t)cum | ((uint64_t)freq << 16) | ((uint64_t)total << 32);
    }
    // per-symbol counts of the committed events -> lane i gets the count of symbol i
    const uint64_t cm = nc >= 64 ? ~0ull : ((1ull << nc) - 1ull);
    const uint64_t eqc = eq & cm;
    if (lane < 32) cb[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    if (commit && (eqc >> lane) == 1ull) cb[s] = (uint32_t)__popcll(eqc);   // highest committed lane of its group
    __builtin_amdgcn_wave_barrier();
    C += lane < 32 ? cb[lane] : 0u;
    __builtin_amdgcn_wave_barrier();
    if (hm) C >>= 1;                                       // bce.cpp:531-533
    pos += nc;
    if (nc == 64) { j = jn; kk = kn; }
    else { j = pos + lane; kk = j < a.nsym ? a.keys[j] : 0xFFFFFFFFu; }
  }
  if (lane < k) ctr[lane] = (uint8_t)C;
}

__global__ __launch_bounds__(K4_T) void k4_simulate_kernel(K4Args a) {
  __shared__ uint32_t cntbuf[K4_T / 64][32];
  const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  volatile uint32_t *cb = cntbuf[w];
  const uint64_t ltm = (1ull << lane) - 1ull;
  const uint64_t nwaves = (uint64_t)gridDim.x * (K4_T / 64);
  for (uint64_t win = (uint64_t)blockIdx.x * (K4_T / 64) + w; win * 64 < a.nsym; win += nwaves) {
    const uint64_t j0 = win * 64 + lane;
    const uint32_t key = j0 < a.nsym ? a.keys[j0] : 0xFFFFFFFFu;
    const uint32_t prev = (j0 > 0 && j0 < a.nsym) ? a.keys[j0 - 1] : 0xFFFFFFFFu;
    const bool head = j0 < a.nsym && (j0 == 0 || (key >> kSymRunShift) != (prev >> kSymRunShift));
    uint64_t heads = __ballot(head);
    while (heads) {                                         // runs that START in this 64-record window
      const int hl = __ffsll((long long)heads) - 1;
      heads &= heads - 1;
      const uint32_t hkey = __shfl(key, hl);
      k4_run(a, win * 64 + (uint64_t)hl, hkey, cb, lane, ltm);
    }
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

int k4_flush(bce_hip_ctx *c, uint64_t nsym64, FlushSlot &slot) {
  if (nsym64 == 0) return BCE_HIP_OK;
  if (nsym64 >= (1ull << 31)) return BCE_HIP_E_OVERFLOW;
  const uint32_t nsym = (uint32_t)nsym64;
  const size_t b4 = (size_t)nsym * 4;
  BCE_TRY(ensure(c, c->skey[1], b4));
  for (int i = 0; i < 2; ++i) BCE_TRY(ensure(c, c->sval[i], b4));
  BCE_TRY(ensure(c, c->sout, (size_t)nsym * 8));
  if (slot.cap < nsym) {
    if (slot.h_out) (void)hipHostFree(slot.h_out);
    if (slot.h_esc) (void)hipHostFree(slot.h_esc);
    slot.h_out = nullptr; slot.h_esc = nullptr; slot.cap = 0;
    size_t cap = (size_t)c->sym_cap > nsym ? (size_t)c->sym_cap : nsym;
    BCE_HIP_TRY(c, hipHostMalloc((void **)&slot.h_out, cap * 8, hipHostMallocDefault));
    BCE_HIP_TRY(c, hipHostMalloc((void **)&slot.h_esc, cap * 4, hipHostMallocDefault));
    slot.cap = cap;
  }
  // the escape words are final as K3 wrote them: start their copy first
  BCE_HIP_TRY(c, hipMemcpyAsync(slot.h_esc, c->sesc.p, b4, hipMemcpyDeviceToHost, c->stream));
  uint32_t *key[2] = {c->skey[0].as<uint32_t>(), c->skey[1].as<uint32_t>()};
  uint32_t *val[2] = {c->sval[0].as<uint32_t>(), c->sval[1].as<uint32_t>()};
  uint64_t gb = ((uint64_t)nsym + K4_T - 1) / K4_T;
  const uint32_t grid = (uint32_t)(gb < 8192 ? gb : 8192);
  hipLaunchKernelGGL(k4_iota_kernel, dim3(grid), dim3(K4_T), 0, c->stream, nsym, val[0]);
  int res = 0;
  BCE_TRY(radix_sort_pairs(c, key, val, nsym, kSymRunShift, kSymRunBits, &res));
  K4Args a;
  a.keys = key[res]; a.vals = val[res];
  a.stat = c->stat.as<uint8_t>();
  a.cfg = c->dcfg.as<PlaneCfg>();
  a.out = c->sout.as<uint64_t>();
  for (int p = 0; p < 8; ++p) a.stat_off[p] = c->stat_off[p];
  a.nsym = nsym;
  // one wave per 64-record window; windows without a run head exit at once
  uint64_t wb = ((uint64_t)nsym + 64 * (K4_T / 64) - 1) / (64 * (K4_T / 64));
  const uint32_t sgrid = (uint32_t)(wb < 16384 ? (wb ? wb : 1) : 16384);
  hipLaunchKernelGGL(k4_simulate_kernel, dim3(sgrid), dim3(K4_T), 0, c->stream, a);
  BCE_HIP_TRY(c, hipMemcpyAsync(slot.h_out, c->sout.p, (size_t)nsym * 8, hipMemcpyDeviceToHost, c->stream));
  BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
  BCE_HIP_TRY(c, hipGetLastError());
  return BCE_HIP_OK;
}

}  // namespace bce
