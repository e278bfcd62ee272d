// k4_model.hip -- K4: AdaptiveCoder probability estimation on the GPU.  Replaces the model half of
// AdaptiveCoder<31>::set(s,k,c1,c2,cs) (bce.cpp:506-518,529,531-533) and get_context (bce.cpp:671-677);
// the range-coder half stays on the host (host_coder.cpp).
//
// The model is sequential PER CONTEXT SLOT (k byte counters, increment + halve-all at 0xFF) and
// independent across slots.  A flush takes the symbol records K3 emitted since the last flush, in
// stream order (round, plane, s):
//   1. key = plane<<16 | slot (19 bits), value = record index; stable LSD radix sort (3 passes)
//      -> each slot's records become one contiguous run, still in stream order;
//   2. one thread per run replays the slot's counter updates from the persistent counter array and
//      writes (cum, freq, total, nesc) back to the record's ORIGINAL index;
//   3. the host range coders read the outputs in stream order.
// Counter state persists in HBM between flushes, so flushes can be arbitrarily small.
#include "common.h"

namespace bce {

constexpr int K4_T = 256;

struct K4Args {
  const uint64_t *recs;
  const uint32_t *keys, *vals;
  uint8_t *stat;
  const PlaneCfg *cfg;
  uint64_t *out;
  uint32_t *esc;
  uint32_t stat_off[8];
  uint32_t nsym;
};

__global__ __launch_bounds__(K4_T) void k4_keys_kernel(const uint64_t *__restrict__ recs, uint32_t nsym,
                                                       uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
  for (uint64_t i = (uint64_t)blockIdx.x * K4_T + threadIdx.x; i < nsym; i += (uint64_t)gridDim.x * K4_T) {
    keys[i] = (uint32_t)(recs[i] >> kSymKeyShift);
    vals[i] = (uint32_t)i;
  }
}

__global__ __launch_bounds__(K4_T) void k4_simulate_kernel(K4Args a) {
  for (uint64_t j = (uint64_t)blockIdx.x * K4_T + threadIdx.x; j < a.nsym; j += (uint64_t)gridDim.x * K4_T) {
    const uint32_t key = a.keys[j];
    if (j != 0 && a.keys[j - 1] == key) continue;   // not the head of a slot run
    const uint32_t p = key >> 16, slot = key & 0xFFFFu;
    uint32_t idx = a.vals[j];
    uint64_t r = a.recs[idx];
    const uint32_t k = sym_k(r);                    // constant within a slot
    const PlaneCfg &cfg = a.cfg[p];
    uint8_t *ctr = a.stat + a.stat_off[p] + cfg.off[k] + (slot - cfg.ctxoff[k]) * k;
    uint64_t jj = j;
    for (;;) {
      a.out[idx] = model_step(ctr, k, sym_sym(r), sym_nesc(r));
      a.esc[idx] = sym_esc(r);
      ++jj;
      if (jj >= a.nsym || a.keys[jj] != key) break;
      idx = a.vals[jj];
      r = a.recs[idx];
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

int k4_flush(bce_hip_ctx *c, uint64_t nsym64) {
  if (nsym64 == 0) return BCE_HIP_OK;
  if (nsym64 >= (1ull << 31)) return BCE_HIP_E_OVERFLOW;
  const uint32_t nsym = (uint32_t)nsym64;
  const size_t b4 = (size_t)nsym * 4;
  for (int i = 0; i < 2; ++i) { BCE_TRY(ensure(c, c->skey[i], b4)); BCE_TRY(ensure(c, c->sval[i], b4)); }
  BCE_TRY(ensure(c, c->sout, (size_t)nsym * 8));
  BCE_TRY(ensure(c, c->sesc, b4));
  if (c->h_out_cap < nsym) {
    if (c->h_out) (void)hipHostFree(c->h_out);
    if (c->h_esc) (void)hipHostFree(c->h_esc);
    c->h_out = nullptr; c->h_esc = nullptr; c->h_out_cap = 0;
    size_t cap = nsym + (nsym >> 2) + 1024;
    if (cap > c->sym_cap) cap = (size_t)c->sym_cap;
    if (cap < nsym) cap = nsym;
    BCE_HIP_TRY(c, hipHostMalloc((void **)&c->h_out, cap * 8, hipHostMallocDefault));
    BCE_HIP_TRY(c, hipHostMalloc((void **)&c->h_esc, cap * 4, hipHostMallocDefault));
    c->h_out_cap = cap;
  }
  uint32_t *key[2] = {c->skey[0].as<uint32_t>(), c->skey[1].as<uint32_t>()};
  uint32_t *val[2] = {c->sval[0].as<uint32_t>(), c->sval[1].as<uint32_t>()};
  uint64_t gb = ((uint64_t)nsym + K4_T - 1) / K4_T;
  const uint32_t grid = (uint32_t)(gb < 8192 ? gb : 8192);
  hipLaunchKernelGGL(k4_keys_kernel, dim3(grid), dim3(K4_T), 0, c->stream, c->syms.as<uint64_t>(), nsym, key[0], val[0]);
  int res = 0;
  BCE_TRY(radix_sort_pairs(c, key, val, nsym, kSymKeyBits, &res));
  K4Args a;
  a.recs = c->syms.as<uint64_t>();
  a.keys = key[res]; a.vals = val[res];
  a.stat = c->stat.as<uint8_t>();
  a.cfg = c->dcfg.as<PlaneCfg>();
  a.out = c->sout.as<uint64_t>();
  a.esc = c->sesc.as<uint32_t>();
  for (int p = 0; p < 8; ++p) a.stat_off[p] = c->stat_off[p];
  a.nsym = nsym;
  hipLaunchKernelGGL(k4_simulate_kernel, dim3(grid), dim3(K4_T), 0, c->stream, a);
  BCE_HIP_TRY(c, hipMemcpyAsync(c->h_out, c->sout.p, (size_t)nsym * 8, hipMemcpyDeviceToHost, c->stream));
  BCE_HIP_TRY(c, hipMemcpyAsync(c->h_esc, c->sesc.p, b4, hipMemcpyDeviceToHost, c->stream));
  BCE_HIP_TRY(c, hipStreamSynchronize(c->stream));
  BCE_HIP_TRY(c, hipGetLastError());
  return BCE_HIP_OK;
}

}  // namespace bce
