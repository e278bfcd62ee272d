// divsufsort_seam.cpp -- libdivsufsort_hip.so: the two libdivsufsort entry points akamiru/bce calls (bce.cpp:901, :1091),
// under their original names and signatures (include/divsufsort_hip.h), served by libbcehip.so on one process-wide
// context.  Plain C++; everything GPU is behind bce_hip_divbwt / bce_hip_inverse_bwt.
#include <stdlib.h>

#include <mutex>

#include "../../include/bce_hip.h"
#include "../../include/divsufsort_hip.h"

namespace {
std::mutex g_mu;
bce_hip_ctx *g_ctx = nullptr;

bce_hip_ctx *context() {                       // g_mu held
  if (!g_ctx) {
    const char *e = getenv("BCE_HIP_DEVICE");
    if (bce_hip_create(&g_ctx, e ? atoi(e) : 0) != BCE_HIP_OK) g_ctx = nullptr;
  }
  return g_ctx;
}
int status_of(int rc) { return rc == BCE_HIP_OK ? 0 : (rc == BCE_HIP_E_ARG ? -1 : -2); }
}  // namespace

extern "C" {

saidx_t divbwt(const sauchar_t *T, sauchar_t *U, saidx_t * /*A: workspace, not needed*/, saidx_t n) {
  if (!T || !U || n < 0) return -1;
  if (n == 0) return 0;
  if (n == 1) { U[0] = T[0]; return 1; }       // (libdivsufsort's own special case)
  std::lock_guard<std::mutex> g(g_mu);
  bce_hip_ctx *c = context();
  if (!c) return -2;
  uint32_t primary = 0;
  const int rc = bce_hip_divbwt(c, T, U, (uint32_t)n, &primary);
  return rc == BCE_HIP_OK ? (saidx_t)primary : (saidx_t)status_of(rc);
}

saint_t inverse_bw_transform(const sauchar_t *T, sauchar_t *U, saidx_t * /*A*/, saidx_t n, saidx_t idx) {
  if (!T || !U || n < 0 || idx < 0 || idx > n || (n > 0 && idx == 0)) return -1;
  if (n == 0) return 0;
  if (n == 1) { U[0] = T[0]; return 0; }
  std::lock_guard<std::mutex> g(g_mu);
  bce_hip_ctx *c = context();
  if (!c) return -2;
  return (saint_t)status_of(bce_hip_inverse_bwt(c, T, U, (uint32_t)n, (uint32_t)idx));
}

}  // extern "C"
