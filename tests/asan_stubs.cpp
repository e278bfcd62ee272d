// asan_stubs.cpp -- TEST-ONLY: the GPU entry points of include/bce_hip.h as "no device" stubs, so that the CLI (main.cpp) and the
// host decoder (decoder.cpp, host_coder.cpp) link into a CPU-only binary that runs under AddressSanitizer / UBSan
// (tests/test_decoder_cpu.py: `bce -ds` on hostile containers).  GPU sanitizers are not available on the pool; the host side
// of the -d path -- argument parsing, the container table, the decoder -- is what parses untrusted bytes.
#include "../include/bce_hip.h"

extern "C" {
int bce_hip_create(bce_hip_ctx **out, int) { if (out) *out = nullptr; return BCE_HIP_E_DEVICE; }
int bce_hip_create_sized(bce_hip_ctx **out, int, uint64_t) { if (out) *out = nullptr; return BCE_HIP_E_DEVICE; }
void bce_hip_destroy(bce_hip_ctx *) {}
const char *bce_hip_strerror(int) { return "no device (sanitizer build)"; }
const char *bce_hip_last_error(const bce_hip_ctx *) { return ""; }
int bce_hip_set_config(bce_hip_ctx *, const uint8_t *) { return BCE_HIP_E_DEVICE; }
int bce_hip_set_gated(bce_hip_ctx *, int) { return BCE_HIP_E_DEVICE; }
int bce_hip_set_progress(bce_hip_ctx *, bce_hip_progress_fn, void *) { return BCE_HIP_E_DEVICE; }
int bce_hip_load_host(bce_hip_ctx *, const uint8_t *, uint32_t) { return BCE_HIP_E_DEVICE; }
int bce_hip_bwt(bce_hip_ctx *, uint32_t *) { return BCE_HIP_E_DEVICE; }
int bce_hip_build_planes(bce_hip_ctx *, uint32_t *) { return BCE_HIP_E_DEVICE; }
int bce_hip_scan(bce_hip_ctx *, uint8_t *, double *) { return BCE_HIP_E_DEVICE; }
int bce_hip_compress(bce_hip_ctx *, const uint8_t *, uint32_t, uint8_t *, size_t, size_t *) { return BCE_HIP_E_DEVICE; }
int bce_hip_archive_copy(bce_hip_ctx *, uint8_t *, size_t) { return BCE_HIP_E_DEVICE; }
int bce_hip_decompress_device(bce_hip_ctx *, const uint8_t *, size_t, uint8_t *, size_t, size_t *) { return BCE_HIP_E_DEVICE; }
}
