// hostreg_probe.hip -- a fact about pinned host memory that DESIGN.md 4.5's fault analysis rests on, measured instead of
// assumed.  Nothing here can fault: the spinning kernel touches one word of device memory only.
//   1. Does hipHostUnregister wait for work that is running on the device (as hipHostFree does)?  A kernel spins for
//      ~150 ms on a stream; the host meanwhile unregisters a buffer the kernel never touches and times the call.
//   (2. where glibc puts a 16 MB posix_memalign block in a Python process: tools/heap_placement.py, no GPU needed.)
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/_build/hostreg_probe tools/hostreg_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <unistd.h>

#include <chrono>

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void spin_kernel(uint64_t ticks, uint32_t *out) {
  const uint64_t t0 = wall_clock64();                       // 100 MHz, constant
  uint32_t it = 0;
  while (wall_clock64() - t0 < ticks && it < 400000000u) ++it;   // bounded twice: by the clock and by the count
  if (threadIdx.x == 0) *out = it;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static void *own_map(size_t bytes) {
  void *q = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (q == MAP_FAILED) return nullptr;
  for (size_t o = 0; o < bytes; o += 4096) static_cast<volatile uint8_t *>(q)[o] = 0;
  return q;
}

int main() {
  // ---- 1. what waits for the device ----
  CK(hipSetDevice(0));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  uint32_t *d = nullptr;
  CK(hipMalloc(&d, 64));
  const uint64_t ticks = 15000000;                         // 150 ms at 100 MHz
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, (uint64_t)1000, d);   // warm-up
  CK(hipStreamSynchronize(s));
  const size_t bytes = (size_t)8 << 20;
  for (int what = 0; what < 3; ++what) {
    void *reg = own_map(bytes), *pin = nullptr;
    if (!reg) return 1;
    CK(hipHostRegister(reg, bytes, hipHostRegisterDefault));
    CK(hipHostMalloc(&pin, bytes, hipHostMallocDefault));
    const double t0 = now_s();
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, ticks, d);
    const double t1 = now_s();
    const char *name = "";
    if (what == 0) { name = "hipHostUnregister (buffer the kernel never touches)"; CK(hipHostUnregister(reg)); reg = (munmap(reg, bytes), nullptr); }
    if (what == 1) { name = "hipHostFree"; CK(hipHostFree(pin)); pin = nullptr; }
    if (what == 2) { name = "hipStreamSynchronize (reference)"; CK(hipStreamSynchronize(s)); }
    const double t2 = now_s();
    CK(hipStreamSynchronize(s));
    const double t3 = now_s();
    printf("%-55s returned after %6.1f ms; the kernel ended %6.1f ms after its launch -> %s\n", name, (t2 - t1) * 1e3, (t3 - t0) * 1e3,
           (t2 - t1) > 0.5 * (t3 - t0) ? "WAITS for the device" : "does NOT wait for the device");
    if (reg) { CK(hipHostUnregister(reg)); munmap(reg, bytes); }
    if (pin) CK(hipHostFree(pin));
  }
  CK(hipFree(d));
  CK(hipStreamDestroy(s));
  return 0;
}
