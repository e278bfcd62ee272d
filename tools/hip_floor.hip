// hip_floor.hip -- the floor under any one-shot HIP program on this box: runtime initialisation, one stream, one launch,
// and what device / pinned allocations cost per call and per GB (what a cold `bce -c` pays before its first kernel).
// Build: hipcc --offload-arch=gfx950 -O2 tools/hip_floor.hip -o tools/_build/hip_floor -lpthread   (tools/cli_cold.sh runs it)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
__global__ void touch(int *p) { p[threadIdx.x] = threadIdx.x; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char **argv) {
  const char *mode = argc > 1 ? argv[1] : "base";
  const double t0 = now();
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) { printf("no device\n"); return 1; }
  const double t1 = now();
  CK(hipSetDevice(0));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const double t2 = now();
  int *d = nullptr; CK(hipMalloc(&d, 4096));
  const double t3 = now();
  hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, s, d);
  CK(hipStreamSynchronize(s));
  const double t4 = now();
  printf("hip floor [%s]: device count %.3f, stream %.3f, first malloc %.3f, first launch+sync %.3f s\n", mode, t1 - t0, t2 - t1, t3 - t2, t4 - t3);
  if (!strcmp(mode, "big")) {
    void *big = nullptr; double a = now(); CK(hipMalloc(&big, (size_t)10 << 30)); double b = now();
    printf("  hipMalloc 10 GB: %.3f s\n", b - a);
    a = now(); CK(hipMemsetAsync(big, 0, (size_t)10 << 30, s)); CK(hipStreamSynchronize(s)); b = now();
    printf("  first memset of it: %.3f s\n", b - a);
    a = now(); CK(hipFree(big)); b = now();
    printf("  hipFree: %.3f s\n", b - a);
  } else if (!strcmp(mode, "many")) {
    std::vector<void *> ps(10); double a = now();
    for (auto &p : ps) CK(hipMalloc(&p, (size_t)1 << 30));
    double b = now(); printf("  hipMalloc 10 x 1 GB: %.3f s\n", b - a);
    std::vector<void *> qs(40); a = now();
    for (auto &p : qs) CK(hipMalloc(&p, (size_t)16 << 20));
    b = now(); printf("  hipMalloc 40 x 16 MB: %.3f s\n", b - a);
    std::vector<void *> rs(40); a = now();
    for (auto &p : rs) CK(hipMalloc(&p, (size_t)64 << 10));
    b = now(); printf("  hipMalloc 40 x 64 KB: %.3f s\n", b - a);
  } else if (!strcmp(mode, "pin")) {
    void *pin = nullptr; double a = now(); CK(hipHostMalloc(&pin, (size_t)128 << 20, hipHostMallocDefault)); double b = now();
    printf("  hipHostMalloc 128 MB: %.3f s\n", b - a);
    void *pin2 = nullptr; a = now(); CK(hipHostMalloc(&pin2, (size_t)16 << 20, hipHostMallocDefault)); b = now();
    printf("  hipHostMalloc 16 MB: %.3f s\n", b - a);
    void *m = aligned_alloc(1 << 21, (size_t)128 << 20); memset(m, 1, (size_t)128 << 20);
    a = now(); CK(hipHostRegister(m, (size_t)128 << 20, hipHostRegisterDefault)); b = now();
    printf("  hipHostRegister 128 MB (touched): %.3f s\n", b - a);
    // pinning on a second thread while this one launches kernels: does the runtime serialise them?
    void *pin3 = nullptr;
    a = now();
    std::thread th([&] { (void)hipSetDevice(0); (void)hipHostMalloc(&pin3, (size_t)256 << 20, hipHostMallocDefault); });
    double worst = 0; int launches = 0;
    while (now() - a < 0.03) { const double l0 = now(); hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, s, d); (void)hipStreamSynchronize(s); const double l = now() - l0; if (l > worst) worst = l; ++launches; }
    th.join(); b = now();
    printf("  256 MB pinned on a second thread: %.3f s; %d launch+sync beside it, worst %.6f s\n", b - a, launches, worst);
  } else if (!strcmp(mode, "vmm")) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    const size_t total = (size_t)10 << 30;
    void *va = nullptr; double a = now(); CK(hipMemAddressReserve(&va, total, gran, nullptr, 0)); double b = now();
    printf("  granularity %zu; reserve 10 GB of addresses: %.3f s\n", gran, b - a);
    hipMemGenericAllocationHandle_t h; a = now(); CK(hipMemCreate(&h, (size_t)1 << 30, &prop, 0)); b = now();
    printf("  hipMemCreate 1 GB: %.3f s\n", b - a);
    a = now(); CK(hipMemMap(va, (size_t)1 << 30, 0, h, 0));
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, (size_t)1 << 30, &acc, 1)); b = now();
    printf("  map + set access 1 GB: %.3f s\n", b - a);
    CK(hipMemsetAsync(va, 0, (size_t)1 << 30, s)); CK(hipStreamSynchronize(s));
    printf("  memset through the mapping ok\n");
  }
  const double t9 = now();
  printf("  total inside main %.3f s\n", t9 - t0);
  return 0;
}
