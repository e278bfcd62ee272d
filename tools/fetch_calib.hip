// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on K3's OWN access patterns (MI355X_MICROARCH.md, HBM
// section: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) ... other access widths
// are uncalibrated: calibrate on a known byte count in your own access pattern before trusting an absolute").
//   a  stream16   16 B per lane, coalesced (the guide's reference pattern: expect FETCH_SIZE = bytes / 2)
//   b  nodes12    K3's node stream: 12 B per lane (three dwords), consecutive lanes 12 B apart, 4 per thread 256 lanes apart
//   c  gather16   K3's rank gathers: one 16-B granule per lane at ascending positions with random gaps (mean 4 granules)
//   d  store12    K3's child writes: 12 B per lane compacted stores
// Each kernel touches BYTES of a buffer far larger than the 256 MB Infinity Cache; tools/fetch_calib.sh runs it under
// rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) and prints counter / known bytes per kernel.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
struct Node { uint32_t s, x0, x1; };
__global__ __launch_bounds__(256) void calib_stream16(const uint4 *src, size_t n16, uint32_t *sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const uint4 v = src[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ __launch_bounds__(256) void calib_nodes12(const Node *src, size_t nn, uint32_t *sink) {
  uint32_t acc = 0;
  const size_t tiles = nn / 1024;
  for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    Node nd[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) nd[it] = src[t * 1024 + (size_t)it * 256 + threadIdx.x];
#pragma unroll
    for (int it = 0; it < 4; ++it) acc += nd[it].s ^ nd[it].x0 ^ nd[it].x1;
  }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ __launch_bounds__(256) void calib_gather16(const uint4 *gran, const uint32_t *idx, size_t nq, uint32_t *sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nq; i += (size_t)gridDim.x * 256) { const uint4 v = gran[idx[i]]; acc += v.x ^ v.w; }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ __launch_bounds__(256) void calib_store12(Node *dst, size_t nn) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nn; i += (size_t)gridDim.x * 256) { Node v = {(uint32_t)i, 1u, 2u}; dst[i] = v; }
}
__global__ void calib_fill_idx(uint32_t *idx, size_t nq, uint32_t ngran) {
  // ascending positions with pseudo-random gaps of 1..7 granules (mean 4), wrapped
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nq; i += (size_t)gridDim.x * 256) {
    uint64_t x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29;
    idx[i] = (uint32_t)((i * 4 + (x % 7)) % ngran);
  }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  const size_t bytes = (size_t)3 << 30;                    // 3 GiB per pattern
  void *buf = nullptr, *buf2 = nullptr; uint32_t *idx = nullptr, *sink = nullptr;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&buf2, bytes)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(buf, 1, bytes));
  const size_t nq = bytes / 16 / 4;                        // one query per 4 granules: 3 GiB of granules spanned, 768 MiB of them touched
  CK(hipMalloc(&idx, nq * 4));
  hipLaunchKernelGGL(calib_fill_idx, dim3(4096), dim3(256), 0, 0, idx, nq, (uint32_t)(bytes / 16));
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(calib_stream16, dim3(2048), dim3(256), 0, 0, (const uint4 *)buf, bytes / 16, sink);
  hipLaunchKernelGGL(calib_nodes12, dim3(2048), dim3(256), 0, 0, (const Node *)buf, bytes / 12 / 1024 * 1024, sink);
  hipLaunchKernelGGL(calib_gather16, dim3(2048), dim3(256), 0, 0, (const uint4 *)buf, idx, nq, sink);
  hipLaunchKernelGGL(calib_store12, dim3(2048), dim3(256), 0, 0, (Node *)buf2, bytes / 12);
  CK(hipDeviceSynchronize());
  printf("known bytes: stream16 %zu nodes12 %zu gather16_lines %zu (16 B x %zu queries = %zu; + index reads %zu) store12 %zu\n",
         bytes, bytes / 12 / 1024 * 1024 * 12, nq * 64, nq, nq * 16, nq * 4, bytes / 12 * 12);
  return 0;
}
