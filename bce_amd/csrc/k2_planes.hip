// k2_planes.hip -- K2: wavelet-matrix bit planes + rank directory.  Replaces the RankFile constructor
// body and Rank::build (bce.cpp:944-970, 138-145).
//
// The reference scatters 8n single bits through a 256-entry cursor heap.  Here plane j is produced by
// level j of a wavelet matrix: plane j = bit j of the bytes in the current order, next order = stable
// partition of the bytes by bit j (zeros first) -- the same permutation as sorting by the low j bits
// (heap index (chr & ((1<<j)-1)) | (1<<j), bce.cpp:965).  Each level is two coalesced passes over n
// bytes: (1) zeros per block, (2) partition + emit 16-byte rank granules {cum, 96 payload bits}.
// A chunk is 3072 positions = 256 threads x 12 bytes = 32 granules, so granules never straddle chunks
// and a granule's payload is assembled from 8 neighbouring lanes with shuffles.
#include "common.h"
#include "scan_util.h"

namespace bce {

constexpr int K2_T = 256;
constexpr uint32_t K2_CHUNK = 3072;  // = 32 granules

struct K2Plan { uint32_t chunks, nb, cpb; };
static K2Plan k2_plan(uint32_t n) {
  // positions 0..n inclusive must have a granule (rank(n) is queried): cover n+1 positions
  uint32_t chunks = (uint32_t)(((uint64_t)n + 1 + K2_CHUNK - 1) / K2_CHUNK);
  uint32_t nb = chunks < 1024u ? chunks : 1024u;
  uint32_t cpb = (chunks + nb - 1) / nb;
  nb = (chunks + cpb - 1) / cpb;
  return {chunks, nb, cpb};
}

// load the 12 bytes of a thread (3 dwords); bytes at positions >= n read as 0
__device__ __forceinline__ void k2_load12(const uint8_t *__restrict__ B, uint64_t pos0, uint32_t n, uint32_t w[3]) {
  if (pos0 + 12 <= n) {
    const uint32_t *p = reinterpret_cast<const uint32_t *>(B + pos0);
    w[0] = p[0]; w[1] = p[1]; w[2] = p[2];
  } else {
    w[0] = w[1] = w[2] = 0;
    for (uint32_t i = 0; i < 12; ++i)
      if (pos0 + i < n) w[i >> 2] |= (uint32_t)B[pos0 + i] << (8 * (i & 3));
  }
}
// bit j of each of the 12 bytes -> 12-bit mask (bit i = byte i)
__device__ __forceinline__ uint32_t k2_mask12(const uint32_t w[3], int j) {
  uint32_t m = 0;
#pragma unroll
  for (int wi = 0; wi < 3; ++wi) {
    const uint32_t t = (w[wi] >> j) & 0x01010101u;
    // gather bits 0,8,16,24 into 4 consecutive bits
    const uint32_t nib = (t | (t >> 7) | (t >> 14) | (t >> 21)) & 0xFu;
    m |= nib << (4 * wi);
  }
  return m;
}

__global__ __launch_bounds__(K2_T) void k2_count_kernel(const uint8_t *__restrict__ B, uint32_t n, uint32_t cpb,
                                                        uint32_t chunks, int j, uint32_t *__restrict__ zc) {
  uint32_t c0 = blockIdx.x * cpb, c1 = c0 + cpb;
  if (c1 > chunks) c1 = chunks;
  uint32_t zeros = 0;
  for (uint32_t ch = c0; ch < c1; ++ch) {
    const uint64_t pos0 = (uint64_t)ch * K2_CHUNK + 12u * threadIdx.x;
    uint32_t w[3];
    k2_load12(B, pos0, n, w);
    const uint32_t m = k2_mask12(w, j);
    const uint32_t v = pos0 >= n ? 0u : (n - pos0 < 12 ? (uint32_t)(n - pos0) : 12u);
    zeros += v - popc32(m);
  }
  zeros = block_reduce_sum<K2_T>(zeros);
  if (threadIdx.x == 0) zc[blockIdx.x] = zeros;
}

__global__ __launch_bounds__(K2_T) void k2_partition_kernel(const uint8_t *__restrict__ B, uint8_t *__restrict__ Bn,
                                                            uint32_t n, uint32_t cpb, uint32_t chunks, uint32_t nb,
                                                            int j, const uint32_t *__restrict__ zc,
                                                            Granule *__restrict__ G, uint32_t *__restrict__ zout) {
  const uint32_t tid = threadIdx.x;
  // zeros in earlier blocks, zeros in total
  uint32_t zb = 0, za = 0;
  for (uint32_t b = tid; b < nb; b += K2_T) { const uint32_t v = zc[b]; za += v; if (b < blockIdx.x) zb += v; }
  uint32_t zrun = block_reduce_sum<K2_T>(zb);   // zeros before the current chunk
  const uint32_t Z = block_reduce_sum<K2_T>(za);
  if (blockIdx.x == 0 && tid == 0) zout[j] = Z;
  uint32_t c0 = blockIdx.x * cpb, c1 = c0 + cpb;
  if (c1 > chunks) c1 = chunks;
  for (uint32_t ch = c0; ch < c1; ++ch) {
    const uint64_t cstart = (uint64_t)ch * K2_CHUNK;
    const uint64_t pos0 = cstart + 12u * tid;
    uint32_t w[3];
    k2_load12(B, pos0, n, w);
    const uint32_t m = k2_mask12(w, j);
    const uint32_t v = pos0 >= n ? 0u : (n - pos0 < 12 ? (uint32_t)(n - pos0) : 12u);
    const uint32_t ones = popc32(m), zeros = v - ones;
    uint32_t tot;
    const uint32_t ex = block_excl_scan_sum<K2_T>(zeros | (ones << 16), &tot);
    const uint32_t zex = ex & 0xFFFFu, oex = ex >> 16;
    // cstart <= n always holds for chunks that exist, so this cannot underflow
    const uint32_t ones_before_chunk = (uint32_t)((cstart < n ? cstart : n) - zrun);
    // rank granule: 8 consecutive lanes = 96 bits
    uint32_t mm[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) mm[k] = __shfl_down(m, k);
    if ((tid & 7u) == 0) {
      Granule g;
      g.cum = ones_before_chunk + oex;
      g.w0 = mm[0] | (mm[1] << 12) | ((mm[2] & 0xFFu) << 24);
      g.w1 = (mm[2] >> 8) | (mm[3] << 4) | (mm[4] << 16) | ((mm[5] & 0xFu) << 28);
      g.w2 = (mm[5] >> 4) | (mm[6] << 8) | (mm[7] << 20);
      G[(size_t)ch * 32 + (tid >> 3)] = g;
    }
    if (Bn) {
      uint32_t zd = zrun + zex;                       // next zero destination
      uint32_t od = Z + ones_before_chunk + oex;      // next one destination
#pragma unroll
      for (uint32_t i = 0; i < 12; ++i) {
        if (i < v) {
          const uint8_t byte = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
          if ((m >> i) & 1u) Bn[od++] = byte; else Bn[zd++] = byte;
        }
      }
    }
    zrun += tot & 0xFFFFu;
  }
}

// test hooks -----------------------------------------------------------------------------------------
__global__ void k2_bits_kernel(const Granule *__restrict__ G, uint32_t n, uint8_t *__restrict__ out) {
  for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t g = div96((uint32_t)p), o = (uint32_t)p - g * 96u;
    const Granule q = G[g];
    const uint32_t wv = o < 32 ? q.w0 : (o < 64 ? q.w1 : q.w2);
    out[p] = (uint8_t)((wv >> (o & 31u)) & 1u);
  }
}
__global__ void k2_rank_kernel(const Granule *__restrict__ G, const uint32_t *__restrict__ idx, uint32_t cnt,
                               uint32_t *__restrict__ out) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += gridDim.x * blockDim.x) {
    const uint32_t s = idx[i], g = div96(s);
    out[i] = granule_rank1(G[g], s - g * 96u);
  }
}

int k2_build_planes(bce_hip_ctx *c) {
  const uint32_t n = c->n;
  const K2Plan pl = k2_plan(n);
  c->ngran = pl.chunks * 32u;
  BCE_TRY(ensure(c, c->gran, (size_t)8 * c->ngran * sizeof(Granule)));
  BCE_TRY(ensure(c, c->ptmp[0], n));
  BCE_TRY(ensure(c, c->ptmp[1], n));
  BCE_TRY(ensure(c, c->blk, (size_t)(pl.nb + 16) * 4));
  uint32_t *zc = c->blk.as<uint32_t>();
  uint32_t *zout = zc + pl.nb;  // 8 entries
  const uint8_t *cur = c->bwt.as<uint8_t>();
  for (int j = 0; j < 8; ++j) {
    uint8_t *nxt = j < 7 ? c->ptmp[j & 1].as<uint8_t>() : nullptr;
    Granule *G = c->gran.as<Granule>() + (size_t)j * c->ngran;
    hipLaunchKernelGGL(k2_count_kernel, dim3(pl.nb), dim3(K2_T), 0, c->stream, cur, n, pl.cpb, pl.chunks, j, zc);
    hipLaunchKernelGGL(k2_partition_kernel, dim3(pl.nb), dim3(K2_T), 0, c->stream, cur, nxt, n, pl.cpb, pl.chunks,
                       pl.nb, j, zc, G, zout);
    cur = nxt;
  }
  BCE_TRY(read_back(c, c->zeros, zout, 32));
  BCE_HIP_TRY(c, hipGetLastError());
  return BCE_HIP_OK;
}

int k2_get_plane_bits(bce_hip_ctx *c, int plane, uint8_t *out) {
  const uint32_t n = c->n;
  DevBuf tmp;
  BCE_TRY(ensure(c, tmp, n));
  const Granule *G = c->gran.as<Granule>() + (size_t)plane * c->ngran;
  hipLaunchKernelGGL(k2_bits_kernel, dim3(1024), dim3(256), 0, c->stream, G, n, tmp.as<uint8_t>());
  hipError_t e = hipMemcpyAsync(out, tmp.p, n, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  release(tmp);
  BCE_HIP_TRY(c, e);
  return BCE_HIP_OK;
}

int k2_rank1(bce_hip_ctx *c, int plane, const uint32_t *idx, uint32_t count, uint32_t *out) {
  if (!count) return BCE_HIP_OK;
  DevBuf di, dout;
  int rc = ensure(c, di, (size_t)count * 4);
  if (rc == BCE_HIP_OK) rc = ensure(c, dout, (size_t)count * 4);
  hipError_t e = hipSuccess;
  if (rc == BCE_HIP_OK) {
    const Granule *G = c->gran.as<Granule>() + (size_t)plane * c->ngran;
    e = hipMemcpyAsync(di.p, idx, (size_t)count * 4, hipMemcpyHostToDevice, c->stream);
    hipLaunchKernelGGL(k2_rank_kernel, dim3(256), dim3(256), 0, c->stream, G, di.as<uint32_t>(), count,
                       dout.as<uint32_t>());
    if (e == hipSuccess) e = hipMemcpyAsync(out, dout.p, (size_t)count * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  }
  release(di);
  release(dout);
  BCE_TRY(rc);
  BCE_HIP_TRY(c, e);
  return BCE_HIP_OK;
}

}  // namespace bce
