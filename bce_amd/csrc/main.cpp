// main.cpp -- the `bce` command line, mirroring the reference's main() (bce.cpp:1376-1484):
//   bce -c archive.bce file [config.bcc]    compress on the MI355X through libbcehip.so
// Banner, usage text, summary line, argument detection and exit codes follow the reference
// (banner :1377-1379, -c :1403-1427, -d :1428-1472, usage :1473-1483).  -d uses the GPU-assisted decoder (kd_decode.hip), -ds the host decoder (decoder.cpp);
// -s runs the enumeration on the GPU in scan mode and the ScanCoder optimisation on the host (scan_coder.cpp).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <exception>
#include <fstream>
#include <string>
#include <algorithm>
#include <atomic>
#include <thread>
#include <memory>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/bce_hip.h"

// "Coded: %u.%02u %%\r" as BCE::code prints it (bce.cpp:1354-1358); cleared at the end like :1372
static void progress(uint64_t done, uint64_t total, void *user) {
  uint64_t *prev = static_cast<uint64_t *>(user);
  const uint64_t cur = total ? done * 10000 / total : 0;
  if (cur > *prev) {
    printf("Coded: %llu.%02llu %%\r", (unsigned long long)(cur / 100), (unsigned long long)(cur % 100));
    fflush(stdout);
    *prev = cur;
  }
}
static void progress_end() { printf("                    \r"); }

// File::File (bce.cpp:842-856): the whole file in memory.  Read on a thread of its own while the main thread initialises the
// HIP runtime (bce_hip_create: ~0.1 s on this platform, the largest fixed cost of a one-shot run) -- plain read(2) into
// memory nobody has zeroed first (a std::vector's resize touches every page once more).
struct HostFile {
  uint8_t *p = nullptr;
  size_t n = 0;
  int status = -1;            // 0 ok, -1 not found / unreadable, -2 short read, -3 at or above the caller's size limit (nothing read)
  ~HostFile() { free(p); }
  const uint8_t *data() const { return p; }
  size_t size() const { return n; }
};
static void read_whole_file(const char *path, HostFile *f, size_t limit) {
  const int fd = open(path, O_RDONLY | O_CLOEXEC);
  if (fd < 0) return;
  struct stat st;
  if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 0) { close(fd); return; }
  const size_t n = (size_t)st.st_size;
  if (limit && n >= limit) { close(fd); f->n = n; f->status = -3; return; }   // too large for the caller: not read at all
  f->p = static_cast<uint8_t *>(malloc(n ? n : 1));
  if (!f->p) { close(fd); return; }
  size_t got = 0;
  while (got < n) {
    const ssize_t r = read(fd, f->p + got, n - got);
    if (r <= 0) break;
    got += (size_t)r;
  }
  close(fd);
  f->n = got;
  f->status = got == n ? 0 : -2;
}

// n < 2^31: saidx_t / the 31-bit getv of the header (bce.cpp:173,374,901)
static const size_t kMaxInput = (size_t)0x80000000ull;

// Writes the whole output and says whether it arrived: a short or failed write (ENOSPC, EIO) must not end in the success
// line and exit code 0 -- after fast_exit nothing later could report it.
static bool write_file(const char *path, const void *p, size_t bytes) {
  std::ofstream f(std::string(path), std::ios::binary | std::ios::trunc);
  if (!f) return false;
  f.write(static_cast<const char *>(p), (std::streamsize)bytes);
  f.close();
  return f.good();
}

// The output is on disk and stdout is flushed: leave without the tear-down of 20 GB of device buffers, 400 MB of pinned
// memory and the runtime's own exit handlers (0.1-0.25 s that a user of a one-shot tool would wait for nothing; the kernel
// reclaims everything).  BCE_CLI_CLEAN_EXIT=1 takes the long way (leak checkers).
static void fast_exit(int code) {
  fflush(stdout);
  fflush(stderr);
  if (!getenv("BCE_CLI_CLEAN_EXIT")) _exit(code);
}

// ---- multi-block container (an extension; the reference has one block per archive, bce.cpp:1151-1157) ----
// b"BCEM" | u32 version = 1 | u32 nblocks | nblocks x (u64 raw_bytes, u64 archive_bytes) | the archives.  Every
// embedded archive is exactly what `bce -c` writes for that block alone.  Same layout as bce_amd/container.py
// (the 8-GPU path of bench.py gathers its blocks into it).  A plain archive cannot start with "BCEM": that would
// be a header of 0x4342 words.
static bool is_container(const HostFile &a) { return a.size() >= 12 && memcmp(a.data(), "BCEM", 4) == 0; }
static void put_u32(std::vector<uint8_t> &v, uint32_t x) { for (int i = 0; i < 4; ++i) v.push_back((uint8_t)(x >> (8 * i))); }
static void put_u64(std::vector<uint8_t> &v, uint64_t x) { for (int i = 0; i < 8; ++i) v.push_back((uint8_t)(x >> (8 * i))); }
static uint64_t get_le(const uint8_t *p, int bytes) { uint64_t x = 0; for (int i = 0; i < bytes; ++i) x |= (uint64_t)p[i] << (8 * i); return x; }

// `bce -cN`: N contiguous blocks over the GPUs of the node.  With more blocks than GPUs every device gets up to four
// gated contexts (bce_hip_set_gated), one host thread each: their GPU phases take turns while the coder threads of the
// context that has just left the GPU finish its block.  Blocks are handed out in order to whichever context is free.
static int compress_blocks(const HostFile &data, uint32_t nblocks, const uint8_t *config, std::vector<uint8_t> &out) {
  std::vector<bce_hip_ctx *> ctx;
  int ndev = 0;
  for (int dev = 0; dev < 64; ++dev) {
    bce_hip_ctx *c = nullptr;
    if (bce_hip_create(&c, dev) != 0) break;
    ctx.push_back(c);
    ndev = dev + 1;
    if (ctx.size() >= nblocks) break;
  }
  if (ctx.empty()) return -3;
  const size_t per_dev = std::min<size_t>(4, (nblocks + (size_t)ndev - 1) / (size_t)ndev);
  for (size_t extra = 1; extra < per_dev; ++extra)
    for (int dev = 0; dev < ndev && ctx.size() < nblocks; ++dev) {
      bce_hip_ctx *c = nullptr;
      if (bce_hip_create(&c, dev) != 0) break;
      ctx.push_back(c);
    }
  for (size_t i = 0; i < ctx.size(); ++i) {
    if (config && bce_hip_set_config(ctx[i], config) != 0) {        // (validated by the caller already: cannot happen)
      printf("Config rejected: %s\n", bce_hip_last_error(ctx[i]));
      for (bce_hip_ctx *o : ctx) bce_hip_destroy(o);
      return -1;
    }
  }
  const size_t n = data.size(), base = n / nblocks, rem = n % nblocks;
  std::vector<std::vector<uint8_t>> arch(nblocks);
  std::vector<size_t> lo(nblocks + 1);
  for (uint32_t b = 0; b <= nblocks; ++b) lo[b] = b * base + (b < rem ? b : rem);
  std::vector<int> rcs(ctx.size(), 0);
  std::atomic<uint32_t> next_block{0};
  std::atomic<bool> failed{false};
  // A context that runs out of device memory (four contexts of a device and blocks of hundreds of MB: a context needs ~180
  // bytes per input byte) gives its memory back and leaves its block for later: what is left over is compressed at the end by
  // one context that has the device to itself.  Only a block that does not fit even then fails.
  std::mutex deferred_mu;
  std::vector<uint32_t> deferred;
  const char *test_nomem = getenv("BCE_CLI_TEST_NOMEM_BLOCK");      // (test hook: this block's first attempt "runs out of memory"; "all": every block's)
  const bool test_all = test_nomem && strcmp(test_nomem, "all") == 0;
  const long test_block = test_nomem && !test_all ? atol(test_nomem) : -1;
  std::vector<char> done(nblocks, 0);                               // block b's archive is in arch[b]
  std::vector<std::thread> th;
  for (size_t d = 0; d < ctx.size(); ++d)
    th.emplace_back([&, d] {
      while (!failed.load()) {
        const uint32_t b = next_block.fetch_add(1);
        if (b >= nblocks) break;
        size_t alen = 0;
        int rc = (test_all || (long)b == test_block) ? BCE_HIP_E_NOMEM
                                         : bce_hip_compress(ctx[d], data.data() + lo[b], (uint32_t)(lo[b + 1] - lo[b]), nullptr, 0, &alen);
        if (rc == 0) { arch[b].resize(alen); rc = bce_hip_archive_copy(ctx[d], arch[b].data(), alen); }
        if (rc == 0) done[b] = 1;
        if (rc == BCE_HIP_E_NOMEM) {
          { std::lock_guard<std::mutex> lk(deferred_mu); deferred.push_back(b); }
          bce_hip_destroy(ctx[d]);
          ctx[d] = nullptr;
          return;
        }
        if (rc) { rcs[d] = rc; failed.store(true); break; }
      }
      (void)bce_hip_set_gated(ctx[d], 1);                           // (gives the gate back if a failed stage left it held)
    });
  for (auto &t : th) t.join();
  int rc = 0;
  for (size_t d = 0; d < ctx.size(); ++d) {
    if (!ctx[d]) continue;
    if (rcs[d] && !rc) { rc = rcs[d]; printf("%s\n", bce_hip_last_error(ctx[d])); }
    bce_hip_destroy(ctx[d]);
  }
  // Every worker may have left that way (another tenant on the device, blocks too large for four contexts side by side):
  // the blocks nobody took are then in no list.  Whatever has no archive yet is compressed here, one block at a time.
  if (!rc) {
    deferred.clear();
    for (uint32_t b = 0; b < nblocks; ++b) if (!done[b]) deferred.push_back(b);
  }
  if (!rc && !deferred.empty()) {
    bce_hip_ctx *c = nullptr;
    rc = bce_hip_create(&c, 0);
    if (rc == 0 && config) rc = bce_hip_set_config(c, config);
    for (size_t i = 0; rc == 0 && i < deferred.size(); ++i) {
      const uint32_t b = deferred[i];
      size_t alen = 0;
      rc = bce_hip_compress(c, data.data() + lo[b], (uint32_t)(lo[b + 1] - lo[b]), nullptr, 0, &alen);
      if (rc == 0) { arch[b].resize(alen); rc = bce_hip_archive_copy(c, arch[b].data(), alen); }
      if (rc == 0) done[b] = 1;
    }
    if (rc && c) printf("%s\n", bce_hip_last_error(c));
    if (c) bce_hip_destroy(c);
  }
  if (rc) return rc;
  for (uint32_t b = 0; b < nblocks; ++b)
    if (!done[b] || arch[b].empty()) return BCE_HIP_E_INTERNAL;     // a container never goes out with a block missing
  out.clear();
  out.insert(out.end(), {'B', 'C', 'E', 'M'});
  put_u32(out, 1);
  put_u32(out, nblocks);
  for (uint32_t b = 0; b < nblocks; ++b) { put_u64(out, lo[b + 1] - lo[b]); put_u64(out, arch[b].size()); }
  for (uint32_t b = 0; b < nblocks; ++b) out.insert(out.end(), arch[b].begin(), arch[b].end());
  return 0;
}

static int usage() {
  printf("Usage:\n");
  printf("  bce -c archive.bce file [config.bcc]\n");
  printf("   Compresses \"file\" to archive \"archive.bce\" [using config \"config.bcc\"]\n");
  printf("\n");
  printf("  bce -d file archive.bce\n");
  printf("   Decompresses archive \"archive.bce\" to \"file\"\n");
  printf("\n");
  printf("  bce -s config.bcc file\n");
  printf("   Scan \"file\" and generate a config file \"config.bcc\" to improve the AdaptiveCoder (uses a lot of memory)\n");
  printf("\n");
  printf("  bce -cN archive.bcem file [config.bcc]      (extension: N = 2..64 blocks, one container, all GPUs of the node; every block < 2^31 bytes)\n");
  return 0;
}

int main(int argc, char **argv) {
  printf("BCE v0.4 Release\n");
  printf("Copyright (C) 2016  Christoph Diegelmann\n");
  printf("This is free software under GNU Lesser General Public License. See <http://www.gnu.org/licenses/lgpl>\n\n");

  if ((argc == 4 || argc == 5) && argv[1][0] == '-' && argv[1][1] == 'c') {
    auto start = std::chrono::high_resolution_clock::now();
    HostFile data;
    // (-cN, the extension: every BLOCK obeys the reference's n < 2^31, the file may be N times that)
    const uint32_t nb_arg = (uint32_t)atoi(argv[1] + 2);
    const size_t file_limit = nb_arg >= 2 && nb_arg <= 64 ? (size_t)nb_arg * (kMaxInput - 1) + 1 : kMaxInput;
    std::thread reader(read_whole_file, argv[3], &data, file_limit);  // File::File, bce.cpp:842-856 -- beside the runtime's start-up
    bce_hip_ctx *ctx = nullptr;
    uint64_t expect = 0;                                          // the file's size, if it says: what the context prepares for
    { struct stat st; if (stat(argv[3], &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) expect = (uint64_t)st.st_size; }
    if (atoi(argv[1] + 2) >= 2) expect = 0;                       // (-cN: the blocks get contexts of their own)
    int rc = bce_hip_create_sized(&ctx, 0, expect);
    if (rc != 0) {
      reader.join();
      printf("No usable HIP device: %s\n", bce_hip_strerror(rc));
      return -3;
    }
    const bool cli_timing = getenv("BCE_CLI_TIMING") != nullptr;
    auto lap = [&](const char *what) {
      if (cli_timing) fprintf(stderr, "cli: %-10s %.3f s\n", what, std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - start).count());
    };
    lap("create");
    // load_config, bce.cpp:626-641.  Read and validated ONCE (on the first context): `-c` and `-cN` apply the same 288
    // bytes or, with the same message, the defaults.
    std::vector<uint8_t> cfgbuf;
    if (argc == 5) {
      std::ifstream cfg(argv[4], std::ios::binary | std::ios::ate);
      std::streamoff size = cfg ? (std::streamoff)cfg.tellg() : -1;
      if (size != (std::streamoff)BCE_HIP_CONFIG_BYTES) {
        printf("Config not found or wrong size.\n");
      } else {
        cfgbuf.resize(BCE_HIP_CONFIG_BYTES);
        cfg.seekg(0, std::ios::beg);
        if (!cfg.read(reinterpret_cast<char *>(cfgbuf.data()), size)) { printf("Could not read Config.\n"); cfgbuf.clear(); }
        else if (bce_hip_set_config(ctx, cfgbuf.data()) != 0) { printf("Config rejected: %s\n", bce_hip_last_error(ctx)); cfgbuf.clear(); }
      }
    }
    reader.join();
    if (data.status != 0 || data.size() == 0 || data.size() >= file_limit) {   // also covers the empty file, on which the reference crashes
      printf("Error loading file\n");
      bce_hip_destroy(ctx);
      return -1;
    }
    lap("file read");
    // `-cN` (N = 2..64, an extension): N blocks in a BCEM container, spread over the GPUs of the node
    const uint32_t nblocks = (uint32_t)atoi(argv[1] + 2);
    if (nblocks >= 2 && nblocks <= 64 && data.size() >= nblocks) {
      bce_hip_destroy(ctx);
      std::vector<uint8_t> blob;
      rc = compress_blocks(data, nblocks, cfgbuf.empty() ? nullptr : cfgbuf.data(), blob);
      if (rc != 0) { printf("Compression failed: %s\n", bce_hip_strerror(rc)); return -4; }
      std::chrono::duration<double> duration = std::chrono::high_resolution_clock::now() - start;
      if (!write_file(argv[2], blob.data(), blob.size())) { printf("Could not write Archive.\n"); return -5; }
      printf("Compressed from %zu B -> %zu B in %.1f s\n", data.size(), blob.size(), duration.count());
      fast_exit(0);
      return 0;
    }
    size_t alen = 0;
    uint64_t prog = 0;
    bce_hip_set_progress(ctx, progress, &prog);
    rc = bce_hip_compress(ctx, data.data(), (uint32_t)data.size(), nullptr, 0, &alen);
    progress_end();
    lap("compress");
    if (rc != 0) {
      printf("Compression failed: %s (%s)\n", bce_hip_strerror(rc), bce_hip_last_error(ctx));
      bce_hip_destroy(ctx);
      return -4;
    }
    std::vector<uint8_t> arch(alen);
    bce_hip_archive_copy(ctx, arch.data(), arch.size());
    auto end = std::chrono::high_resolution_clock::now();
    std::chrono::duration<double> duration = end - start;
    if (!write_file(argv[2], arch.data(), arch.size())) { printf("Could not write Archive.\n"); bce_hip_destroy(ctx); return -5; }
    printf("Compressed from %zu B -> %zu B in %.1f s\n", data.size(), arch.size(), duration.count());
    lap("written");
    fast_exit(0);
    bce_hip_destroy(ctx);
    lap("destroyed");
    return 0;
  } else if (argc == 4 && argv[1][0] == '-' && argv[1][1] == 'd') {
    // Decompress (bce.cpp:1428-1472).
    auto start = std::chrono::high_resolution_clock::now();
    HostFile adata;
    const bool use_gpu = argv[1][2] != 's';
    std::thread reader(read_whole_file, argv[3], &adata, (size_t)0); // beside the runtime's start-up (-d)
    bce_hip_ctx *ctx0 = nullptr;
    int rc0 = use_gpu ? bce_hip_create(&ctx0, 0) : 0;
    reader.join();
    const bool cli_timing = getenv("BCE_CLI_TIMING") != nullptr;
    auto lap = [&](const char *what) {
      if (cli_timing) fprintf(stderr, "cli: %-10s %.3f s\n", what, std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - start).count());
    };
    lap("create");
    if (adata.status == -1) { printf("Archive not found.\n"); if (ctx0) bce_hip_destroy(ctx0); return -1; }
    if (adata.status != 0 || adata.size() == 0) { printf("Could not read Archive.\n"); if (ctx0) bce_hip_destroy(ctx0); return -2; }
    // -d: GPU-assisted decoder (kd_decode.hip), needs the GPU like -c.  -ds (the reference's low-memory unbwt variant,
    // :1466): the plain host decoder (decoder.cpp), on purpose and by name -- there is no silent fallback.
    // A BCEM container (bce -cN, the sharded bench) is decoded block by block.
    std::vector<std::pair<size_t, size_t>> blocks;               // (offset, length) of each archive inside adata
    if (is_container(adata)) {
      const uint32_t ver = (uint32_t)get_le(adata.data() + 4, 4), nb = (uint32_t)get_le(adata.data() + 8, 4);
      size_t pos = 12 + (size_t)nb * 16;
      if (ver != 1 || pos > adata.size()) { printf("Could not read Archive.\n"); return -2; }
      for (uint32_t b = 0; b < nb; ++b) {
        const size_t alen = (size_t)get_le(adata.data() + 12 + (size_t)b * 16 + 8, 8);
        if (alen > adata.size() - pos) { printf("Could not read Archive.\n"); return -2; }
        blocks.emplace_back(pos, alen);
        pos += alen;
      }
    } else {
      blocks.emplace_back(0, adata.size());
    }
    int rc = 0;
    // the output: plain memory, not value-initialised (a vector would write 10^8 zeroes first that the decoder overwrites)
    // (2 MB-aligned with huge pages asked for, and its pages faulted in by the kernel -- MADV_POPULATE_WRITE leaves the
    // contents alone -- on a thread beside the decoding: the last device-to-host copy lands in mapped memory)
    struct FreeDeleter { void operator()(uint8_t *q) const { free(q); } };
    std::unique_ptr<uint8_t, FreeDeleter> out;
    std::thread prefault;
    auto alloc_out = [&](size_t bytes) {
      void *q = nullptr;
      const size_t two_mb = (size_t)2 << 20, len = ((bytes ? bytes : 1) + two_mb - 1) & ~(two_mb - 1);
      if (posix_memalign(&q, two_mb, len) != 0) q = nullptr;
      out.reset(static_cast<uint8_t *>(q));
      if (q) {
        (void)madvise(q, len, MADV_HUGEPAGE);
        try { prefault = std::thread([q, len] { (void)madvise(q, len, 23 /* MADV_POPULATE_WRITE */); }); } catch (...) {}
      }
      return q != nullptr;
    };
    struct JoinOnExit { std::thread &t; ~JoinOnExit() { if (t.joinable()) t.join(); } } join_prefault{prefault};
    size_t out_size = 0;
    bce_hip_ctx *keep = nullptr;                                   // the one-block context: given back after the file is written
    if (blocks.size() == 1) {
      uint64_t prog = 0;
      bce_hip_ctx *ctx = ctx0;
      if (use_gpu) {
        if (rc0 != 0) {
          printf("No usable HIP device: %s (bce -ds decodes on the host)\n", bce_hip_strerror(rc0));
          return -3;
        }
        bce_hip_set_progress(ctx, progress, &prog);
      }
      const uint8_t *ap = adata.data() + blocks[0].first;
      size_t n = 0;
      rc = ctx ? bce_hip_decompress_device(ctx, ap, blocks[0].second, nullptr, 0, &n) : bce_hip_decompress(ap, blocks[0].second, nullptr, 0, &n);
      if (rc == 0) {
        if (!alloc_out(n)) { printf("Could not read Archive.\n"); if (ctx) bce_hip_destroy(ctx); return -2; }
        out_size = n;
        rc = ctx ? bce_hip_decompress_device(ctx, ap, blocks[0].second, out.get(), n, &n) : bce_hip_decompress(ap, blocks[0].second, out.get(), n, &n);
      }
      if (ctx) {
        progress_end();
        if (rc != 0) { printf("%s\n", bce_hip_last_error(ctx)); bce_hip_destroy(ctx); }
        else keep = ctx;
      }
      lap("decoded");
    } else {
      // a container: the blocks are independent, their sizes are in the table -- decoded side by side, two contexts per GPU
      // (a block's decoding is mostly its eight sequential range decoders on the host) or, for -ds, up to 8 host threads
      // The table's raw sizes are untrusted 64-bit values: each must be what the block's own header says (asked without
      // an output buffer: the header is parsed, nothing is written), at least one byte and below the encoder's 2^31 limit,
      // and the output is sized from the verified values only -- a wrapping or oversized table is "Could not read Archive".
      std::vector<size_t> at(blocks.size() + 1, 0);
      for (size_t b = 0; b < blocks.size(); ++b) {
        const uint64_t raw = get_le(adata.data() + 12 + b * 16, 8);
        size_t hn = 0;
        const int hr = bce_hip_decompress(adata.data() + blocks[b].first, blocks[b].second, nullptr, 0, &hn);
        if (hr != 0 || raw < 1 || raw >= 0x80000000ull || (uint64_t)hn != raw || raw > SIZE_MAX - at[b]) {
          printf("Could not read Archive.\n");
          return -2;
        }
        at[b + 1] = at[b] + (size_t)raw;
      }
      if (!alloc_out(at.back())) { printf("Could not read Archive.\n"); return -2; }
      out_size = at.back();
      std::vector<bce_hip_ctx *> ctxs;
      if (use_gpu) {
        int ndev = 0;
        for (int dev = 0; dev < 64 && ctxs.size() < blocks.size(); ++dev) {
          bce_hip_ctx *c = dev == 0 ? ctx0 : nullptr;
          if (!c && bce_hip_create(&c, dev) != 0) break;
          ctxs.push_back(c);
          ndev = dev + 1;
        }
        if (ctxs.empty()) { printf("No usable HIP device (bce -ds decodes on the host)\n"); return -3; }
        for (int dev = 0; dev < ndev && ctxs.size() < blocks.size(); ++dev) {
          bce_hip_ctx *c = nullptr;
          if (bce_hip_create(&c, dev) != 0) break;
          ctxs.push_back(c);
        }
      }
      const size_t workers = use_gpu ? ctxs.size() : std::min<size_t>(blocks.size(), 8);
      std::atomic<size_t> next_block{0};
      std::atomic<int> first_rc{0};
      // as in compress_blocks: a context that runs out of device memory (blocks of a GB and more, two contexts per device) gives
      // its memory back and leaves its block to ONE context that has the device to itself at the end
      std::vector<char> done(blocks.size(), 0);
      auto decode_block = [&](bce_hip_ctx *c, size_t b) -> int {
        const uint8_t *ap = adata.data() + blocks[b].first;
        size_t n = 0;
        const size_t want = at[b + 1] - at[b];
        int r = c ? bce_hip_decompress_device(c, ap, blocks[b].second, out.get() + at[b], want, &n)
                  : bce_hip_decompress(ap, blocks[b].second, out.get() + at[b], want, &n);
        if (r == 0 && n != want) r = BCE_HIP_E_INTERNAL;         // the table and the block's own header disagree
        return r;
      };
      std::vector<std::thread> th;
      for (size_t w = 0; w < workers; ++w)
        th.emplace_back([&, w] {
          while (first_rc.load() == 0) {
            const size_t b = next_block.fetch_add(1);
            if (b >= blocks.size()) break;
            bce_hip_ctx *c = use_gpu ? ctxs[w] : nullptr;
            const int r = decode_block(c, b);
            if (r == 0) { done[b] = 1; continue; }
            if (r == BCE_HIP_E_NOMEM && use_gpu) { bce_hip_destroy(ctxs[w]); ctxs[w] = nullptr; return; }   // (its block stays undone)
            int z = 0;
            first_rc.compare_exchange_strong(z, r);
          }
        });
      for (auto &t : th) t.join();
      rc = first_rc.load();
      for (bce_hip_ctx *&c : ctxs) { if (!c) continue; if (rc != 0 && bce_hip_last_error(c)[0]) printf("%s\n", bce_hip_last_error(c)); bce_hip_destroy(c); c = nullptr; }
      if (rc == 0 && use_gpu) {
        bce_hip_ctx *c = nullptr;
        for (size_t b = 0; rc == 0 && b < blocks.size(); ++b) {
          if (done[b]) continue;
          if (!c) rc = bce_hip_create(&c, 0);
          if (rc == 0) rc = decode_block(c, b);
          if (rc == 0) done[b] = 1;
        }
        if (rc != 0 && c && bce_hip_last_error(c)[0]) printf("%s\n", bce_hip_last_error(c));
        if (c) bce_hip_destroy(c);
      }
      if (rc == 0) for (size_t b = 0; b < blocks.size(); ++b) if (!done[b]) rc = BCE_HIP_E_INTERNAL;
    }
    if (rc != 0) {
      printf("Decompression failed: %s\n", bce_hip_strerror(rc));
      if (rc == BCE_HIP_E_NOMEM && use_gpu) printf("(the GPU-assisted decoder holds 32 bytes of boundary ranks per input byte beside its node lists; `bce -ds` decodes on the host)\n");
      return -4;
    }
    auto end = std::chrono::high_resolution_clock::now();
    std::chrono::duration<double> duration = end - start;
    if (prefault.joinable()) prefault.join();
    if (!write_file(argv[2], out.get(), out_size)) { printf("Could not write file.\n"); if (keep) bce_hip_destroy(keep); return -5; }
    printf("Decompressed from %zu B -> %zu B in %.1f s\n", adata.size(), out_size, duration.count());
    lap("written");
    fast_exit(0);
    if (keep) bce_hip_destroy(keep);
    return 0;
  } else if (argc == 4 && argv[1][0] == '-' && argv[1][1] == 's') {
    // Scan (bce.cpp:1384-1402): enumeration on the GPU, ScanCoder optimisation on the host, 288-byte config out
    auto start = std::chrono::high_resolution_clock::now();
    HostFile data;
    std::thread reader(read_whole_file, argv[3], &data, kMaxInput);
    bce_hip_ctx *ctx = nullptr;
    int rc = bce_hip_create(&ctx, 0);
    reader.join();
    if (rc != 0) { printf("No usable HIP device: %s\n", bce_hip_strerror(rc)); return -3; }
    if (data.status != 0 || data.size() == 0 || data.size() >= kMaxInput) { printf("Error loading file\n"); bce_hip_destroy(ctx); return -1; }
    uint8_t cfg[BCE_HIP_CONFIG_BYTES];
    double res[9];
    rc = bce_hip_load_host(ctx, data.data(), (uint32_t)data.size());
    if (rc == 0) rc = bce_hip_bwt(ctx, nullptr);
    if (rc == 0) rc = bce_hip_build_planes(ctx, nullptr);
    uint64_t prog = 0;
    bce_hip_set_progress(ctx, progress, &prog);
    if (rc == 0) rc = bce_hip_scan(ctx, cfg, res);
    progress_end();
    if (rc != 0) { printf("Scan failed: %s (%s)\n", bce_hip_strerror(rc), bce_hip_last_error(ctx)); bce_hip_destroy(ctx); return -4; }
    for (int i = 0; i < 9; ++i) printf("Result size: %.1f B\n", res[i]);            // ScanCoder::flush, :799
    if (!write_file(argv[2], cfg, BCE_HIP_CONFIG_BYTES)) { printf("Could not write Config.\n"); bce_hip_destroy(ctx); return -5; }   // save_config, :810-813
    auto end = std::chrono::high_resolution_clock::now();
    std::chrono::duration<double> duration = end - start;
    printf("Scanned %zu B in %.1f s\n", data.size(), duration.count());
    fast_exit(0);
    bce_hip_destroy(ctx);
    return 0;
  }
  return usage();
}
