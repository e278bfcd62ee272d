// main.cpp -- the `bce` command line, mirroring the reference's main() (bce.cpp:1376-1484):
//   bce -c archive.bce file [config.bcc]    compress on the MI355X through libbcehip.so
// Banner, usage text, summary line, argument detection and exit codes follow the reference
// (banner :1377-1379, -c :1403-1427, -d :1428-1472, usage :1473-1483).  -d uses the GPU-assisted decoder (kd_decode.hip), -ds the host decoder (decoder.cpp);
// -s runs the enumeration on the GPU in scan mode and the ScanCoder optimisation on the host (scan_coder.cpp).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <fstream>
#include <string>
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
  return 0;
}

int main(int argc, char **argv) {
  printf("BCE v0.4 Release\n");
  printf("Copyright (C) 2016  Christoph Diegelmann\n");
  printf("This is free software under GNU Lesser General Public License. See <http://www.gnu.org/licenses/lgpl>\n\n");

  if ((argc == 4 || argc == 5) && argv[1][0] == '-' && argv[1][1] == 'c') {
    auto start = std::chrono::high_resolution_clock::now();
    bce_hip_ctx *ctx = nullptr;
    int rc = bce_hip_create(&ctx, 0);
    if (rc != 0) {
      printf("No usable HIP device: %s\n", bce_hip_strerror(rc));
      return -3;
    }
    const bool cli_timing = getenv("BCE_CLI_TIMING") != nullptr;
    auto lap = [&](const char *what) {
      if (cli_timing) fprintf(stderr, "cli: %-10s %.3f s\n", what, std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - start).count());
    };
    lap("create");
    if (argc == 5) {   // load_config, bce.cpp:626-641
      std::ifstream cfg(argv[4], std::ios::binary | std::ios::ate);
      std::streamoff size = cfg ? (std::streamoff)cfg.tellg() : -1;
      if (size != (std::streamoff)BCE_HIP_CONFIG_BYTES) {
        printf("Config not found or wrong size.\n");
      } else {
        std::vector<uint8_t> buf(BCE_HIP_CONFIG_BYTES);
        cfg.seekg(0, std::ios::beg);
        if (!cfg.read(reinterpret_cast<char *>(buf.data()), size)) printf("Could not read Config.\n");
        else if (bce_hip_set_config(ctx, buf.data()) != 0) printf("Config rejected: %s\n", bce_hip_last_error(ctx));
      }
    }
    std::ifstream file(argv[3], std::ios::binary | std::ios::ate);   // File::File, bce.cpp:842-856
    std::streamoff fsize = file ? (std::streamoff)file.tellg() : -1;
    std::vector<uint8_t> data;
    bool ok = fsize > 0 && fsize < (std::streamoff)0x80000000ll;
    if (ok) {
      data.resize((size_t)fsize);
      file.seekg(0, std::ios::beg);
      ok = (bool)file.read(reinterpret_cast<char *>(data.data()), fsize);
    }
    if (!ok) {   // also covers the empty file, on which the reference crashes
      printf("Error loading file\n");
      bce_hip_destroy(ctx);
      return -1;
    }
    lap("file read");
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
    printf("Compressed from %zu B -> %zu B in %.1f s\n", data.size(), arch.size(), duration.count());
    std::ofstream archive(std::string(argv[2]), std::ios::binary | std::ios::trunc);
    archive.write(reinterpret_cast<const char *>(arch.data()), (std::streamsize)arch.size());
    bce_hip_destroy(ctx);
    return 0;
  } else if (argc == 4 && argv[1][0] == '-' && argv[1][1] == 'd') {
    // Decompress (bce.cpp:1428-1472).
    auto start = std::chrono::high_resolution_clock::now();
    std::ifstream archive(std::string(argv[3]), std::ios::binary | std::ios::ate);
    std::streamoff size = archive ? (std::streamoff)archive.tellg() : -1;
    if (size < 0) { printf("Archive not found.\n"); return -1; }
    std::vector<uint8_t> adata((size_t)size);
    archive.seekg(0, std::ios::beg);
    if (size == 0 || !archive.read(reinterpret_cast<char *>(adata.data()), size)) { printf("Could not read Archive.\n"); return -2; }
    // -d: GPU-assisted decoder (kd_decode.hip), needs the GPU like -c.  -ds (the reference's low-memory unbwt variant,
    // :1466): the plain host decoder (decoder.cpp), on purpose and by name -- there is no silent fallback.
    size_t n = 0;
    int rc;
    std::vector<uint8_t> out;
    uint64_t prog = 0;
    if (argv[1][2] != 's') {
      bce_hip_ctx *ctx = nullptr;
      rc = bce_hip_create(&ctx, 0);
      if (rc != 0) {
        printf("No usable HIP device: %s (bce -ds decodes on the host)\n", bce_hip_strerror(rc));
        return -3;
      }
      bce_hip_set_progress(ctx, progress, &prog);
      rc = bce_hip_decompress_device(ctx, adata.data(), adata.size(), nullptr, 0, &n);
      if (rc == 0) { out.resize(n); rc = bce_hip_decompress_device(ctx, adata.data(), adata.size(), out.data(), out.size(), &n); }
      progress_end();
      if (rc != 0) printf("%s\n", bce_hip_last_error(ctx));
      bce_hip_destroy(ctx);
    } else {
      rc = bce_hip_decompress(adata.data(), adata.size(), nullptr, 0, &n);
      if (rc == 0) { out.resize(n); rc = bce_hip_decompress(adata.data(), adata.size(), out.data(), out.size(), &n); }
    }
    if (rc != 0) { printf("Decompression failed: %s\n", bce_hip_strerror(rc)); return -4; }
    auto end = std::chrono::high_resolution_clock::now();
    std::chrono::duration<double> duration = end - start;
    printf("Decompressed from %zu B -> %zu B in %.1f s\n", adata.size(), out.size(), duration.count());
    std::ofstream file(std::string(argv[2]), std::ios::binary | std::ios::trunc);
    file.write(reinterpret_cast<const char *>(out.data()), (std::streamsize)out.size());
    return 0;
  } else if (argc == 4 && argv[1][0] == '-' && argv[1][1] == 's') {
    // Scan (bce.cpp:1384-1402): enumeration on the GPU, ScanCoder optimisation on the host, 288-byte config out
    auto start = std::chrono::high_resolution_clock::now();
    bce_hip_ctx *ctx = nullptr;
    int rc = bce_hip_create(&ctx, 0);
    if (rc != 0) { printf("No usable HIP device: %s\n", bce_hip_strerror(rc)); return -3; }
    std::ifstream file(argv[3], std::ios::binary | std::ios::ate);
    std::streamoff fsize = file ? (std::streamoff)file.tellg() : -1;
    std::vector<uint8_t> data;
    bool ok = fsize > 0 && fsize < (std::streamoff)0x80000000ll;
    if (ok) {
      data.resize((size_t)fsize);
      file.seekg(0, std::ios::beg);
      ok = (bool)file.read(reinterpret_cast<char *>(data.data()), fsize);
    }
    if (!ok) { printf("Error loading file\n"); bce_hip_destroy(ctx); return -1; }
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
    std::ofstream f(std::string(argv[2]), std::ios::binary | std::ios::trunc);     // save_config, :810-813
    f.write(reinterpret_cast<const char *>(cfg), BCE_HIP_CONFIG_BYTES);
    auto end = std::chrono::high_resolution_clock::now();
    std::chrono::duration<double> duration = end - start;
    printf("Scanned %zu B in %.1f s\n", data.size(), duration.count());
    bce_hip_destroy(ctx);
    return 0;
  }
  return usage();
}
