/*
 * bce_hip.h -- C ABI of libbcehip.so, the MI355X (gfx950) implementation of the `bce -c` hot path.
 *
 * The reference (akamiru/bce, /root/reference/bce.cpp) exposes no FFI; its seams are C++ template
 * policies inside one translation unit plus the libdivsufsort C ABI.  Each entry point below names
 * the reference interface it replaces (file:line).  All functions return 0 on success and a negative
 * bce_hip_status on failure (the reference's convention is status ints + printf, no exceptions:
 * CMakeLists.txt:25).  The context owns every device buffer; callers own every host buffer they
 * pass.  A context is used from one host thread at a time.
 */
#ifndef BCE_HIP_H
#define BCE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bce_hip_ctx bce_hip_ctx;

enum bce_hip_status {
  BCE_HIP_OK = 0,
  BCE_HIP_E_ARG = -1,       /* bad argument (null pointer, n == 0, n >= 2^31, wrong config size) */
  BCE_HIP_E_DEVICE = -2,    /* HIP runtime error (no device, launch failure, ...) */
  BCE_HIP_E_NOMEM = -3,     /* device or host allocation failed */
  BCE_HIP_E_STATE = -4,     /* stage called out of order */
  BCE_HIP_E_OVERFLOW = -5,  /* output buffer too small / internal capacity exceeded */
  BCE_HIP_E_INTERNAL = -6   /* device-side consistency check failed */
};

#define BCE_HIP_CONFIG_BYTES 288u /* (31+1)*9, bce.cpp:629 */

/* ---- lifetime ---------------------------------------------------------------------------------- */
int bce_hip_create(bce_hip_ctx **out, int device);
/* The same for a one-shot caller that knows the size of the input it is about to compress (`bce -c`: the file's size, main()
 * bce.cpp:1403-1427): host-side preparation that depends on it -- the pinned staging of the model flushes, 8 bytes per
 * symbol record of a flush in three slots: 3 x 8 MB from 128 KB of input, 3 x 128 MB from 2 MB up -- runs on threads of
 * its own beside the HIP runtime's initialisation instead of inside the first compression.  A hint only: any input may
 * follow; 0 = unknown (= bce_hip_create). */
int bce_hip_create_sized(bce_hip_ctx **out, int device, uint64_t expected_input_bytes);
void bce_hip_destroy(bce_hip_ctx *ctx);
const char *bce_hip_strerror(int status);
/* last HIP error string seen by this context (for diagnostics) */
const char *bce_hip_last_error(const bce_hip_ctx *ctx);

/* ---- configuration ----------------------------------------------------------------------------- */
/* AdaptiveCoder<31>::load_config (bce.cpp:626-641): 9 rows x 32 context-bit counts.  NULL restores
 * the built-in defaults (bce.cpp:713-724). */
int bce_hip_set_config(bce_hip_ctx *ctx, const uint8_t *config288);
/* Several contexts on ONE device (a stream of files or blocks, `bce -cN` with more blocks than GPUs): a compression is a
 * GPU phase followed by a host phase in which the eight coder threads finish the last batches, so contexts driven by
 * one host thread each overlap one's coding -- and one's rotation sort -- with the other's enumeration.  Gated contexts
 * of a device take turns for the ENUMERATION (its single-launch rounds must not run beside another context's):
 * bce_hip_encode / bce_hip_scan take the device's gate, give it back when the last model flush is queued or when they
 * fail, and lend it to the next context while they wait for their own coder threads and -- since round 4 -- behind every model
 * flush they queue (the model's kernels wait for nothing; the gate comes back before the next round is queued);
 * bce_hip_destroy and this call (with either value) give it back too.  Every context is gated by default (an uncontended gate costs nothing); contexts of
 * other PROCESSES on the same device are kept apart by an advisory file lock keyed by the device's PCI address
 * (/dev/shm/bce_hip_gate_<bdf>; BCE_HIP_NO_FILE_GATE=1 switches that part off).  on = 0 opts a context out: only for
 * a context that is alone on its device. */
int bce_hip_set_gated(bce_hip_ctx *ctx, int on);
/* capacity (in symbol records) of the device symbol buffer between model flushes; 0 = automatic */
int bce_hip_set_symbol_capacity(bce_hip_ctx *ctx, uint64_t records);

/* Progress of BCE::code (the reference prints "Coded: %u.%02u %%\r" from it, bce.cpp:1354-1358): called from the calling
 * thread with the nodes visited so far and the total (8n) whenever the host looks at the enumeration's state, during
 * bce_hip_encode / _compress / _scan / _decompress_device.  NULL switches it off (default). */
typedef void (*bce_hip_progress_fn)(uint64_t nodes_done, uint64_t nodes_total, void *user);
int bce_hip_set_progress(bce_hip_ctx *ctx, bce_hip_progress_fn fn, void *user);

/* test knobs for the enumeration's alternative code paths (all 0 by default): 0 = nodes a depth-first walker
 * classifies per pass, 1 = disable the depth-first tail, 2 = disable the persistent LDS tail kernel, 3 = disable chain skipping, 4 = disable the one-launch kernel for narrow rounds, 6 = three launches per wide round (separate scan kernel) instead of two, 7 = disable the workgroup-local rounds before the depth-first tail,
 * 8 = live nodes below which the walkers take over from the workgroup-local rounds (default 16 384; the tail itself starts at 1 M live nodes with the local rounds, at 65 536 without them: BCE_HIP_DFS_ENTER), 9 = rounds a workgroup runs per pass before it hands on (default 192),
 * 10 = round from which the tail may start although the node count still grows (exercises the spill path),
 * 11 = model flushes (K4) on a stream of their own beside the next K3 rounds, double-buffered symbol records (also BCE_HIP_OVERLAP=1),
 * 12 = d: the node lists start with n / d + 4096 nodes instead of n / 8 + 4096 (also BCE_HIP_CAPP_DIV=d; large d: the lists grow many times).
 * The archive never depends on them. */
int bce_hip_debug_set(bce_hip_ctx *ctx, int knob, uint32_t value);

/* ---- stage 0: input ---------------------------------------------------------------------------- */
/* File::File (bce.cpp:842-856): take the n input bytes.  _host copies host->HBM, _device copies
 * HBM->HBM from a device pointer of the same GPU (input already resident). 1 <= n < 2^31.
 * Capacity: device memory is ~45 n bytes for the suffix sort and planes plus the enumeration's node lists: 16 lists (8 planes x
 * the two parities of a round) that start with n/8 nodes each (text fills 0.02-0.03 n, random bytes 0.15-0.3 n).  A round whose
 * children do not fit is not run: the lists it would write -- the other parity's, empty at that moment -- are replaced by larger
 * ones (bce_hip_stats.list_grows; no copy, never old and new side by side), up to the worst case of n/2 + 2 nodes, and when the
 * device runs out of memory the buffers of the stages that are not running (the suffix sort's scratch) go back first.  Every
 * valid input -- any 1 <= n < 2^31, as the reference (bce.cpp:173,374,901) -- fits an otherwise idle 288 GB MI355X: the worst
 * case is 72 n bytes of lists beside 13 n bytes that stay.  A round that emits more symbols than one model flush takes (2^31
 * records) is run plane group by plane group (bce_hip_stats.split_rounds).  (The GPU-assisted DECODER, bce_hip_decompress_device,
 * holds 32 n bytes of boundary ranks beside its lists and a round's queries: a high-entropy archive of more than ~1.6 * 10^9 bytes
 * ends in BCE_HIP_E_NOMEM there -- never in wrong bytes -- and is decoded by bce_hip_decompress, `bce -ds`.) */
int bce_hip_load_host(bce_hip_ctx *ctx, const uint8_t *in, uint32_t n);
int bce_hip_load_device(bce_hip_ctx *ctx, const void *d_in, uint32_t n);

/* ---- stage 1 (K1): rotation + BWT --------------------------------------------------------------- */
/* File::rotate + File::bwt (bce.cpp:858-910), i.e. the libdivsufsort call
 *   saidx_t divbwt(const sauchar_t *T, sauchar_t *U, saidx_t *A, saidx_t n)   (bce.cpp:901)
 * plus the two std::rotate fix-ups: sorts all cyclic rotations on the GPU, leaves the n-byte BWT in
 * HBM and returns offset_ (index of the first minimal rotation, bce.cpp:893). */
int bce_hip_bwt(bce_hip_ctx *ctx, uint32_t *offset);
/* Test hook: inject a BWT computed elsewhere instead of running K1. */
int bce_hip_set_bwt(bce_hip_ctx *ctx, const uint8_t *bwt, uint32_t n, uint32_t offset);
/* Copy the BWT bytes back (n bytes). */
int bce_hip_get_bwt(bce_hip_ctx *ctx, uint8_t *bwt_out);

/* The libdivsufsort seam itself (bce.cpp:901, :1091): the two functions of that library's C ABI which the reference calls,
 * with their argument meaning, on a context (include/divsufsort_hip.h + libdivsufsort_hip.so wrap them under their
 * original names and signatures, so that an unmodified bce.cpp links against them).  Host buffers; in == out allowed.
 * divbwt: BWT of in[0, n) with an implicit smallest sentinel, *primary = 1-based row of suffix 0.  The context's
 * compression state is dropped by either call. */
int bce_hip_divbwt(bce_hip_ctx *ctx, const uint8_t *in, uint8_t *out, uint32_t n, uint32_t *primary);
int bce_hip_inverse_bwt(bce_hip_ctx *ctx, const uint8_t *in, uint8_t *out, uint32_t n, uint32_t primary);

/* ---- stage 2 (K2): wavelet-matrix bit planes + rank directory ----------------------------------- */
/* RankFile ctor body + Rank::build (bce.cpp:944-970, 138-145).  zeros[j] = rank0_j(n). */
int bce_hip_build_planes(bce_hip_ctx *ctx, uint32_t zeros[8]);
/* Test hook: plane j, positions [0,n) -> one byte (0/1) each, as Rank::bit (bce.cpp:196-198). */
int bce_hip_get_plane_bits(bce_hip_ctx *ctx, int plane, uint8_t *bits_out);
/* Test hook: rank1_j(index) for an array of query positions, as Rank::get<1> (bce.cpp:147-151). */
int bce_hip_rank1(bce_hip_ctx *ctx, int plane, const uint32_t *idx, uint32_t count, uint32_t *out);

/* ---- stage 3+4 (K3, K4) + host coder: BCE::encode ------------------------------------------------ */
/* BCE::encode (bce.cpp:1117-1167): runs BCE::code mode 1 (bce.cpp:1236-1374) as per-round GPU
 * passes (K3), the AdaptiveCoder model half (bce.cpp:506-518,531-533,671-677) on the GPU (K4), the
 * range-coder half (bce.cpp:520-529,538-553,610-615,655-661) on 8 host threads, then frames the
 * archive (bce.cpp:1140-1157).  The archive is kept in the context until the next load. */
int bce_hip_encode(bce_hip_ctx *ctx);
/* size of / copy of the finished archive (native-endian u16 words, bce.cpp:1426) */
int bce_hip_archive_size(bce_hip_ctx *ctx, size_t *bytes);
int bce_hip_archive_copy(bce_hip_ctx *ctx, uint8_t *out, size_t cap);

/* ---- extension: ONE archive from several contexts / GPUs (SURVEY section 8e-2's aim) ------------------
 * The eight plane coders of BCE::encode (bce.cpp:1124-1150) are independent sequential streams; the archive is header +
 * the eight streams (:1152-1157).  A context codes only the planes of `mask` (bit p = plane p; default 0xFF; set before
 * bce_hip_encode, it stays until changed); after bce_hip_encode the finished stream of a plane it owns can be read, and
 * the stream of a plane another context owns can be put in its place -- the header is coded again from the new sizes --
 * so that bce_hip_archive_size / _copy give the archive `bce -c` writes.  Every context must have encoded the same input
 * with the same config.  Streams are native-endian u16 words. */
int bce_hip_set_plane_mask(bce_hip_ctx *ctx, uint32_t mask);
int bce_hip_plane_stream_size(bce_hip_ctx *ctx, int plane, size_t *words);
int bce_hip_plane_stream_copy(bce_hip_ctx *ctx, int plane, uint16_t *out, size_t cap_words);
int bce_hip_plane_stream_set(bce_hip_ctx *ctx, int plane, const uint16_t *words, size_t count);

/* ---- one-shot: main() -c branch minus file I/O (bce.cpp:1403-1427) -------------------------------- */
int bce_hip_compress(bce_hip_ctx *ctx, const uint8_t *in, uint32_t n, uint8_t *out, size_t cap, size_t *out_len);
/* same with the input already in HBM */
int bce_hip_compress_device(bce_hip_ctx *ctx, const void *d_in, uint32_t n, uint8_t *out, size_t cap, size_t *out_len);

/* ---- stepping interface for parity tests (BCE::code one round at a time) --------------------------- */
/* begin: sets up the roots (bce.cpp:1237-1240).  round: runs one round over all 8 planes and
 * reports how many nodes the NEXT round holds.  nodes: copies plane p's current node list as
 * (s absolute, x0, x1) triples, sorted by s (j=0 list then j=1 list, bce.cpp:1256-1264). */
int bce_hip_enum_begin(bce_hip_ctx *ctx);
int bce_hip_enum_nodes(bce_hip_ctx *ctx, int plane, uint32_t *triples_out, uint32_t cap_nodes, uint32_t *count);
int bce_hip_enum_round(bce_hip_ctx *ctx, uint64_t *next_nodes);
/* symbol records emitted so far and not yet flushed, in (round, plane, s) order.  Each record:
 * out[6*i+0..5] = plane, s, k, nesc, escbits, slot  (s,k after the k>31 escape, bce.cpp:507-510) */
int bce_hip_enum_symbols(bce_hip_ctx *ctx, uint32_t *out, uint64_t cap_records, uint64_t *count);
/* run K4 on the symbols emitted so far: out[3*i+0..2] = cum, freq, total per record (same order) */
int bce_hip_enum_model(bce_hip_ctx *ctx, uint32_t *out, uint64_t cap_records, uint64_t *count);

/* ---- config scan (SURVEY section 8f "next #2") ---------------------------------------------------------------- */
/* main() -s branch (bce.cpp:1384-1402): BCE<ScanCoder<31>, noop>::encode + ScanCoder::save_config
 * (bce.cpp:726-834).  Call after bce_hip_build_planes: the enumeration runs on the GPU, the eight ScanCoders and
 * their cost optimisation on the host.  config288 = the .bcc file content; result_bytes[9] (optional) = the values
 * of the nine "Result size: %.1f B" lines (bce.cpp:799). */
int bce_hip_scan(bce_hip_ctx *ctx, uint8_t config288[BCE_HIP_CONFIG_BYTES], double result_bytes[9]);

/* ---- decoder (SURVEY section 8f "next #1"): bce_hip_decompress = plain host C++ (`bce -ds`),
 *      bce_hip_decompress_device = the GPU-assisted decoder of kd_decode.hip (`bce -d`); same bytes ---------------- */
/* BCE::decode + unbwt::bytewise + inverse BWT + rotate (bce.cpp:1169-1233, 1043-1102): archive -> original bytes.
 * out == NULL: only report the decoded size in *out_len.  Needs no context and no GPU. */
int bce_hip_decompress(const uint8_t *archive, size_t len, uint8_t *out, size_t cap, size_t *out_len);
/* Same result with the GPU doing everything but the eight sequential range decoders + adaptive models (kd_decode.hip):
 * node classification, children and boundary ranks per round on the device, the decoders' answers on 8 host threads,
 * then plane fill, unbwt::bytewise as wavelet-matrix access and the inverse BWT on the device.  Uses the context's
 * device, stream and scratch buffers; a compression in progress in the same context is dropped. */
int bce_hip_decompress_device(bce_hip_ctx *ctx, const uint8_t *archive, size_t len, uint8_t *out, size_t cap, size_t *out_len);

/* ---- statistics of the last bce_hip_encode / bce_hip_compress ------------------------------------ */
typedef struct bce_hip_stats {
  uint64_t n;            /* input bytes */
  uint64_t nodes;        /* nodes visited (= 8n-8 on primitive inputs) */
  uint64_t symbols;      /* coded symbols (adaptive calls, bce.cpp:1302) */
  uint32_t rounds;       /* rounds of BCE::code */
  uint32_t sort_rounds;  /* prefix-doubling rounds of K1 */
  uint32_t flushes;      /* K4 model flushes */
  uint32_t spine_levels; /* byte levels of the enumeration's tail done by spine bursts (k3_dfs.hip) */
  /* t_load, t_bwt, t_planes, t_total: host wall seconds.  t_enum, t_model: GPU seconds (HIP events) of K3 and of
   * K4 + device-to-host copies.  t_coder: host seconds spent waiting for the coder threads (the part of the range
   * coding that nothing hides).  The last three overlap, so they do not add up to t_total. */
  double t_load, t_bwt, t_planes, t_enum, t_model, t_coder, t_total;
  double k3_ms, k3_launches;   /* HIP-event time and launch count of the interval-count kernels */
  double t_coder_busy;         /* busiest host coder thread (t_coder is only the part not hidden behind GPU work) */
  double list_grows;           /* times a round did not fit the node lists and they were doubled (k3_grow_lists) */
  double list_nodes;           /* nodes per list at the end (the larger of the two parities) */
  double split_rounds;         /* rounds whose symbols did not fit one model flush and were run plane group by plane group */
  /* since the context was created (not reset by a load): */
  double reg_maps;             /* host mappings registered with the runtime (flush slots, the decoder's boundary ranks) */
  double reg_unmaps;           /* ... and given back (after waiting for the work that touches them) */
  double dec_restarts;         /* GPU-assisted decodes started again with larger node lists (kd_decode.hip) */
  double t_model_kernels;      /* of t_model: GPU seconds of K4's kernels alone (sort, window, long, emit), without the device-to-host copies */
} bce_hip_stats;
int bce_hip_get_stats(const bce_hip_ctx *ctx, bce_hip_stats *out);

/* ---- synthetic inputs (SURVEY.md section 8c generators; host side, for benchmarks and tests) ------- */
void bce_hip_synth_text(uint64_t seed, uint8_t *out, size_t n);
void bce_hip_synth_rand(uint64_t seed, uint8_t *out, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* BCE_HIP_H */
