// k3_args.h -- kernel argument block and constants shared by the K3 kernels (k3_enumerate.hip, k3_dfs.hip).
#pragma once
#include "common.h"

namespace bce {

constexpr int K3_T = 256;
#ifndef K3_NPT_VALUE
#define K3_NPT_VALUE 4
#endif
constexpr int K3_NPT = K3_NPT_VALUE;            // nodes per thread
constexpr uint32_t K3_TILE = K3_T * K3_NPT;     // 1024 nodes per tile
constexpr uint32_t K3_MAXBATCH = 256;           // rounds per run-table batch

struct K3Args {
  EnumCtl *ctl;
  Node *nodes[2];         // [parity] -> [8][cap[parity]]: the two parities' lists are buffers of their own (a round reads one
                          // parity and writes the other, so the one that is written can be replaced by a larger one without a copy)
  const Granule *gran;    // [8][ngran]
  const PlaneCfg *cfg;    // [8]
  uint32_t *symkey;       // symbol records: key words (K4 sort input)
  uint32_t *symesc;       // symbol records: escape words
  uint32_t *scanrec;      // scan mode (`bce -s`): one scan_pack word per record instead of key/escape words
  uint32_t *tilecnt;      // [tiles][4]
  uint32_t *tileoff;      // [tiles][4]
  RunEntry *runs;         // [K3_MAXBATCH][8]
  // two-launch rounds (k3_count2_kernel): per-tile count words, per-group (256 tiles of a plane) totals / offsets
  unsigned long long *tw, *gwa, *gwb;
  uint32_t *goff;         // [groups][4]
  uint32_t fused;         // the write kernel adds goff to the (group-relative) tile offsets
  uint32_t cap[2];        // nodes per list of each parity
  uint32_t cap32;         // lists up to this many nodes are read with 32-bit byte offsets (K3_CAP32; BCE_HIP_CAP32 lowers it for tests)
  uint32_t ngran, n;
  uint32_t zeros[8];
  uint32_t par, round, run_slot;
  uint32_t repeat;        // a further pass over a round that has been counted into nodes_total already (a round whose symbols do
                          // not fit one flush is run once per group of planes, pmask selecting whose symbols are recorded; the
                          // children are written again, the same values to the same places)
  uint32_t pmask;         // planes whose symbols are recorded (bit p; 0xFF = all).  One archive from several contexts
                          // (bce_hip_set_plane_mask): the rounds of this file emit no record for a plane another context codes, so
                          // the model, its sort and the device-to-host copy shrink with the mask.  (The tail's kernels, k3_dfs.hip,
                          // record every plane: a few per cent of the symbols, which the masked coders skip.)
};

__device__ __forceinline__ uint32_t list_cap(const K3Args &a, uint32_t par) { return par ? a.cap[1] : a.cap[0]; }
__device__ __forceinline__ Node *plane_nodes(const K3Args &a, uint32_t par, uint32_t p) {
  return (par ? a.nodes[1] : a.nodes[0]) + (size_t)p * list_cap(a, par);
}

K3Args k3_make_args(bce_hip_ctx *c, uint32_t round, uint32_t run_slot);   // k3_enumerate.hip

}  // namespace bce
