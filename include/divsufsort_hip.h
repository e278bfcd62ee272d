/*
 * divsufsort_hip.h -- the libdivsufsort C ABI that akamiru/bce calls, served by the MI355X rotation sorter (K1) and the
 * GPU inverse BWT of libbcehip.so.  Link libdivsufsort_hip.so (bce_amd/lib) in place of libdivsufsort and an UNMODIFIED
 * bce.cpp resolves its two calls here:
 *     bce.cpp:901    divbwt(file_.data(), file_.data(), 0, file_.size() - 1)
 *     bce.cpp:1091   inverse_bw_transform(out.data(), out.data(), nullptr, n, 1)
 * Types and signatures are libdivsufsort's (divsufsort.h: sauchar_t = uint8_t, saidx_t = saint_t = int32_t); install this
 * file as divsufsort.h, or keep libdivsufsort's own header -- the two declare the same functions.
 *
 * The functions use one process-wide GPU context (device $BCE_HIP_DEVICE, default 0), created on first use and guarded
 * by a mutex; buffers are host pointers, T and U may be the same buffer (bce calls them in place); the workspace A is
 * ignored (libdivsufsort allocates its own when it is NULL, which is what bce passes).
 */
#ifndef DIVSUFSORT_HIP_H
#define DIVSUFSORT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint8_t sauchar_t;
typedef int32_t saint_t;
typedef int32_t saidx_t;

/* Burrows-Wheeler transform of T[0, n) with an implicit smallest sentinel: U[0] = T[n-1], then T[SA[i] - 1] for the
 * suffixes in sorted order with the suffix 0 left out.  Returns its 1-based position (the primary index, 1..n), 0 for
 * n = 0, -1 on bad arguments, -2 when memory (host or device) runs out or no GPU is usable.  n < 2^31 - 1. */
saidx_t divbwt(const sauchar_t *T, sauchar_t *U, saidx_t *A, saidx_t n);

/* Inverse of divbwt: T = the transformed bytes, idx = the primary index it returned; U receives the text.
 * Returns 0, -1 on bad arguments (idx outside 1..n for n > 0, or bytes that are no BWT), -2 as above. */
saint_t inverse_bw_transform(const sauchar_t *T, sauchar_t *U, saidx_t *A, saidx_t n, saidx_t idx);

#ifdef __cplusplus
}
#endif
#endif /* DIVSUFSORT_HIP_H */
