"""bce_amd -- MI355X-native (gfx950) implementation of the `bce -c` hot path of akamiru/bce.

The product is bce_amd/lib/libbcehip.so (hand-written HIP kernels behind the C ABI of
include/bce_hip.h) plus the `bce` command line (bce_amd/bin/bce).  This package is the thin Python
binding used by tests and bench.py; it mirrors the reference's own interface names
(RankFile, BCE.encode, `bce -c archive file [config]`, bce.cpp:932-984,1117-1167,1403-1427).
There is no CPU fallback: every compute call goes through the HIP library and fails loudly
when it is missing or when no GPU is present.
"""
from .api import (BCE, BceError, ContextPool, RankFile, compress, compress_device, compress_many, decompress, decompress_device,  # noqa: F401
                  library_path, load_library, scan, synth_rand, synth_text, stats, stats_of, archive_of, plane_stream, set_plane_mask, set_plane_stream)
from .build import build as build_native  # noqa: F401
