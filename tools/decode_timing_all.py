#!/usr/bin/env python3
"""Compress the three 10^8-byte workloads and decode each with BCE_DEC_TIMING=1 (stage seconds on stderr).
   python tools/decode_timing_all.py [text|natural|binary ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BCE_DEC_TIMING", "1")
import bce_amd  # noqa: E402
from scan_time import load  # noqa: E402

for kind in sys.argv[1:] or ["text", "natural", "binary"]:
    d = load(kind)
    arch = bce_amd.compress(d)
    ctx = bce_amd.api._Ctx(0)
    for i in range(2):
        t0 = time.time()
        back = bce_amd.decompress_device(arch, ctx=ctx)
        dt = time.time() - t0
        print("%-8s decode %.3f s  %.1f MB/s" % (kind, dt, len(back) / dt / 1e6), flush=True)
    sys.stderr.flush()
    ctx.close() if hasattr(ctx, "close") else None
