#!/usr/bin/env python3
"""Compress synth-text 10^9 B and time two GPU-assisted decodes in one context (BCE_DEC_TIMING=1 for the stage seconds)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bce_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
d = bce_amd.synth_text(1, n)
t0 = time.time()
arch = bce_amd.compress(d)
print("compress %.2f s, %d B" % (time.time() - t0, len(arch)), flush=True)
ctx = bce_amd.api._Ctx(0)
for i in range(2):
    t0 = time.time()
    back = bce_amd.decompress_device(arch, ctx=ctx)
    dt = time.time() - t0
    print("decode %.2f s  %.1f MB/s  %s" % (dt, n / dt / 1e6, "identical" if back == d.tobytes() else "DIFFERENT"), flush=True)
ctx.close()
