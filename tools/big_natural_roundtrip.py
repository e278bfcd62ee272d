#!/usr/bin/env python3
"""Round trip of natural corpus || binary corpus || synth-text (default 5 x 10^8 B: long tails at scale -- the decoder's host
tail pins 16 GB -- and rounds of 4 x 10^7 nodes)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bce_amd  # noqa: E402

from scan_time import load  # noqa: E402

extra = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000_000
d = np.concatenate([load("natural"), load("binary"), bce_amd.synth_text(9, extra)])      # long tails AND wide rounds at 5 x 10^8 B
n = len(d)
ctx = bce_amd.api._Ctx(0)
t0 = time.time()
rf = bce_amd.RankFile(d, ctx=ctx)
arch = bce_amd.BCE().encode(rf)
st = bce_amd.stats(rf)
print("compress %d B -> %d B in %.2f s (K1 %.0f ms, K3 %.0f ms, %d rounds)" % (n, len(arch), time.time() - t0, st["t_bwt"] * 1e3, st["k3_ms"], st["rounds"]), flush=True)
buf = np.zeros(n + 64, dtype=np.uint8)
for i in range(2):
    t0 = time.time()
    got = bce_amd.decompress_device(arch, ctx=ctx, out=buf)
    dt = time.time() - t0
    print("decode %.2f s  %.1f MB/s  %s" % (dt, n / dt / 1e6, "identical" if got == n and np.array_equal(buf[:n], d) else "DIFFERENT"), flush=True)
ctx.close()
