#!/usr/bin/env python3
"""Resident memory of the process over repeated compress / decode / scan / one-archive calls in one context."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bce_amd  # noqa: E402


def rss_mb():
    with open("/proc/self/status") as f:
        for line in f:
            if line.startswith("VmRSS"):
                return int(line.split()[1]) / 1024.0
    return 0.0


ctx = bce_amd.api._Ctx(0)
t = np.frombuffer(bytes(bce_amd.synth_text(5, 12 << 20)), dtype=np.uint8)
data = np.concatenate([t, t[1000000:1400000], np.zeros(200000, dtype=np.uint8), t[:333333]])     # wide rounds and a long tail
buf = np.zeros(len(data) + 64, dtype=np.uint8)
for it in range(25):
    arch = bce_amd.BCE().encode(bce_amd.RankFile(data, ctx=ctx))
    assert bce_amd.decompress_device(arch, ctx=ctx, out=buf) == len(data) and np.array_equal(buf[:len(data)], data)
    bce_amd.set_plane_mask(ctx, 0x0F)
    bce_amd.BCE().encode(bce_amd.RankFile(data, ctx=ctx))
    bce_amd.set_plane_mask(ctx, 0xFF)
    if it % 4 == 0:
        bce_amd.scan(data[:4 << 20])
    if it in (0, 1, 2, 4, 8, 16, 24):
        print("iteration %2d: RSS %.0f MB" % (it, rss_mb()), flush=True)
ctx.close()
print("after close: RSS %.0f MB" % rss_mb())
