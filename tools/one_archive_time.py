#!/usr/bin/env python3
"""One archive from N contexts (bce_hip_set_plane_mask): what a context that codes only some planes still does.  Since round 4
the enumeration's rounds record no symbol for a plane another context codes, so the model (K4), its sort and the device-to-host
copy shrink with the mask.  Two contexts on ONE GPU, masks 0x0F | 0xF0 (and the four-way split), 10^8 B of synth-text: per
context symbols, t_model (K4 kernels + copies, GPU seconds), busiest coder, and the joined archive against the full one."""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bce_amd  # noqa: E402
from bce_amd import api  # noqa: E402
import ctypes as C  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
    data = bce_amd.synth_text(1, n)
    t = torch.from_numpy(data).to("cuda:0")
    torch.cuda.synchronize()
    full_ctx = api._Ctx(0)
    for _ in range(2):
        full, st = bce_amd.compress_device(t.data_ptr(), n, ctx=full_ctx)
    print("all planes (0xFF):    symbols %11d  t_model %6.1f ms  coder busiest %6.1f ms  step %6.1f ms" % (
        st["symbols"], st["t_model"] * 1e3, st["t_coder_busy"] * 1e3, st["t_total"] * 1e3))
    full = bytes(full)
    for masks in ((0x0F, 0xF0), (0x11, 0x22, 0x44, 0x88)):
        ctxs = [api._Ctx(0) for _ in masks]
        stats = []
        for c, m in zip(ctxs, masks):
            api.set_plane_mask(c, m)
            for _ in range(2):
                rf = api.RankFile(n=n, device_ptr=t.data_ptr(), ctx=c)
                api.BCE(None).encode(rf)
            stats.append(api.stats_of(c))
        # join: context 0 receives the other contexts' plane streams
        for c, m in list(zip(ctxs, masks))[1:]:
            for p in range(8):
                if (m >> p) & 1:
                    s = api.plane_stream(c, p)
                    ctxs[0].check(ctxs[0].lib.bce_hip_plane_stream_set(ctxs[0].h, p, s.ctypes.data, len(s)), "plane_stream_set")
        joined = bytes(api.archive_of(ctxs[0]))
        for m, s in zip(masks, stats):
            print("mask 0x%02X:            symbols %11d  t_model %6.1f ms  coder busiest %6.1f ms  step %6.1f ms" % (
                m, s["symbols"], s["t_model"] * 1e3, s["t_coder_busy"] * 1e3, s["t_total"] * 1e3))
        print("  joined archive %s the one of a single context (%d B, sha256 %s)" % ("==" if joined == full else "DIFFERS FROM", len(joined), hashlib.sha256(joined).hexdigest()[:16]))
        for c in ctxs:
            c.close()


if __name__ == "__main__":
    main()
