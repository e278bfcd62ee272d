#!/usr/bin/env python3
"""Warm in-process latency of compress / decompress_device / scan by input size (one context, best of 5)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bce_amd  # noqa: E402
import oracle   # noqa: E402

ctx = bce_amd.api._Ctx(0)
for n in (1000, 16384, 65536, 262144, 1 << 20, 4 << 20, 16 << 20):
    d = bce_amd.synth_text(3, n)
    best = {}
    arch = None
    for _ in range(5):
        t0 = time.perf_counter(); rf = bce_amd.RankFile(d, ctx=ctx); arch = bce_amd.BCE().encode(rf); t1 = time.perf_counter()
        back = bce_amd.decompress_device(arch, ctx=ctx); t2 = time.perf_counter()
        best["c"] = min(best.get("c", 9), t1 - t0); best["d"] = min(best.get("d", 9), t2 - t1)
        assert back == d.tobytes()
    t0 = time.perf_counter(); ref = oracle.compress(d.tobytes()); to = time.perf_counter() - t0
    print("%9d B: compress %7.2f ms (%6.1f MB/s)  decode %7.2f ms   CPU oracle %8.1f ms   identical %s" % (n, best["c"] * 1e3, n / best["c"] / 1e6, best["d"] * 1e3, to * 1e3, ref == arch), flush=True)
