#!/usr/bin/env python3
"""Time `bce -s` (bce_amd.scan) on the 10^8-byte workloads; BCE_HIP_SCAN_DEBUG=1 prints the stage split to stderr.
   python tools/scan_time.py [text|natural|binary ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bce_amd  # noqa: E402


def load(kind, n=100_000_000):
    if kind == "text":
        return bce_amd.synth_text(1, n)
    path = "/tmp/bce_%s_%d.bin" % (kind, n)
    if not os.path.exists(path):
        import subprocess
        tool = "make_corpus.py" if kind == "natural" else "make_binary_corpus.py"
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", tool), "--out", path, "--size", str(n)],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return np.fromfile(path, dtype=np.uint8)


def main():
    kinds = sys.argv[1:] or ["text", "natural", "binary"]
    bce_amd.scan(bce_amd.synth_text(1, 1 << 20))        # HIP init
    for k in kinds:
        d = load(k)
        best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            bce_amd.scan(d)
            best = min(best, time.perf_counter() - t)
        # where the time outside bce_hip_scan goes: context + K1/K2 (RankFile), the call, giving the context back
        import ctypes as C
        from bce_amd import api
        t0 = time.perf_counter()
        rf = api.RankFile(d)
        t1 = time.perf_counter()
        cfg = np.zeros(288, dtype=np.uint8)
        res = (C.c_double * 9)()
        rf._c.check(rf._c.lib.bce_hip_scan(rf._c.h, cfg.ctypes.data, res), "bce_hip_scan")
        t2 = time.perf_counter()
        rf.close()
        t3 = time.perf_counter()
        print("%-8s context + load + K1 + K2 %.3f s, bce_hip_scan %.3f s, close %.3f s" % (k, t1 - t0, t2 - t1, t3 - t2), flush=True)
        print("%-8s scan of %d bytes: best of 3 %.2f s" % (k, len(d), best), flush=True)


if __name__ == "__main__":
    main()
