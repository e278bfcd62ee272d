#!/usr/bin/env python3
"""K1 (rotation sort + BWT) in several contexts at once on one GPU, against the same input sorted alone.
   tools/k1_concurrent.py FILE [threads] [repeats] [bytes]"""
import ctypes as C
import hashlib
import sys
import threading

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bce_amd  # noqa: E402

path = sys.argv[1]
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 3
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
nbytes = int(sys.argv[4]) if len(sys.argv) > 4 else 0
data = np.fromfile(path, dtype=np.uint8)
if nbytes:
    data = data[:nbytes]


def bwt_of(ctx):
    rf = bce_amd.RankFile(data, ctx=ctx, build=False)
    return hashlib.sha256(rf.bwt()).hexdigest(), rf.offset()


c0 = bce_amd.api._Ctx(0)
ref = bwt_of(c0)
c0.close()
print("alone:", ref)
bad = []


ctxs = [bce_amd.api._Ctx(0) for _ in range(nthreads)]
if len(sys.argv) > 5 and sys.argv[5] == "warm":
    for c in ctxs:
        print("warm:", bwt_of(c) == ref)


def run(i):
    ctx = ctxs[i]
    try:
        for r in range(reps):
            try:
                got = bwt_of(ctx)
            except Exception as e:  # noqa: BLE001
                got = ("error", str(e))
            if got != ref:
                bad.append((i, r, got))
    finally:
        ctx.close()


ts = [threading.Thread(target=run, args=(i,)) for i in range(nthreads)]
for t in ts:
    t.start()
for t in ts:
    t.join()
print("threads %d x %d: %d bad" % (nthreads, reps, len(bad)))
for b in bad[:10]:
    print("  ", b)
sys.exit(1 if bad else 0)
