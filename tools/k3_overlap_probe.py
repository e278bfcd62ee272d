#!/usr/bin/env python3
"""What would two half-width enumeration pipelines on two streams buy (the one K3 design not built)?
An upper-bound estimate without building it: TWO ungated contexts enumerate the two HALVES of an input at the same time (each
half has all eight tries but half the nodes per round -- the width a pipeline of four tries would have), three launches per
round (they wait for nothing: two grids that spin on their own tiles must not share the device), against ONE context on
the whole input with the same knobs.  If the two overlapped halves take clearly less GPU time than the whole, the rounds'
fixed latency (prologue, scan chain, epilogue: ~45 us of a ~100 us round) can be hidden behind another pipeline's tiles.
    python tools/k3_overlap_probe.py [text|natural|binary]"""
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["BCE_HIP_NO_FUSED"] = "1"
import bce_amd  # noqa: E402
from scan_time import load  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "text"
data = load(kind)
n = len(data)
t_whole = torch.from_numpy(np.ascontiguousarray(data)).to("cuda:0")
halves = [torch.from_numpy(np.ascontiguousarray(data[:n // 2])).to("cuda:0"), torch.from_numpy(np.ascontiguousarray(data[n // 2:])).to("cuda:0")]
torch.cuda.synchronize()


def run(ctx, t, reps, out, key):
    for _ in range(reps):
        _, st = bce_amd.compress_device(t.data_ptr(), t.numel(), ctx=ctx)
        out.setdefault(key, []).append((st["k3_ms"], st["t_total"] * 1e3))


ctxs = [bce_amd.api._Ctx(0) for _ in range(3)]
for c in ctxs:
    c.check(c.lib.bce_hip_set_gated(c.h, 0), "set_gated")
    c.check(c.lib.bce_hip_debug_set(c.h, 4, 1), "debug_set")       # no one-launch rounds (they spin on tiles of their own grid)
res = {}
run(ctxs[0], t_whole, 3, res, "whole")
run(ctxs[1], halves[0], 2, res, "half0_alone")
run(ctxs[2], halves[1], 2, res, "half1_alone")
for rep in range(3):
    th = [threading.Thread(target=run, args=(ctxs[1 + i], halves[i], 1, res, "half%d_together" % i)) for i in range(2)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    res.setdefault("together_wall_ms", []).append((time.perf_counter() - t0) * 1e3)
for k, v in res.items():
    print(k, [tuple(round(x, 2) for x in e) if isinstance(e, tuple) else round(e, 1) for e in v])
w = min(e[0] for e in res["whole"])
a = min(e[0] for e in res["half0_alone"]) + min(e[0] for e in res["half1_alone"])
tg = min(max(x[0], y[0]) for x, y in zip(res["half0_together"], res["half1_together"]))
print("K3 ms: whole %.2f; the halves one after the other %.2f; the halves at the same time (the longer of the two) %.2f" % (w, a, tg))
for c in ctxs:
    c.close()
