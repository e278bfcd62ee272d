#!/usr/bin/env python3
"""Stream of identical inputs through N gated contexts; checks every archive and that the input tensor is intact.
   tools/stream_check.py FILE contexts steps [keep] [pre] [pin] [stats] [seqload] [seqwarm]
   keep: a used context stays alive beside the pool; pre: the pool is warmed on synth-text first; seqload / seqwarm: every
   context runs K1 + K2 / a whole compression alone before the stream starts.  (This is the tool that cornered the K1
   active-list bug: run with GPU_MAX_HW_QUEUES=8 it failed on 10^8 bytes of source code until k1_plan's scalars moved.)"""
import hashlib
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bce_amd  # noqa: E402

data = np.fromfile(sys.argv[1], dtype=np.uint8)
nctx, steps = int(sys.argv[2]), int(sys.argv[3])
n = len(data)
t_in = torch.from_numpy(data).to("cuda:0")
opts = sys.argv[4:]
if "pin" in opts:
    from bce_amd import sharding
    sharding.pin_to_local_numa(0)
keep = bce_amd.api._Ctx(0) if "keep" in opts else None
ref, _ = bce_amd.compress_device(t_in.data_ptr(), n, ctx=keep)
ref = hashlib.sha256(ref).hexdigest()
print("alone:", ref[:16])
with bce_amd.ContextPool(nctx, 0) as pool:
    if "pre" in opts:       # warm the pool on another input of the same size first
        other = torch.from_numpy(bce_amd.synth_text(1, n)).to("cuda:0")
        pool.compress_many([(other.data_ptr(), n)] * (2 * nctx), on_device=True)
        print("pool warmed on synth-text")
    if "seqload" in opts:
        for c in pool.ctxs:
            rf = bce_amd.RankFile(n=n, device_ptr=t_in.data_ptr(), ctx=c)
            print("sequential load + K1 + K2 only, offset", rf.offset())
    if "seqwarm" in opts:
        for c in pool.ctxs:
            a, _ = bce_amd.compress_device(t_in.data_ptr(), n, ctx=c)
            print("sequential warm-up:", hashlib.sha256(a).hexdigest()[:16] == ref[:16])
    for phase in ("warm", "run"):
        k = nctx if phase == "warm" else steps
        try:
            res = pool.compress_many([(t_in.data_ptr(), n)] * k, on_device=True, with_stats="stats" in opts)
            print(phase, [hashlib.sha256(a[0] if "stats" in opts else a).hexdigest()[:16] == ref[:16] for a in res])
        except Exception as e:  # noqa: BLE001
            print(phase, "FAILED:", e)
        same = bool((t_in.cpu().numpy() == data).all())
        print(phase, "input tensor intact:", same)
        if not same:
            diff = np.nonzero(t_in.cpu().numpy() != data)[0]
            print("  first/last differing byte", diff[0], diff[-1], "count", len(diff))
            break
