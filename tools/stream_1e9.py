#!/usr/bin/env python3
"""A stream of enwik9-sized inputs (synth-text v1 seed 1, 10^9 B) through gated contexts on one GPU: MB/s, every archive against
the oracle's hash.  A context holds ~85 GB after such a compression; side by side they fit because a context that runs out of
device memory gives the buffers of its idle stages back (ctx_trim, api.hip).   python tools/stream_1e9.py [contexts] [inputs]"""
import hashlib
import json
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bce_amd  # noqa: E402

nctx = int(sys.argv[1]) if len(sys.argv) > 1 else 3
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = 1_000_000_000
gold = {v["name"]: v for v in json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_fullsize.json")))["vectors"]}["synth-text-1e9"]
data = bce_amd.synth_text(1, n)
t = torch.from_numpy(data).to("cuda:0")
torch.cuda.synchronize()
with bce_amd.ContextPool(nctx, 0) as pool:
    t0 = time.perf_counter()
    res = pool.compress_many([(t.data_ptr(), n)] * nctx, on_device=True)          # warm-up: one input per context
    print("warm-up: %d inputs in %.2f s" % (nctx, time.perf_counter() - t0), flush=True)
    t0 = time.perf_counter()
    res = pool.compress_many([(t.data_ptr(), n)] * reps, on_device=True, with_stats=True)
    dt = time.perf_counter() - t0
    ok = all(hashlib.sha256(a).hexdigest() == gold["archive_sha256"] for a, _ in res)
    free_b, total_b = torch.cuda.mem_get_info()
    print(json.dumps({"contexts": nctx, "inputs": reps, "seconds": round(dt, 3), "value": round(n * reps / dt / 1e6, 1), "unit": "MB/s",
                      "ms_per_input": round(dt / reps * 1e3, 1), "oracle_golden": "identical" if ok else "DIFFERENT",
                      "device_memory_in_use_GB": round((total_b - free_b) / 1e9, 1)}), flush=True)
