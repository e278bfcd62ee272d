#!/usr/bin/env python3
"""The inputs that failed before round 5 (include/bce_hip.h, capacity note): compress on the GPU, print the statistics that show
which mechanisms ran (list_nodes, list_grows, split_rounds), decode with the GPU-assisted decoder and compare with the input.
    python tools/bigrange_check.py rand 1500000000
    python tools/bigrange_check.py text 2147483646
Known answers (the oracle's archive hash) are in tests/golden/oracle_fullsize.json when tools/make_oracle_golden.py has run."""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bce_amd  # noqa: E402

kind, n = sys.argv[1], int(sys.argv[2])
decode = len(sys.argv) < 4 or sys.argv[3] != "nodecode"
t0 = time.time()
data = getattr(bce_amd, "synth_" + kind)(1, n)
print("input: synth-%s v1 seed 1, %d B, sha256 %s (%.1f s)" % (kind, n, hashlib.sha256(data).hexdigest(), time.time() - t0), flush=True)
ctx = bce_amd.api._Ctx(0)
t0 = time.time()
rf = bce_amd.RankFile(data, ctx=ctx)
print("K1 + K2: %.1f s" % (time.time() - t0), flush=True)
t1 = time.time()
arch = bce_amd.BCE().encode(rf)
st = bce_amd.stats(rf)
print("encode: %.1f s; archive %d B sha256 %s" % (time.time() - t1, len(arch), hashlib.sha256(arch).hexdigest()), flush=True)
print("stats:", {k: st[k] for k in ("nodes", "symbols", "rounds", "flushes", "list_grows", "list_nodes", "split_rounds", "k3_ms", "t_bwt", "t_model", "t_coder_busy")}, flush=True)
assert st["nodes"] == 8 * n - 8, (st["nodes"], 8 * n - 8)
if decode:
    want = hashlib.sha256(data).hexdigest()
    del data
    out = np.empty(n, dtype=np.uint8)
    t2 = time.time()
    got = bce_amd.decompress_device(arch, ctx=ctx, out=out)
    print("decode: %.1f s, restarts %d" % (time.time() - t2, bce_amd.stats_of(ctx)["dec_restarts"]), flush=True)
    assert got == n and hashlib.sha256(out).hexdigest() == want
    print("round trip ok", flush=True)
ctx.close()
