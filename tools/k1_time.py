#!/usr/bin/env python3
"""K1 (rotation sort + BWT) time on the three 10^8-byte workloads, and the archive check against the oracle's hashes.
    python tools/k1_time.py [--trace]      (--trace: BCE_K1_TRACE=1 for one extra compression per workload: one line per round)
    BCE_K1_V1=1 python tools/k1_time.py    (the round-2 sorter, for A/B)"""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np   # noqa: E402
import torch         # noqa: E402
import bce_amd       # noqa: E402

gold = {v["name"]: v for v in json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_fullsize.json")))["vectors"]}
n = 100_000_000
ctx = bce_amd.api._Ctx(0)
for name, kind in (("synth-text-1e8", "synth_text"), ("natural-1e8", "natural"), ("binary-1e8", "binary")):
    if kind == "synth_text":
        d = bce_amd.synth_text(1, n)
    else:
        path = "/tmp/bce_%s_%d.bin" % (kind, n)
        if not os.path.exists(path):
            tool = "make_corpus.py" if kind == "natural" else "make_binary_corpus.py"
            subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), "--out", path, "--size", str(n)], check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        d = np.fromfile(path, dtype=np.uint8)
    t = torch.from_numpy(d).to("cuda:0")
    torch.cuda.synchronize()
    best = None
    for it in range(4):
        t0 = time.time()
        arch, st = bce_amd.compress_device(t.data_ptr(), len(d), ctx=ctx)
        dt = time.time() - t0
        if it and (best is None or st["t_bwt"] < best[0]):
            best = (st["t_bwt"], dt, st["sort_rounds"], st["k3_ms"])
    ok = hashlib.sha256(arch).hexdigest() == gold[name]["archive_sha256"]
    print("%-16s k1 %.2f ms  step %.1f ms  sort_rounds %d  k3 %.2f ms  archive %s" % (name, best[0] * 1e3, best[1] * 1e3, best[2], best[3],
                                                                              "== oracle" if ok else "DIFFERENT"), flush=True)
    if "--trace" in sys.argv:
        os.environ["BCE_K1_TRACE"] = "1"
        bce_amd.compress_device(t.data_ptr(), len(d), ctx=ctx)
        del os.environ["BCE_K1_TRACE"]
    del t
ctx.close()
