#!/usr/bin/env python3
"""K1's scatters by destination range (k1_scatter32_kernel, round 4) against the direct ones: K1 seconds at 10^8 .. 10^9 B of
synth-text with BCE_K1_PART_MIN (elements from which the two-step form is used) = the default, 0 (always) and 2^31 (never), and
the archive check against the oracle's hashes.   python tools/k1_part_time.py [sizes...]"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
gold = {v["input_sha256"]: v for v in json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_fullsize.json")))["vectors"]}
child = r'''
import sys, time, hashlib, json
sys.path.insert(0, %r)
import numpy as np, torch, bce_amd
n = int(sys.argv[1])
d = bce_amd.synth_text(1, n)
t = torch.from_numpy(d).to("cuda:0"); torch.cuda.synchronize()
ctx = bce_amd.api._Ctx(0)
best = None
for it in range(3):
    arch, st = bce_amd.compress_device(t.data_ptr(), n, ctx=ctx)
    if it and (best is None or st["t_bwt"] < best): best = st["t_bwt"]
print(json.dumps({"k1_ms": best * 1e3, "sha": hashlib.sha256(arch).hexdigest(), "in": hashlib.sha256(d.tobytes()).hexdigest(), "rounds": st["sort_rounds"]}))
''' % ROOT
sizes = [int(float(x)) for x in sys.argv[1:]] or [100_000_000, 250_000_000, 1_000_000_000]
for n in sizes:
    line = "%11d B:" % n
    for label, env in (("default", None), ("always", "0"), ("never", str(1 << 31))):
        e = dict(os.environ)
        if env is not None:
            e["BCE_K1_PART_MIN"] = env
        r = subprocess.run([sys.executable, "-c", child, str(n)], capture_output=True, text=True, env=e)
        try:
            j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        except Exception:
            print(r.stdout[-500:], r.stderr[-1500:]); raise
        g = gold.get(j["in"])
        line += "  %s %.1f ms (%s)" % (label, j["k1_ms"], "== oracle" if g and g["archive_sha256"] == j["sha"] else ("no golden" if not g else "DIFFERENT"))
    print(line, flush=True)
