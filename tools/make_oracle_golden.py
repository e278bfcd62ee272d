#!/usr/bin/env python3
"""Full-size known answers from the CPU ORACLE (oracle/bce_oracle.c, single thread), written to
tests/golden/oracle_fullsize.json.  The GPU tests and bench.py then pin archives at BASELINE.json's sizes against
hashes the oracle produced -- never against the GPU path's own output.

Workloads (all regenerable on any box with this image; inputs are not stored, their sha256 is):
  synth-text / synth-rand  SURVEY 8c generators (xorshift64*), seed 1
  natural                  tools/make_corpus.py          (the image's Python sources + ROCm headers)
  binary                   tools/make_binary_corpus.py   (the image's shared libraries)

    python tools/make_oracle_golden.py                 # everything (several minutes of CPU, ~2 GB of memory per job)
    python tools/make_oracle_golden.py --only synth-text-1e8 natural-1e8
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "oracle_fullsize.json")

JOBS = [
    # name, kind, n
    ("synth-text-1e8", "synth_text", 100_000_000),          # BASELINE configs[1] stand-in (enwik8-sized)
    ("synth-text-1.5e8", "synth_text", 150_000_000),        # n > 2^27: get_context's uint32 wrap (quirk Q1)
    ("synth-rand-32Mi", "synth_rand", 32 << 20),            # 6 symbols per byte
    ("natural-1e8", "natural", 100_000_000),
    ("binary-1e8", "binary", 100_000_000),
    ("natural-16Mi", "natural", 16 << 20),
    ("binary-16Mi", "binary", 16 << 20),
    # BASELINE configs[2] stand-in (enwik9-sized): the only size at which bits = 3, 4 of get_context wrap (bce.cpp:674,
    # c1 >= 2^29 / 2^28) and getv's 31-bit path (:374) matters.  ~8 min, ~13 GB.
    ("synth-text-1e9", "synth_text", 1_000_000_000),
    # BASELINE configs[4] stand-in (Silesia-sized, mixed): natural corpus || binary corpus, compressed with the table
    # `bce -s` (oracle.scan) finds for it.  The 288-byte table itself is stored so that compress parity can be checked
    # apart from scan parity.
    ("mixed-2e8-scanned", "mixed", 200_000_000),
    # The reference's full input range (any 1 <= n < 2^31: saidx_t at bce.cpp:901, getv's 31-bit path :374, Rank :173): the
    # largest even-sized text input, and a high-entropy input whose node lists pass 357 M nodes (the former cap of a list)
    # and whose widest rounds emit more than 2^31 symbols.  ~25 and ~40 minutes of oracle, ~28 / ~22 GB.
    ("synth-text-2p31m2", "synth_text", (1 << 31) - 2),
    ("synth-rand-1.5e9", "synth_rand", 1_500_000_000),
    # (synth-rand at 2^31 - 2 bytes needs more than 62 GB here: the oracle was killed at 65 GB after 37 minutes -- no vector)
]
# BASELINE configs[3] stand-in: ONE input (synth-text v1 seed 1, 10^9 B) cut into N contiguous blocks (sharding.block_range),
# one archive per block as the reference would write it for that block alone (bce.cpp:1151-1157: one block per archive).
# 14 blocks for N = 2, 4, 8; ~0.6 s of oracle per MB, ~13 B of memory per input byte.
BLOCK_WORLDS = (2, 4, 8)
BLOCK_N = 1_000_000_000
for _w in BLOCK_WORLDS:
    for _r in range(_w):
        JOBS.append(("synth-text-1e9-b%dof%d" % (_r, _w), "synth_text_block", BLOCK_N))


def corpus(kind, n, cache="/tmp"):
    path = os.path.join(cache, "bce_%s_%d.bin" % (kind, n))
    if not (os.path.exists(path) and os.path.getsize(path) == n):
        tool = "make_corpus.py" if kind == "natural" else "make_binary_corpus.py"
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", tool), "--out", path, "--size", str(n)])
    return np.fromfile(path, dtype=np.uint8)


_whole = {}


def make_input(kind, n, name=None):
    if kind == "synth_text_block":
        from bce_amd.sharding import block_range
        if n not in _whole:
            _whole[n] = np.frombuffer(oracle.synth_text(1, n), dtype=np.uint8)
        r, w = (int(x) for x in name.rsplit("-b", 1)[1].split("of"))
        lo, hi = block_range(n, w, r)
        return _whole[n][lo:hi]
    if kind in ("synth_text", "synth_rand"):
        return np.frombuffer(getattr(oracle, kind)(1, n), dtype=np.uint8)
    if kind == "mixed":
        return np.concatenate([corpus("natural", n // 2), corpus("binary", n - n // 2)])
    return corpus(kind, n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*")
    ap.add_argument("--out", default=OUT, help="JSON to update (default: tests/golden/oracle_fullsize.json)")
    a = ap.parse_args()
    oracle.build()
    oracle.set_threads(1)
    res = {}
    if os.path.exists(a.out):
        res = {v["name"]: v for v in json.load(open(a.out))["vectors"]}
    for name, kind, n in JOBS:
        if a.only and name not in a.only:
            continue
        data = make_input(kind, n, name)
        block = None
        if kind == "synth_text_block":
            r_, w_ = (int(x) for x in name.rsplit("-b", 1)[1].split("of"))
            block = {"rank": r_, "world": w_, "of_n": n}
            n = len(data)
        assert len(data) == n, (name, len(data))
        extra = {}
        cfg = None
        if kind == "mixed":
            t0 = time.time()
            cfg, sizes = oracle.scan(data)
            extra = {"config_hex": cfg.hex(), "config_sha256": hashlib.sha256(cfg).hexdigest(),
                     "scan_result_sizes": sizes, "oracle_scan_seconds": round(time.time() - t0, 1)}
        t0 = time.time()
        arch = oracle.compress(data, cfg)
        dt = time.time() - t0
        res[name] = {"name": name, "kind": kind, "seed": 1 if kind.startswith("synth") else None, "n": n,
                     "input_sha256": hashlib.sha256(data.tobytes()).hexdigest(),
                     "archive_bytes": len(arch), "archive_sha256": hashlib.sha256(arch).hexdigest(),
                     "oracle_seconds_1_thread": round(dt, 1), **extra}
        if block:
            res[name]["block"] = block
        print(json.dumps(res[name]), flush=True)
        doc = {"provenance": "oracle/bce_oracle.c (CPU restatement of bce -c, single thread) run by tools/make_oracle_golden.py; "
                             "no GPU code involved.  The natural/binary corpora are built from the files of this container image, so "
                             "their input_sha256 only matches on a box with the same image (tests skip otherwise).",
               "vectors": [res[k] for k in sorted(res)]}
        if os.path.exists(a.out):       # another job may have added vectors meanwhile
            for v in json.load(open(a.out))["vectors"]:
                res.setdefault(v["name"], v)
            doc["vectors"] = [res[k] for k in sorted(res)]
        with open(a.out, "w") as f:
            json.dump(doc, f, indent=1)
            f.write("\n")


if __name__ == "__main__":
    main()
