#!/usr/bin/env python3
"""Parity on the natural corpus of tools/make_corpus.py: GPU archive == oracle archive on the first --size bytes,
and decode(GPU archive) == input.   python tools/corpus_check.py /tmp/corpus100.bin --size 16777216"""
import argparse
import hashlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bce_amd   # noqa: E402
import oracle    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("file")
    ap.add_argument("--size", type=int, default=16 << 20)
    ap.add_argument("--skip", type=int, default=0)
    a = ap.parse_args()
    data = np.fromfile(a.file, dtype=np.uint8, count=a.size, offset=a.skip).tobytes()
    t0 = time.time(); got = bce_amd.compress(data); t1 = time.time()
    want = oracle.compress(data); t2 = time.time()
    ok = got == want
    back = bce_amd.decompress(got) == data
    print("corpus_check: %d B -> %d B, gpu %.2f s, oracle %.2f s, archive %s, roundtrip %s, sha256 %s" % (
        len(data), len(got), t1 - t0, t2 - t1, "IDENTICAL" if ok else "DIFFERENT", "ok" if back else "FAILED",
        hashlib.sha256(got).hexdigest()[:16]))
    return 0 if ok and back else 1


if __name__ == "__main__":
    sys.exit(main())
