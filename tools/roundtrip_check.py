#!/usr/bin/env python3
"""Full-size round trip: compress on the GPU, decode (GPU-assisted decoder by default, or the host decoder), compare.
The decoder mirrors the reference's model step by step, so a decodable archive of the right input is the
size-independent evidence that the GPU encoder's model and coder state never diverged.
    python tools/roundtrip_check.py --size 1000000000"""
import argparse
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bce_amd   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1_000_000_000)
    ap.add_argument("--workload", default="synth-text", choices=["synth-text", "synth-rand"])
    ap.add_argument("--decoder", default="gpu", choices=["gpu", "host"], help="gpu = kd_decode.hip, host = decoder.cpp")
    ap.add_argument("--file", default=None)
    a = ap.parse_args()
    gen = bce_amd.synth_text if a.workload == "synth-text" else bce_amd.synth_rand
    data = gen(1, a.size) if not a.file else __import__("numpy").fromfile(a.file, dtype="uint8", count=a.size)
    h_in = hashlib.sha256(data.tobytes()).hexdigest()
    t0 = time.time()
    arch = bce_amd.compress(data)
    t1 = time.time()
    print("compressed %d -> %d B in %.2f s (cold), archive sha256 %s" % (a.size, len(arch), t1 - t0, hashlib.sha256(arch).hexdigest()), flush=True)
    back = bce_amd.decompress_device(arch) if a.decoder == "gpu" else bce_amd.decompress(arch)
    t2 = time.time()
    ok = len(back) == a.size and hashlib.sha256(back).hexdigest() == h_in
    print("decoded (%s decoder) in %.2f s = %.1f MB/s: %s" % (a.decoder, t2 - t1, a.size / (t2 - t1) / 1e6, "IDENTICAL to the input" if ok else "MISMATCH"), flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
