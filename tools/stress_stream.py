#!/usr/bin/env python3
"""Randomised soak test of streams through gated contexts (bce_amd.ContextPool): batches of inputs of varied structure and
size, 1-4 contexts, now and then a tiny symbol buffer (forces the hand-over of the device gate); every archive against
the oracle.      python tools/stress_stream.py --seconds 240 --seed 1"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bce_amd   # noqa: E402
import oracle    # noqa: E402
from stress import gen   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rs = np.random.RandomState(a.seed)
    t_end = time.time() + a.seconds
    batches = cases = fails = 0
    sizes = [1, 7, 100, 5000, 30000, 100000, 400000, 1500000, 4000000]
    pools = {k: bce_amd.ContextPool(k, 0) for k in (1, 2, 3, 4)}
    while time.time() < t_end:
        k = int(rs.choice([1, 2, 2, 3, 3, 4]))
        ins = []
        for _ in range(int(rs.randint(1, 9))):
            d = gen(rs, int(rs.choice(sizes)) + int(rs.randint(0, 97)))
            if len(d):
                ins.append(d.tobytes())
        if not ins:
            continue
        cap = int(rs.choice([0, 0, 0, 900, 30000]))
        got = pools[k].compress_many(ins, symbol_capacity=cap)
        batches += 1
        for raw, arch in zip(ins, got):
            cases += 1
            if bytes(arch) != oracle.compress(raw):
                fails += 1
                name = "/tmp/stress_stream_fail_%d_%d.bin" % (a.seed, cases)
                open(name, "wb").write(raw)
                print("FAIL case %d n=%d contexts=%d cap=%d saved %s" % (cases, len(raw), k, cap, name), flush=True)
        if batches % 50 == 0:
            print("%d batches, %d inputs, %d failures, %.0f s left" % (batches, cases, fails, t_end - time.time()), flush=True)
    for p in pools.values():
        p.close()
    print("stress_stream: %d batches, %d inputs, %d failures (seed %d)" % (batches, cases, fails, a.seed))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
