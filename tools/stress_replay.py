#!/usr/bin/env python3
"""Replay cases LO+1..HI of `tools/stress.py --seed SEED` one by one, printing each case before it runs (to find a
case that hangs or crawls) and its encode / decode times.
    python tools/stress_replay.py SEED LO HI"""
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bce_amd   # noqa: E402
import oracle    # noqa: E402

src = open(os.path.join(ROOT, "tools", "stress.py")).read()
mod = types.ModuleType("stress_gen")
mod.__dict__["__file__"] = os.path.join(ROOT, "tools", "stress.py")
mod.__dict__["__name__"] = "stress_gen"
exec(compile(src, "stress.py", "exec"), mod.__dict__)


def main():
    seed, lo, hi = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    rs = np.random.RandomState(seed)
    sizes = [1, 2, 7, 100, 1000, 5000, 30000, 100000, 400000, 1500000, 6000000]
    ctx = bce_amd.api._Ctx(0)
    cases = 0
    while cases < hi:
        n = int(rs.choice(sizes)) + int(rs.randint(0, 97))
        data = mod.gen(rs, n)
        if len(data) == 0:
            continue
        raw = data.tobytes()
        knobs = {}
        if n < 500000 and rs.randint(0, 3) == 0:
            for k in (1, 2, 3, 4, 6):
                if rs.randint(0, 2):
                    knobs[k] = 1
            if rs.randint(0, 3) == 0:
                knobs[0] = int(rs.choice([3, 40, 700]))
        cap = int(rs.choice([0, 0, 0, 700, 20000]))
        cases += 1
        if cases <= lo:
            continue
        print("case %d n=%d knobs=%r cap=%d ..." % (cases, len(raw), knobs, cap), end=" ", flush=True)
        for k in (0, 1, 2, 3, 4, 6):
            ctx.check(ctx.lib.bce_hip_debug_set(ctx.h, k, knobs.get(k, 0)), "bce_hip_debug_set")
        t0 = time.time()
        rf = bce_amd.RankFile(raw, ctx=ctx)
        arch = bce_amd.BCE(symbol_capacity=cap).encode(rf)
        t1 = time.time()
        print("encode %.2f s" % (t1 - t0), end=" ", flush=True)
        ref = oracle.compress(raw)
        t2 = time.time()
        back = bce_amd.decompress_device(ref, ctx=ctx)
        print("oracle %.2f s decode %.2f s %s" % (t2 - t1, time.time() - t2, "ok" if (arch == ref and back == raw) else "FAIL"), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
