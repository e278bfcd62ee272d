#!/usr/bin/env python3
"""Randomised soak test on the GPU box: many inputs of varied structure and size through compress (vs the oracle),
the GPU-assisted decoder and -- for the smaller ones -- every alternative enumeration path.
    python tools/stress.py --seconds 300 --seed 1"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bce_amd   # noqa: E402
import oracle    # noqa: E402


def gen(rs, n):
    kind = rs.randint(0, 12)
    if kind in (2, 3, 8) and n > 60000:      # long exact repeats: the CPU oracle walks them bit by bit (minutes at MBs)
        n = 20000 + n % 40000
    if kind == 0:
        return rs.randint(0, rs.randint(1, 6) + 1, n).astype(np.uint8) + 65
    if kind == 1:
        vals = rs.randint(0, 256, max(1, n // 40 + 1)).astype(np.uint8)
        return np.repeat(vals, rs.randint(1, 90, len(vals)))[:n]
    if kind == 2:
        p = rs.randint(1, 40)
        return np.tile(rs.randint(97, 103, p).astype(np.uint8), n // p + 1)[:n]
    if kind == 3:
        p = rs.randint(2, 3000)
        a = np.tile(rs.randint(0, 256, p).astype(np.uint8), n // p + 1)[:n].copy()
        for _ in range(rs.randint(1, 6)):
            a[rs.randint(0, n)] ^= 1 << rs.randint(0, 8)
        return a
    if kind in (4, 5):
        base = np.frombuffer(oracle.synth_text(int(rs.randint(1, 10 ** 6)), n), dtype=np.uint8).copy()
        for _ in range(rs.randint(0, 5)):
            if n > 400:
                L = rs.randint(10, min(n // 3, 30000))
                src, dst = rs.randint(0, n - L), rs.randint(0, n - L)
                base[dst:dst + L] = base[src:src + L].copy()
        return base
    if kind == 6:
        return rs.randint(0, 256, n).astype(np.uint8)
    if kind == 7:
        return (rs.randint(0, 1 << rs.randint(1, 5), n).astype(np.uint8) << rs.randint(0, 5)).astype(np.uint8)
    if kind in (10, 11):   # spines (k3_dfs.hip spine_burst): a context that keeps thousands of rows and loses a few per byte
        parts, size = [], 0
        while size < n:
            if kind == 10:     # runs of one byte of many lengths between random bytes
                p = rs.bytes(int(rs.randint(1, 24))) + bytes([int(rs.choice([0, 0, 255, 32]))]) * int(rs.randint(1, 2500))
            else:              # little-endian 16-bit tables with small values, records with constant fields
                if rs.randint(0, 2):
                    p = np.minimum(rs.geometric(0.2, int(rs.randint(50, 3000))), 200).astype("<u2").tobytes() + rs.bytes(2)
                else:
                    p = b"".join(rs.bytes(3) + b"\x00\x00\x00\x01\x00\x00\x00\xff\x10" for _ in range(int(rs.randint(20, 800)))) + rs.bytes(int(rs.randint(1, 9)))
            parts.append(p)
            size += len(p)
        return np.frombuffer(b"".join(parts), dtype=np.uint8)[:n].copy()
    if kind == 9:          # long runs of one byte inside other data (executables): chains that cannot be skipped
        base = rs.randint(0, 256, n).astype(np.uint8) if rs.randint(0, 2) else np.frombuffer(oracle.synth_text(int(rs.randint(1, 10 ** 6)), n), dtype=np.uint8).copy()
        unit0 = rs.randint(0, 4, rs.randint(1, 9)).astype(np.uint8) * rs.choice([1, 1, 85])    # the same pattern in several places
        for _ in range(rs.randint(1, 12)):
            if n > 10:
                L = rs.randint(2, max(3, min(n // 2, 60000)))
                at = rs.randint(0, n - L)
                if rs.randint(0, 3) == 0:
                    base[at:at + L] = np.resize(np.roll(unit0, rs.randint(0, len(unit0))), L)
                elif rs.randint(0, 2):
                    base[at:at + L] = rs.choice([0, 0, 255, 32, rs.randint(0, 256)])
                else:                                           # a table of equal records (period 2..24)
                    unit = rs.randint(0, 4, rs.randint(2, 25)).astype(np.uint8) * rs.choice([1, 1, 85])
                    base[at:at + L] = np.resize(unit, L)
        return base
    blk = np.frombuffer(oracle.synth_text(int(rs.randint(1, 10 ** 6)), max(50, n // rs.randint(3, 200))), dtype=np.uint8)
    return np.tile(blk, n // len(blk) + 1)[:n].copy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rs = np.random.RandomState(a.seed)
    ctx = bce_amd.api._Ctx(0)
    t_end = time.time() + a.seconds
    cases = fails = 0
    sizes = [1, 2, 7, 100, 1000, 5000, 30000, 100000, 400000, 1500000, 6000000]
    while time.time() < t_end:
        n = int(rs.choice(sizes)) + int(rs.randint(0, 97))
        data = gen(rs, n)
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
        # round 2: the workgroup-local rounds (8 = from how many live nodes, 9 = rounds per pass, 10 = tail entered while the
        # node count still grows: spills), 7 = without them, 11 = model flushes beside the rounds
        if rs.randint(0, 2) == 0:
            knobs[8] = int(rs.choice([16, 64, 300, 5000]))
            if rs.randint(0, 2):
                knobs[9] = int(rs.choice([1, 3, 17, 100]))
            if rs.randint(0, 3) == 0:
                knobs[10] = int(rs.choice([17, 18, 20, 23]))
        elif rs.randint(0, 4) == 0:
            knobs[7] = 1
        if rs.randint(0, 4) == 0:
            knobs[11] = 1
        for k in (0, 1, 2, 3, 4, 6, 7, 8, 9, 10, 11):
            ctx.check(ctx.lib.bce_hip_debug_set(ctx.h, k, knobs.get(k, 0)), "bce_hip_debug_set")
        rf = bce_amd.RankFile(raw, ctx=ctx)
        cap = int(rs.choice([0, 0, 0, 700, 20000]))
        arch = bce_amd.BCE(symbol_capacity=cap).encode(rf)
        ref = oracle.compress(raw)
        # round 3: the decoder's routes (environment variables, read at every decode): six launches per round everywhere
        # (which makes every round one of eight lanes, plane by plane) or all planes at once, where the host takes the tail
        denv = {}
        if rs.randint(0, 2) == 0:
            if rs.randint(0, 2):
                denv["BCE_DEC_NO_SMALL"] = "1"
            if rs.randint(0, 3) == 0:
                denv["BCE_DEC_NO_SPLIT"] = "1"
            if rs.randint(0, 2) == 0:
                denv["BCE_DEC_HOST_ENTER"] = str(int(rs.choice([0, 64, 3000, 100000])))
            if rs.randint(0, 4) == 0:
                denv["BCE_DEC_FORCE_HOST_TAIL"] = "1"
            if rs.randint(0, 4) == 0:
                denv["BCE_DEC_NO_EARLY_PIN"] = "1"
        os.environ.update(denv)
        try:
            ok = arch == ref and bce_amd.decompress_device(ref, ctx=ctx) == raw
        finally:
            for k in denv:
                del os.environ[k]
        cases += 1
        if not ok:
            fails += 1
            name = "/tmp/stress_fail_%d_%d.bin" % (a.seed, cases)
            open(name, "wb").write(raw)
            print("FAIL case %d n=%d knobs=%r decoder env=%r cap=%d saved %s" % (cases, len(raw), knobs, denv, cap, name), flush=True)
        if cases % 200 == 0:
            print("%d cases, %d failures, %.0f s left" % (cases, fails, t_end - time.time()), flush=True)
    ctx.close()
    print("stress: %d cases, %d failures (seed %d)" % (cases, fails, a.seed))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
