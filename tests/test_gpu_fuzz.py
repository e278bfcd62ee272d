"""GPU: randomized small inputs of varied structure (tiny alphabets, runs, periodic and near-periodic strings,
repeats at several scales) against the oracle, all through one context (buffer reuse across calls included)."""
import numpy as np
import pytest

import bce_amd
import oracle

pytestmark = pytest.mark.gpu


def gen_case(rs):
    kind = rs.randint(0, 8)
    n = int(rs.choice([1, 2, 3, 5, 17, 64, 95, 96, 97, 255, 256, 1000, 2047, 2049, 3072, 5000, 12000, 40000]))
    if kind == 0:      # tiny alphabet
        return rs.randint(0, rs.randint(1, 4) + 1, n).astype(np.uint8) + 97
    if kind == 1:      # runs
        vals = rs.randint(0, 256, max(1, n // 50 + 1)).astype(np.uint8)
        return np.repeat(vals, rs.randint(1, 100, len(vals)))[:n] if n > 1 else vals[:1]
    if kind == 2:      # exactly periodic
        p = rs.randint(1, 9)
        return np.tile(rs.randint(97, 100, p).astype(np.uint8), n // p + 1)[:n]
    if kind == 3:      # periodic with one defect
        p = rs.randint(1, 9)
        a = np.tile(rs.randint(97, 100, p).astype(np.uint8), n // p + 1)[:n].copy()
        a[rs.randint(0, n)] ^= 1
        return a
    if kind == 4:      # text with an internal long repeat
        base = np.frombuffer(oracle.synth_text(int(rs.randint(1, 1000)), n), dtype=np.uint8).copy()
        if n > 200:
            L = rs.randint(10, n // 3)
            src, dst = rs.randint(0, n - L), rs.randint(0, n - L)
            base[dst:dst + L] = base[src:src + L].copy()
        return base
    if kind == 5:      # random bytes
        return rs.randint(0, 256, n).astype(np.uint8)
    if kind == 6:      # high bit planes constant / few distinct bytes
        return (rs.randint(0, 4, n).astype(np.uint8) << rs.randint(0, 7)).astype(np.uint8)
    return np.frombuffer(oracle.synth_text(int(rs.randint(1, 1000)), n), dtype=np.uint8).copy()


def test_fuzz_archives_match_oracle():
    rs = np.random.RandomState(20261003)
    ctx = bce_amd.api._Ctx(0)
    try:
        for case in range(300):
            data = gen_case(rs)
            if len(data) == 0:
                continue
            raw = data.tobytes()
            rf = bce_amd.RankFile(raw, ctx=ctx)
            cap = int(rs.choice([0, 0, 0, 500, 5000]))
            arch = bce_amd.BCE(symbol_capacity=cap).encode(rf)
            ref = oracle.compress(raw)
            assert arch == ref, "case %d: n=%d head=%r" % (case, len(raw), raw[:24])
            assert bce_amd.decompress(arch) == raw, "decode case %d" % case
            # the GPU-assisted decoder through the SAME context the next compression will use (buffers change hands)
            assert bce_amd.decompress_device(arch, ctx=ctx) == raw, "gpu decode case %d" % case
    finally:
        ctx.close()
