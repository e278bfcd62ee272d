"""GPU: the wide rounds as two half-width pipelines on two streams (api.hip enumerate_wide_dual).

Tries 0-3 and tries 4-7 run their rounds side by side, each with a control block, tile arrays, run table and symbol region of its
own, on node lists indexed by trie; a plane's records alternate between the two regions every four rounds and K4 gathers them
into stream order.  By default only rounds above 2 M nodes take this path (inputs of tens of MB: the full-size tests); debug
knob 13 = 1 sends every wide-path round through it, however narrow, so that small inputs reach it -- against the oracle, with the
paths around it varied (no one-launch rounds, small lists that make it leave and come back, small record buffers that make it
flush often, masked planes), and 13 = 2 / BCE_HIP_NO_DUAL switch it off."""
import numpy as np
import pytest

import bce_amd
import oracle
from conftest import edge_inputs

pytestmark = pytest.mark.gpu


def _encode(data, knobs, cap=0, config=None):
    ctx = bce_amd.api._Ctx(0)
    try:
        for k, v in knobs.items():
            ctx.check(ctx.lib.bce_hip_debug_set(ctx.h, k, v), "bce_hip_debug_set")
        rf = bce_amd.RankFile(data, ctx=ctx)
        arch = bytes(bce_amd.BCE(config, cap).encode(rf))
        return arch, bce_amd.stats(rf)
    finally:
        ctx.close()


CASES = {
    "text": lambda: oracle.synth_text(71, 400000),
    "rand": lambda: oracle.synth_rand(72, 250000),
    "mixed": lambda: oracle.synth_text(73, 150000) + bytes(20000) + oracle.synth_rand(74, 100000) + oracle.synth_text(73, 60000),
    "periodic": lambda: b"abcab" * 30000 + b"x",
    "two-roots-only": lambda: bytes([0, 255]) * 50000 + b"\x0f",
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_forced_dual_pipelines_match_the_oracle(case):
    data = CASES[case]()
    want = oracle.compress(data)
    ran = 0
    for knobs in ({13: 1}, {13: 1, 4: 1}, {13: 1, 4: 1, 1: 1}, {13: 1, 12: 64}, {13: 1, 6: 1}, {13: 2}):
        arch, st = _encode(data, knobs)
        assert arch == want, (case, knobs)
        assert st["nodes"] == 8 * len(data) - 8 or case in ("periodic", "two-roots-only"), (case, knobs, st)
        if knobs.get(13) == 1:
            ran += st["dual_rounds"]
        else:
            assert st["dual_rounds"] == 0
    if case in ("text", "rand", "mixed"):
        assert ran >= 3, (case, ran)                              # (the pipelines did run; the other two inputs never have a wide round)
    assert bce_amd.decompress_device(want) == data


def test_forced_dual_with_small_record_buffers_and_masks():
    data = oracle.synth_text(75, 300000) + oracle.synth_rand(76, 80000)
    cfg = np.random.RandomState(9).randint(0, 6, 288).astype(np.uint8).tobytes()
    want = oracle.compress(data, cfg)
    for cap in (0, 50000, 6000):
        arch, st = _encode(data, {13: 1, 4: 1}, cap=cap, config=cfg)      # (4: no one-launch rounds, or a small input never takes the wide path)
        assert arch == want, cap
        assert st["dual_rounds"] >= 1
    # one archive from two contexts (plane masks), both through the two pipelines
    ctxs = []
    try:
        for mask in (0x35, 0xCA):
            c = bce_amd.api._Ctx(0)
            ctxs.append(c)
            c.check(c.lib.bce_hip_debug_set(c.h, 13, 1), "bce_hip_debug_set")
            c.check(c.lib.bce_hip_debug_set(c.h, 4, 1), "bce_hip_debug_set")
            bce_amd.set_plane_mask(c, mask)
            rf = bce_amd.RankFile(data, ctx=c)
            bce_amd.BCE(cfg).encode(rf)
        for p in range(8):
            if (0xCA >> p) & 1:
                bce_amd.set_plane_stream(ctxs[0], p, bce_amd.plane_stream(ctxs[1], p))
        assert bytes(bce_amd.archive_of(ctxs[0])) == want
    finally:
        for c in ctxs:
            c.close()


def test_forced_dual_on_the_edge_inputs():
    for name, data in edge_inputs():
        arch, _ = _encode(data, {13: 1, 4: 1})
        assert arch == oracle.compress(data), name
