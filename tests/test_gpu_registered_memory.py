"""GPU: host memory registered with the runtime (hipHostRegister) -- the encoder's flush slots and the decoder's boundary
ranks -- grows, is given back and lives through fork()s of the host program.

Round 4 ended two runs of the suite in "Memory access fault by GPU ... write access to a read-only page" at an address inside
the C library's heap.  What stands against that by construction (bce_amd/csrc/common.h, "host memory registered
with the runtime"): a registered range is always a private mapping of its own, MADV_DONTFORK, its pages fixed after the first
touch, and every unregister waits for the work that touches the range.  This is the one deterministic test of that route:
BCE_HIP_REG_MIN lowers the sizes from which buffers take it (8 MB / 256 MB by default) so that small inputs reach it, the
sizes go up and down so that the buffers are replaced many times, and a child process is forked between the steps while the
mappings are live.  Not a soak: one pass, fixed inputs, every result checked against the oracle."""
import subprocess
import sys

import numpy as np
import pytest

import bce_amd
import oracle

pytestmark = pytest.mark.gpu


def _fork():
    r = subprocess.run([sys.executable, "-c", "pass"])          # fork + exec of the host program with live registered mappings
    assert r.returncode == 0


def test_registered_buffers_grow_are_freed_and_survive_forks(monkeypatch):
    monkeypatch.setenv("BCE_HIP_REG_MIN", "4096")                # every slot / boundary-rank buffer of >= 4 KB is a registered mapping
    monkeypatch.setenv("BCE_DEC_FORCE_HOST_TAIL", "1")           # every decode takes its tail to the host: the boundary ranks come over
    sizes = [3000, 40000, 9000, 300000, 70000, 1200000, 5000, 2500000, 600000]
    ctx = bce_amd.api._Ctx(0)
    try:
        before = bce_amd.stats_of(ctx)
        assert before["reg_maps"] == 0 and before["reg_unmaps"] == 0
        peak = 0
        for i, n in enumerate(sizes):
            data = (oracle.synth_text(100 + i, n) if i % 3 else oracle.synth_text(100 + i, n // 2) + bytes(n // 8) + oracle.synth_rand(i, n - n // 2 - n // 8))
            want = oracle.compress(data)
            got = bce_amd.compress(data, ctx=ctx)                 # flush slots: 8n + 1024 records each, replaced whenever n exceeds the peak
            assert bytes(got) == want, "compress, n = %d" % n
            _fork()
            out = np.empty(len(data), dtype=np.uint8)
            assert bce_amd.decompress_device(want, ctx=ctx, out=out) == len(data)   # boundary ranks: 32 (n + 1) bytes, replaced likewise
            assert out.tobytes() == data, "decode, n = %d" % n
            _fork()
            st = bce_amd.stats_of(ctx)
            if n > peak:
                peak = n
            assert st["reg_maps"] > 0
        st = bce_amd.stats_of(ctx)
        # four growing sizes: each replaces the three slots that were used and the boundary ranks -- the route really was taken
        assert st["reg_maps"] >= 8 and st["reg_unmaps"] >= 4, st
        assert st["reg_unmaps"] < st["reg_maps"]                  # (the last generation is still held by the context)
    finally:
        ctx.close()
    # a second context beside a pool that forks while the contexts work: archives still the oracle's
    datas = [oracle.synth_text(200 + i, 200000 + 50000 * i) for i in range(6)]
    with bce_amd.ContextPool(2, 0) as pool:
        res = pool.compress_many(datas[:3])
        _fork()
        res += pool.compress_many(datas[3:])
    for d, a in zip(datas, res):
        assert bytes(a) == oracle.compress(d)


def test_default_thresholds_leave_small_buffers_to_the_runtime():
    """Without the knob a small input's slots are hipHostMalloc memory: nothing is registered."""
    ctx = bce_amd.api._Ctx(0)
    try:
        data = oracle.synth_text(7, 50000)
        assert bytes(bce_amd.compress(data, ctx=ctx)) == oracle.compress(data)
        st = bce_amd.stats_of(ctx)
        assert st["reg_maps"] == 0 and st["reg_unmaps"] == 0
    finally:
        ctx.close()
