"""GPU: wall-clock limits, kept apart from the parity tests (a parity test must not fail because a shared box was slow).
Generous: each limit is ~10x what the path takes on an idle MI355X; the failure they guard against is a fast path that
silently stopped being taken (seconds turning into minutes)."""
import time

import pytest

import bce_amd
import oracle

pytestmark = [pytest.mark.gpu, pytest.mark.perf]


def _encode_s(data):
    rf = bce_amd.RankFile(data)
    try:
        t0 = time.time()
        arch = bce_amd.BCE().encode(rf)
        return arch, time.time() - t0
    finally:
        rf.close()


def test_whole_file_duplicate_takes_the_chain_skip():
    half = oracle.synth_text(5, 2 << 20)
    _, dt = _encode_s(half + half + b"#")
    assert dt < 10.0, "16 M rounds walked node by node? encode took %.1f s" % dt


def test_megabyte_runs_take_the_closed_forms():
    text = oracle.synth_text(6, 600000)
    data = text[:200000] + bytes(1200000) + text[200000:400000] + bytes(800000) + b"\x01" + text[400000:]
    arch, t_enc = _encode_s(data)
    t0 = time.time()
    assert bce_amd.decompress_device(arch) == data
    t_dec = time.time() - t0
    assert t_enc < 10.0 and t_dec < 40.0, "encode %.1f s, decode %.1f s" % (t_enc, t_dec)
