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


def test_a_stream_through_three_contexts_beats_one_at_a_time():
    """The point of gated contexts: the coding tail and K1 of one input overlap the enumeration of another.  On an idle
    MI355X three contexts move 3 x 10^7-byte inputs at ~1.5x the rate of one context; the limit here only catches
    the overlap silently not happening (the gate around everything, contexts serialised)."""
    import numpy as np
    data = np.frombuffer(oracle.synth_text(31, 30_000_000), dtype=np.uint8)
    ins = [data] * 9

    def rate(contexts):
        with bce_amd.ContextPool(contexts, 0) as pool:
            pool.compress_many(ins[:contexts])                  # warm-up: buffers
            t0 = time.time()
            got = pool.compress_many(ins)
            return len(ins) * len(data) / (time.time() - t0), got

    r1, a1 = rate(1)
    r3, a3 = rate(3)
    assert all(bytes(x) == bytes(a1[0]) for x in a1 + a3)
    assert r3 > 1.1 * r1, "one context %.0f MB/s, three contexts %.0f MB/s" % (r1 / 1e6, r3 / 1e6)


def test_spine_bursts_keep_zero_run_contexts_cheap():
    """10^4 zero runs of up to 4 KB between random bytes: without the bursts the all-zero context is walked node by node by
    one wave per trie (seconds at this size), with them the whole input takes a fraction of a second."""
    import numpy as np
    rng = np.random.RandomState(3)
    data = b"".join(rng.bytes(int(rng.randint(1, 16))) + bytes(int(rng.randint(1, 4096))) for _ in range(10000))
    arch, dt = _encode_s(data)
    assert dt < 10.0, "encode took %.1f s" % dt
    assert bce_amd.decompress_device(arch) == data


def test_scan_of_1e8_bytes_stays_under_seconds():
    """`bce -s` at 10^8 bytes: the host part (recording 1.4 x 10^8 symbols in the reference's unordered_maps, then 9 x 29 x 6
    simulated adaptive codings) runs on a thread pool (scan_coder.cpp ScanSet); sequentially it took ~10 s.  The table is the
    one the sequential code gives (tests/test_core_cpu.py, tests/test_gpu_scan.py); this only guards the time."""
    data = bce_amd.synth_text(1, 100_000_000)
    bce_amd.scan(data[:1 << 20])                               # HIP init, kernels loaded
    t0 = time.time()
    cfg, sizes = bce_amd.scan(data)
    dt = time.time() - t0
    print("bce -s of 1e8 B: %.2f s" % dt)
    assert len(cfg) == 288 and dt < 4.0, "scan took %.1f s" % dt
