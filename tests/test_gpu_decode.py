"""GPU: the GPU-assisted decoder (kd_decode.hip) -- decode(reference archive) == input.

Archives come from the oracle (the reference's `-c`), so this is parity with the reference's `-d` on the reference's
own output; the host decoder (decoder.cpp, CPU-tested against the same archives) must agree byte for byte."""
import numpy as np
import time

import pytest

import bce_amd
import oracle
from conftest import edge_inputs

pytestmark = pytest.mark.gpu


def _cases():
    rs = np.random.RandomState(5)
    cases = list(edge_inputs())
    cases += [
        ("text-300k", oracle.synth_text(3, 300000)),
        ("rand-100k", oracle.synth_rand(4, 100000)),
        ("text-2M", oracle.synth_text(1, 2 << 20)),
        ("binary-low-planes", (rs.randint(0, 4, 70000).astype(np.uint8) << 3).tobytes()),
        ("runs", np.repeat(rs.randint(0, 256, 3000).astype(np.uint8), rs.randint(1, 60, 3000)).tobytes()),
        ("repeat-20k", oracle.synth_text(8, 150000) + oracle.synth_text(8, 150000)[40000:60000] + b"#"),
    ]
    return cases


CASES = _cases()


@pytest.mark.parametrize("name,data", CASES, ids=[c[0] for c in CASES])
def test_gpu_decoder_inverts_reference_archives(name, data):
    data = bytes(data)
    arch = oracle.compress(data)
    assert bce_amd.decompress_device(arch) == data
    if len(data) <= 400000:
        assert bce_amd.decompress(arch) == data          # the host decoder agrees


def test_periodic_inputs():
    """Several LF cycles (the reference's decoder returns zeros here): the walk from row 0 goes round one
    cycle; the GPU path writes that cycle once and unrolls it, decoder.cpp walks it n steps."""
    for data in (b"ab" * 500, b"abcabcabd" * 3000, oracle.synth_text(2, 5000) * 7):
        arch = oracle.compress(data)
        assert bce_amd.decompress_device(arch) == data
        assert bce_amd.decompress(arch) == data


def test_tail_kernels_and_plain_rounds_agree(monkeypatch):
    """Forced rounds on the device (dec_tail64_kernel / dec_tail_kernel) against the same rounds run one by one."""
    base = oracle.synth_text(21, 120000)
    data = base[:60000] + base[1000:3000] + base[60000:] + base[5000:5400] * 3
    arch = oracle.compress(data)
    assert bce_amd.decompress_device(arch) == data
    monkeypatch.setenv("BCE_DEC_NO_TAIL", "1")
    assert bce_amd.decompress_device(arch) == data


def test_tail_query_rounds_with_and_without_the_mailbox(monkeypatch):
    """Tails in which many rounds ask something (a row leaves a run or a table every few rounds): the resident kernels
    answered through the mailbox, and the leave / answer / relaunch protocol they fall back to."""
    text = oracle.synth_text(23, 80000)
    data = (text[:30000] + bytes(6000) + text[30000:50000] + (b"\x00\x02" * 2500) + b"\x07" + text[50000:] +
            bytes(3000) + b"\x01" + (b"\x00\x02" * 1800))
    arch = oracle.compress(data)
    assert bce_amd.decompress_device(arch) == data
    monkeypatch.setenv("BCE_DEC_NO_MAILBOX", "1")
    assert bce_amd.decompress_device(arch) == data


def test_custom_config_archives_decode():
    data = oracle.synth_text(12, 200000)
    cfg, _ = oracle.scan(data)
    arch = oracle.compress(data, bytes(cfg))
    assert bce_amd.decompress_device(arch) == data


def test_one_context_many_archives_and_roundtrip_with_the_gpu_encoder():
    ctx = bce_amd.api._Ctx(0)
    try:
        for seed, n in ((1, 50000), (2, 700000), (3, 1000), (4, 1 << 20)):
            data = oracle.synth_text(seed, n)
            arch = bce_amd.compress(data)
            assert bce_amd.decompress_device(arch, ctx=ctx) == data
    finally:
        ctx.close()


def test_corrupt_archive_is_rejected_or_differs():
    data = oracle.synth_text(5, 100000)
    arch = bytearray(oracle.compress(data))
    arch[len(arch) // 2] ^= 0x55
    try:
        out = bce_amd.decompress_device(bytes(arch))
    except bce_amd.BceError:
        return
    assert out != data


def test_deep_tail_finishes_on_the_host(monkeypatch):
    """A run long enough that the wave tail kernel hands the rest of the rounds to the host (dec_host_tail), and the
    same archive with that switched off."""
    text = oracle.synth_text(29, 40000)
    data = text[:20000] + bytes(150000) + text[20000:] + bytes(90000) + b"\x01"
    arch = oracle.compress(data)
    t0 = time.time()
    assert bce_amd.decompress_device(arch) == data
    dt = time.time() - t0
    monkeypatch.setenv("BCE_DEC_NO_HOST_TAIL", "1")
    t0 = time.time()
    assert bce_amd.decompress_device(arch) == data
    print("deep tail: %.2f s with the host tail, %.2f s without" % (dt, time.time() - t0))


def _with_long_tail(n_text, seed):
    """Text with big duplicated stretches: wide (six-launch) rounds first, then a tail of tens of thousands of rounds."""
    t = np.frombuffer(bytes(bce_amd.synth_text(seed, n_text)), dtype=np.uint8)
    parts = [t, t[n_text // 5:n_text // 5 + 300000], np.zeros(120000, dtype=np.uint8), t[n_text // 2:n_text // 2 + 200000], t[:77777]]
    return np.concatenate(parts)


def test_decoder_paths_of_round_three_agree(monkeypatch):
    """The switches added in round 3 change the route, never the bytes: six-launch rounds plane by plane or all planes at
    once (BCE_DEC_NO_SPLIT), the long tail taken over by the host at 8192 nodes a round, after the probe (0) or much
    earlier, its pinned buffer allocated beside the rounds or when the tail begins, the host tail on one thread.
    The input is large enough for rounds of > 2 M nodes (six launches) and has a tail long enough for the host."""
    data = _with_long_tail(40 << 20, 41)
    arch = bce_amd.compress(data)
    ref = bce_amd.decompress_device(arch)
    assert ref == data.tobytes()
    for env in ({"BCE_DEC_NO_SPLIT": "1"}, {"BCE_DEC_HOST_ENTER": "0"}, {"BCE_DEC_HOST_ENTER": "200000"}, {"BCE_DEC_NO_EARLY_PIN": "1"},
                {"BCE_DEC_TAIL_SERIAL": "1"}, {"BCE_DEC_NO_SMALL": "1", "BCE_DEC_NO_SPLIT": "1"}, {"BCE_DEC_NO_SMALL": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        assert bce_amd.decompress_device(arch) == ref, env
        for k in env:
            monkeypatch.delenv(k)


def test_one_context_decodes_wide_then_small_then_wide():
    """The decoder's pinned query buffers stay with the context and only grow: a small archive after a large one (and
    the large one again) must not see stale sizes or contents."""
    big = np.frombuffer(bytes(bce_amd.synth_text(5, 9 << 20)), dtype=np.uint8)
    small = np.frombuffer(bytes(bce_amd.synth_rand(6, 70000)), dtype=np.uint8)
    a_big, a_small = bce_amd.compress(big), bce_amd.compress(small)
    ctx = bce_amd.api._Ctx(0)
    try:
        for arch, want in ((a_big, big), (a_small, small), (a_big, big), (a_small, small)):
            assert bce_amd.decompress_device(arch, ctx=ctx) == want.tobytes()
    finally:
        ctx.close()


def test_decode_into_a_callers_buffer():
    """decompress_device(out=...) is the C ABI's own shape: the bytes land in the caller's array, the length comes back."""
    data = np.frombuffer(bytes(bce_amd.synth_text(3, 300000)), dtype=np.uint8)
    arch = bce_amd.compress(data)
    buf = np.full(len(data) + 100, 0xEE, dtype=np.uint8)
    assert bce_amd.decompress_device(arch, out=buf) == len(data)
    assert np.array_equal(buf[:len(data)], data) and (buf[len(data):] == 0xEE).all()
    with pytest.raises(bce_amd.BceError):
        bce_amd.decompress_device(arch, out=np.zeros(len(data) - 1, dtype=np.uint8))      # too small: refused, nothing written past it
    with pytest.raises(ValueError):
        bce_amd.decompress_device(arch, out=np.zeros(len(data), dtype=np.uint16))


def test_node_lists_of_the_decoder_grow_by_starting_again(capfd, monkeypatch):
    """The decoder's node lists start with n / 8 (+ 4096) nodes per plane; a round whose children would not fit stops the decode
    before it writes them, and the decode starts again with twice the room (kd_decode.hip: decompress_device_body).  Lists of
    4096 nodes (test knob 12) on inputs whose rounds hold tens of thousands of nodes per plane: several restarts, the right
    bytes; then the same context without the knob decodes in one go (its lists are large by now)."""
    monkeypatch.setenv("BCE_ALLOC_TRACE", "1")
    for data in (oracle.synth_rand(12, 400000), oracle.synth_text(12, 3_000_000), bytes(_with_long_tail(1_500_000, 7))):
        arch = oracle.compress(data)
        ctx = bce_amd.api._Ctx(0)
        try:
            ctx.check(ctx.lib.bce_hip_debug_set(ctx.h, 12, 1 << 30), "bce_hip_debug_set")
            capfd.readouterr()
            assert bce_amd.decompress_device(arch, ctx=ctx) == data
            err = capfd.readouterr().err
            assert err.count("does not fit the node lists") >= 2, err[-2000:]
            ctx.check(ctx.lib.bce_hip_debug_set(ctx.h, 12, 0), "bce_hip_debug_set")
            assert bce_amd.decompress_device(arch, ctx=ctx) == data
            assert "does not fit the node lists" not in capfd.readouterr().err
        finally:
            ctx.close()
