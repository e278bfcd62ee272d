"""GPU: inputs with long repeats (10^4..10^5 rounds holding a handful of nodes each) -- the persistent tail
kernel (k3_tail_kernel) against the oracle, and against the wide-round kernels."""
import os
import time

import pytest

import bce_amd
import oracle

pytestmark = pytest.mark.gpu


def repeat_input(n, rep, seed=21):
    base = oracle.synth_text(seed, n)
    chunk = base[1000:1000 + rep]
    return base[:n // 2] + chunk + base[n // 2:]


@pytest.mark.parametrize("n,rep", [(200000, 3000), (1 << 20, 20000)])
def test_long_repeat_matches_oracle(n, rep):
    data = repeat_input(n, rep)
    rf = bce_amd.RankFile(data)
    try:
        t0 = time.time()
        arch = bce_amd.BCE().encode(rf)
        dt = time.time() - t0
        st = bce_amd.stats(rf)
    finally:
        rf.close()
    assert arch == oracle.compress(data)
    assert st["rounds"] >= 8 * rep          # the repeat is walked bit by bit
    print("n=%d rep=%d rounds=%d encode %.3f s k3 %.1f ms launches %d" % (n, rep, st["rounds"], dt, st["k3_ms"], st["k3_launches"]))


def test_whole_file_duplicate_is_one_skipped_chain():
    """Two identical 2 MiB halves (+1 byte so the input is primitive): one 2-row chain 16 M rounds deep.  The
    depth-first tail must take it in one exact chain skip (wave-cooperative backward comparison of the text)."""
    half = oracle.synth_text(5, 2 << 20)
    data = half + half + b"#"
    rf = bce_amd.RankFile(data)
    try:
        t0 = time.time()
        arch = bce_amd.BCE().encode(rf)
        dt = time.time() - t0
        st = bce_amd.stats(rf)
    finally:
        rf.close()
    assert arch == oracle.compress(data)
    assert st["nodes"] == 8 * len(data) - 8 and st["rounds"] > 8 * (2 << 20)
    print("whole-file duplicate: encode %.2f s (time limits live in tests/test_gpu_perf.py)" % dt)
    assert bce_amd.decompress(arch) == data


def _encode_with_knobs(data, knobs):
    ctx = bce_amd.api._Ctx(0)
    try:
        for k, v in knobs.items():
            ctx.check(ctx.lib.bce_hip_debug_set(ctx.h, k, v), "bce_hip_debug_set")
        rf = bce_amd.RankFile(data, ctx=ctx)
        arch = bce_amd.BCE().encode(rf)
        return arch, bce_amd.stats(rf)
    finally:
        ctx.close()


@pytest.mark.parametrize("knobs", [{}, {1: 1}, {1: 1, 2: 1}, {3: 1}, {0: 50}, {0: 50, 2: 1}, {4: 1}, {1: 1, 4: 1}, {1: 1, 2: 1, 4: 1}, {6: 1}, {1: 1, 2: 1, 4: 1, 6: 1},
                                   {12: 1 << 30}, {12: 1 << 30, 1: 1}, {12: 1 << 30, 1: 1, 4: 1}, {12: 1 << 30, 1: 1, 4: 1, 6: 1}],
                         ids=["default", "no-dfs", "rounds-only", "no-skip", "dfs-short-passes", "dfs-short-passes-no-tail", "no-small", "no-dfs-no-small",
                              "wide-rounds-only", "three-launch-rounds", "three-launch-wide-rounds-only",
                              "small-lists", "lists-grow-no-dfs", "lists-grow-no-dfs-no-small", "lists-grow-three-launch-rounds"])
def test_every_enumeration_path_gives_the_same_archive(knobs):
    """Wide rounds, LDS tail, depth-first walkers (with / without chain skips, and giving up half way) are
    interchangeable: the archive and the node count never depend on which of them ran."""
    data = repeat_input(300000, 2500, seed=33)
    ref = oracle.compress(data)
    arch, st = _encode_with_knobs(data, knobs)
    assert arch == ref
    assert st["nodes"] == 8 * len(data) - 8


def test_exactly_periodic_input_with_a_large_period():
    """Two identical halves, nothing else: every rotation ties with one other.  The chain skip works on K1's
    group-head ranks; the encoder output is well defined (the reference's DEcoder fails on such inputs, quirk Q9;
    ours does not)."""
    half = oracle.synth_text(8, 1 << 20)
    data = half + half
    rf = bce_amd.RankFile(data)
    try:
        t0 = time.time()
        arch = bce_amd.BCE().encode(rf)
        dt = time.time() - t0
    finally:
        rf.close()
    assert arch == oracle.compress(data)
    assert bce_amd.decompress(arch) == data
    print("exactly periodic input: encode %.2f s" % dt)


def many_copies(copies, length, seed):
    """`copies` copies of one block, each followed by a different separator: chains of `copies` rows."""
    block = oracle.synth_text(seed, length)
    out = bytearray(oracle.synth_text(seed + 1, 20000))
    for i in range(copies):
        out += block + b"<%06d>" % (i * 7919 % 1000003)
    out += oracle.synth_text(seed + 2, 20000)
    return bytes(out)


@pytest.mark.parametrize("copies,length", [(40, 3000), (150, 600), (9, 200), (700, 40), (3, 70000)])
def test_many_row_chains_match_oracle(copies, length):
    """Chains of many rows: the skip compares all rows (one lane per row for the near bytes, then the whole wave
    per row pair), in batches of 64 rows; every path and the walkers without skipping give the oracle's archive."""
    data = many_copies(copies, length, 31)
    want = oracle.compress(data)
    for knobs in ({}, {3: 1}, {0: 7}, {1: 1}):
        arch, st = _encode_with_knobs(data, knobs)
        assert arch == want, "knobs %r" % (knobs,)
        assert st["nodes"] == 8 * len(data) - 8
    assert bce_amd.decompress(want) == data


def test_periodic_blocks_with_many_rows():
    """A block repeated back to back (periodic stretch, overlapping rows) inside other text."""
    unit = oracle.synth_text(77, 257)
    data = oracle.synth_text(78, 30000) + unit * 60 + oracle.synth_text(79, 30000) + unit * 12 + b"!"
    want = oracle.compress(data)
    for knobs in ({}, {3: 1}, {0: 5}):
        arch, _ = _encode_with_knobs(data, knobs)
        assert arch == want, "knobs %r" % (knobs,)


def _region(unit, length):
    return (unit * (length // len(unit) + 1))[:length]


@pytest.mark.parametrize("case", ["zeros", "ff", "period2", "period3", "period7-in-random", "two-regions", "at-start", "at-end",
                                  "record-table", "period2-twice", "three-regions-phases", "many-zero-runs", "same-length-twice",
                                  "whole-text-periodic", "two-tables-first-at-start"])
def test_staircase_chains_match_oracle(case):
    """Long runs of one byte and periodic tables (executables): chains whose rows are equally spaced text positions.
    The walkers expand them analytically (stair_run); every other path must give the same archive."""
    import numpy as np
    rs = np.random.RandomState(5)
    text = oracle.synth_text(91, 60000)
    rnd = rs.randint(0, 256, 60000).astype(np.uint8).tobytes()
    if case == "zeros":
        data = text[:30000] + bytes(9000) + text[30000:]
    elif case == "ff":
        data = text[:30000] + b"\xff" * 7000 + text[30000:] + b"\xff" * 100 + b"z"
    elif case == "period2":
        data = text[:20000] + _region(b"\x00\x02", 12001) + text[20000:]
    elif case == "period3":
        data = rnd[:20000] + _region(b"abc", 10000) + text[:20000] + _region(b"cab", 500) + rnd[20000:40000]
    elif case == "period7-in-random":
        data = rnd[:30000] + _region(b"\x01\x00\x00\x00\x00\x00\x80", 15000) + rnd[30000:]
    elif case == "two-regions":     # the same pattern twice: the node is a staircase only below the shorter one
        data = text[:20000] + bytes(6000) + text[20000:40000] + bytes(4000) + b"\x01" + bytes(300) + text[40000:]
    elif case == "at-start":        # the region reaches position 0: the walkers must not take the shortcut blindly
        data = bytes(5000) + text + _region(b"\x00\x07", 3000) + b"q"
    elif case == "at-end":
        data = text + _region(b"xy", 8000)
    elif case == "period2-twice":   # the same table in two places, same bytes before it: two rows leave together
        data = text[:20000] + b"\x00\x00" + _region(b"\x00\x02", 9001) + b"\x01" + text[20000:40000] + b"\x00\x00" + _region(b"\x00\x02", 7000) + text[40000:]
    elif case == "three-regions-phases":   # three places, different phases at the start and different bytes before
        data = (rnd[:10000] + b"Q" + _region(b"abc", 6000) + rnd[10000:20000] + b"R" + _region(b"bca", 5000) + b"a" + text[:20000] +
                b"Q" + _region(b"cab", 4001) + b"\xff" + rnd[20000:30000])
    elif case == "many-zero-runs":  # more regions than the closed form takes at once: the walkers take over until few are left
        data = b"".join(text[i * 3000:(i + 1) * 3000] + bytes(300 * (i + 1) + (i % 3)) for i in range(14)) + text[42000:]
    elif case == "whole-text-periodic":   # the region IS the text up to its last byte: it starts at position 0, where the
        data = b"ab" * 30000 + b"c"          # byte "before" is the last one (rotations)
    elif case == "two-tables-first-at-start":
        data = _region(b"\x00\x00\x01", 20000) + text[:30000] + b"\x07" + _region(b"\x00\x01\x00", 14000) + text[30000:]
    elif case == "same-length-twice":
        data = text[:20000] + b"A" + bytes(5000) + b"B" + text[20000:40000] + b"A" + bytes(5000) + b"C" + text[40000:]
    else:                           # 16-byte records that differ in one counter byte, then identical ones
        data = text[:10000] + b"".join(b"REC" + bytes([i & 255]) + bytes(12) for i in range(300)) + (b"REC\x00" + bytes(12)) * 400 + text[10000:]
    want = oracle.compress(data)
    for knobs in ({}, {3: 1}, {0: 9}, {1: 1}):
        arch, st = _encode_with_knobs(data, knobs)
        assert arch == want, "knobs %r" % (knobs,)
        assert st["nodes"] == 8 * len(data) - 8
    assert bce_amd.decompress_device(want) == data


def test_staircases_are_expanded_not_walked(capfd, monkeypatch):
    """The closed forms must actually run (the walkers alone would give the same archive, only slowly): the walker
    statistics (BCE_HIP_DFS_DEBUG) count single staircases and those of several regions."""
    import re
    text = oracle.synth_text(92, 60000)
    data = (text[:20000] + b"\x00\x00" + _region(b"\x00\x02", 9001) + b"\x01" + text[20000:40000] + b"\x00\x00" +
            _region(b"\x00\x02", 7000) + text[40000:50000] + bytes(5000) + text[50000:])
    monkeypatch.setenv("BCE_HIP_DFS_DEBUG", "1")
    arch, st = _encode_with_knobs(data, {})
    err = capfd.readouterr().err
    assert arch == oracle.compress(data)
    m = re.search(r"stairs (\d+) \+ (\d+) of several regions \((\d+) symbols\)", err)
    assert m, err[-2000:]
    assert int(m.group(1)) >= 1 and int(m.group(2)) >= 1 and int(m.group(3)) >= 5000


@pytest.mark.parametrize("data", [b"\xd2" * 14, b"a", b"\x00" * 1000, b"ab" * 7])
def test_inputs_without_nodes_on_every_path(data):
    """One byte repeated has no node at all (every plane is constant): every knob combination must still finish."""
    want = oracle.compress(data)
    for knobs in ({}, {1: 1, 2: 1, 4: 1}, {2: 1}, {2: 1, 4: 1, 6: 1}, {1: 1}):
        arch, _ = _encode_with_knobs(data, knobs)
        assert arch == want, "knobs %r" % (knobs,)


def test_megabyte_runs_do_not_crawl():
    """Two zero runs of 1.2 MB and 0.8 MB (19 M rounds): the closed forms keep the encoder at a fraction of a second
    (the walkers alone took a minute), the decoder's host tail keeps it at a few seconds; both exact."""
    text = oracle.synth_text(6, 600000)
    data = text[:200000] + bytes(1200000) + text[200000:400000] + bytes(800000) + b"\x01" + text[400000:]
    want = oracle.compress(data)
    rf = bce_amd.RankFile(data)
    try:
        t0 = time.time()
        arch = bce_amd.BCE().encode(rf)
        t_enc = time.time() - t0
    finally:
        rf.close()
    assert arch == want
    t0 = time.time()
    assert bce_amd.decompress_device(arch) == data
    t_dec = time.time() - t0
    print("megabyte runs: encode %.2f s, decode %.2f s (time limits live in tests/test_gpu_perf.py)" % (t_enc, t_dec))


# ---- workgroup-local rounds (k3_local_kernel): the stage between the wide rounds and the walkers ----
LOCAL_KNOBS = [
    ({8: 64}, "local-from-64"),                       # small live sets go through the local rounds too
    ({8: 64, 9: 3}, "budget-3"),                      # a workgroup hands on after 3 rounds: almost everything ends at the walkers
    ({8: 64, 9: 40, 0: 50}, "budget-40-short-walker-passes"),
    ({8: 64, 3: 1}, "no-chain-exit"),                 # no chain skipping: nothing leaves the local rounds early
    ({8: 64, 10: 17}, "tail-from-round-17"),          # the tail starts while the node count still doubles: lists overflow and spill
    ({8: 64, 10: 19, 9: 5}, "tail-from-round-19-budget-5"),
    ({7: 1}, "no-local"),
    ({11: 1}, "k4-beside-k3"),                        # model flushes on their own stream: double-buffered symbols, three-launch rounds
    ({11: 1, 8: 64}, "k4-beside-k3-local"),
]


@pytest.mark.parametrize("knobs,name", LOCAL_KNOBS, ids=[n for _, n in LOCAL_KNOBS])
@pytest.mark.parametrize("which", ["repeat", "copies", "rand", "text"])
def test_local_rounds_give_the_same_archive(which, knobs, name):
    data = {"repeat": lambda: repeat_input(300000, 2500, seed=34),
            "copies": lambda: many_copies(60, 900, 71),
            "rand": lambda: oracle.synth_rand(12, 150000),
            "text": lambda: oracle.synth_text(13, 700000)}[which]()
    ref = oracle.compress(data)
    arch, st = _encode_with_knobs(data, knobs)
    assert arch == ref
    assert st["nodes"] == 8 * len(data) - 8


def test_model_flushes_beside_the_rounds_with_many_small_flushes():
    """Knob 11 with a small symbol buffer: dozens of flushes, each running on the K4 stream while the next rounds fill the
    other pair of symbol buffers (the wide rounds fall back to three launches while a flush is in flight)."""
    data = oracle.synth_text(17, 3 << 20)
    ref = oracle.compress(data)
    ctx = bce_amd.api._Ctx(0)
    try:
        ctx.check(ctx.lib.bce_hip_debug_set(ctx.h, 11, 1), "bce_hip_debug_set")
        for cap in (200000, 1 << 20):
            rf = bce_amd.RankFile(data, ctx=ctx)
            arch = bce_amd.BCE(symbol_capacity=cap).encode(rf)
            assert arch == ref
            assert bce_amd.stats(rf)["flushes"] >= 3
    finally:
        ctx.close()


# ---- spine bursts (k3_dfs.hip): chains of many rows that lose a few rows per byte ---------------------------------------
def _spine_input(kind, seed=5):
    import numpy as np
    rng = np.random.RandomState(seed)
    text = oracle.synth_text(70 + seed, 120000)
    if kind == "zero-runs":            # runs of zeros of hundreds of different lengths between random bytes: the all-zero context keeps
        parts = []                     # thousands of rows for thousands of bytes, a few runs end at every length
        for _ in range(1500):
            parts.append(rng.bytes(int(rng.randint(1, 24))))
            parts.append(bytes(int(rng.randint(1, 3000))))
        return text[:60000] + b"".join(parts) + text[60000:]
    if kind == "equal-runs":           # hundreds of runs of the SAME length: the first rows of the chain all leave at once
        return text[:50000] + b"".join(rng.bytes(3) + bytes(2048) for _ in range(300)) + text[50000:] + b"".join(
            rng.bytes(5) + bytes(int(rng.randint(1500, 2500))) for _ in range(200))
    if kind == "u16-ramps":            # little-endian 16-bit arrays with small values: the context alternates between two bytes
        arrs = [np.minimum(rng.geometric(0.2, int(rng.randint(200, 4000))), 200).astype("<u2").tobytes() for _ in range(300)]
        return text[:30000] + b"".join(a + rng.bytes(2) for a in arrs) + text[30000:]
    if kind == "records":              # records of 12 bytes whose last 9 are constant, in tables of many lengths
        recs = [b"".join(rng.bytes(3) + b"\x00\x00\x00\x01\x00\x00\x00\xff\x10" for _ in range(int(rng.randint(50, 1500)))) for _ in range(200)]
        return text[:40000] + b"".join(r + rng.bytes(int(rng.randint(1, 9))) for r in recs) + text[40000:]
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["zero-runs", "equal-runs", "u16-ramps", "records"])
def test_spine_bursts_match_oracle(kind, monkeypatch):
    data = _spine_input(kind)
    ref = oracle.compress(data)
    arch, st = _encode_with_knobs(data, {})
    assert arch == ref
    monkeypatch.setenv("BCE_HIP_NO_SPINE", "1")
    arch2, st2 = _encode_with_knobs(data, {})
    assert arch2 == ref and st2["spine_levels"] == 0
    assert st["nodes"] == st2["nodes"] and st["rounds"] == st2["rounds"] and st["symbols"] == st2["symbols"]


def test_spine_bursts_are_taken():
    """The bursts must actually run on the input they were made for (the walkers alone give the same archive, slowly)."""
    data = _spine_input("zero-runs", seed=8)
    arch, st = _encode_with_knobs(data, {})
    assert arch == oracle.compress(data)
    assert st["spine_levels"] >= 500, st


@pytest.mark.parametrize("budget", [3, 50])
def test_spine_bursts_with_tiny_walker_budgets(budget):
    """Knob 0: walkers hand their work on every few nodes, so bursts start and end at arbitrary levels and passes."""
    data = _spine_input("zero-runs", seed=9)[:900000]
    arch, st = _encode_with_knobs(data, {0: budget})
    assert arch == oracle.compress(data)
