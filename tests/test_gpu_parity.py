"""GPU parity: every stage of the HIP path against the CPU oracle, through the C ABI (libbcehip.so).

Bit-exact everywhere (the path is integer/byte work).  Sizes are chosen so the oracle finishes in
seconds; full-size runs are covered by size-independent properties in test_gpu_properties.py.
"""
import hashlib

import numpy as np
import pytest

import bce_amd
import oracle
from conftest import edge_inputs, golden_input, load_golden

pytestmark = pytest.mark.gpu


def small_inputs():
    return edge_inputs() + [
        ("text-64k", oracle.synth_text(1, 65536)),
        ("rand-64k", oracle.synth_rand(1, 65536)),
        ("text-300k", oracle.synth_text(2, 300000)),
    ]


IDS = [n for n, _ in small_inputs()]


@pytest.mark.parametrize("name,data", small_inputs(), ids=IDS)
def test_k1_bwt_and_offset(name, data):
    """K1 == File::rotate + File::bwt (bce.cpp:858-910)."""
    bwt, off = oracle.bwt_stage(data)
    rf = bce_amd.RankFile(data, build=False)
    try:
        assert rf.offset() == off
        assert bytes(rf.bwt()) == bytes(bwt)
    finally:
        rf.close()


@pytest.mark.parametrize("name,data", small_inputs(), ids=IDS)
def test_k2_planes_and_rank(name, data):
    """K2 == RankFile ctor + Rank::build/get (bce.cpp:944-970,138-151): plane bits, zero counts, rank1."""
    bwt, off = oracle.bwt_stage(data)
    ref_bits = oracle.plane_bits(bwt)
    rf = bce_amd.RankFile(bwt=bwt, offset=off)
    try:
        n = len(data)
        assert rf.zeros == [int(n - ref_bits[j].sum()) for j in range(8)]
        rs = np.random.RandomState(0)
        idx = np.unique(np.concatenate([np.arange(0, min(n, 300) + 1), [n], rs.randint(0, n + 1, 2000)])).astype(np.uint32)
        for j in range(8):
            assert (rf.plane_bits(j) == ref_bits[j]).all(), "plane %d" % j
            cum = np.concatenate([[0], np.cumsum(ref_bits[j], dtype=np.int64)])
            assert (rf.rank1(j, idx) == cum[idx]).all(), "rank plane %d" % j
    finally:
        rf.close()


def expected_symbol_records(tr, config=None):
    """Oracle (plane, s, k, c1, c2, cs) tuples -> (plane, s', k', nesc, esc, slot) as K3 packs them."""
    cfg = np.frombuffer(config, dtype=np.uint8).reshape(9, 32) if config is not None else DEFAULT_CFG
    ctxoff = np.zeros((8, 32), dtype=np.int64)
    for p in range(8):
        acc = 0
        for k in range(2, 32):
            ctxoff[p, k] = acc
            acc += 1 << (2 * int(cfg[p, k]))
    out = []
    for plane, s, k, c1, c2, cs in tr["syms"].tolist():
        nesc = esc = 0
        while k > 31:
            esc |= (s & 1) << nesc
            nesc += 1
            k = (k + ((~s) & 1)) >> 1
            s >>= 1
        bits = int(cfg[plane, k])
        ctx = ((((c1 << bits) & 0xFFFFFFFF) // cs) << bits) | (((c2 << bits) & 0xFFFFFFFF) // cs)
        out.append((plane, s, k, nesc, esc, int(ctxoff[plane, k]) + ctx))
    return np.array(out, dtype=np.uint32).reshape(-1, 6)


DEFAULT_CFG = np.array([
    [0, 0, 5, 5, 5] + [4] * 26 + [0],
    [0, 0, 5, 5, 5] + [4] * 26 + [0],
    [0, 0, 5, 5, 5] + [4] * 22 + [3] * 4 + [0],
    [0, 0, 5, 5, 5] + [4] * 17 + [3] * 9 + [0],
    [0, 0, 5, 5] + [4] * 8 + [3] * 19 + [0],
    [0, 0, 5, 5] + [4] * 8 + [3] * 19 + [0],
    [0, 0, 5] + [4] * 6 + [3] * 22 + [0],
    [0, 0] + [4] * 4 + [3] * 19 + [2] * 6 + [0],
    [0] * 32], dtype=np.uint8)


def test_default_config_table_is_32_wide():
    assert DEFAULT_CFG.shape == (9, 32)


@pytest.mark.parametrize("name,data", small_inputs(), ids=IDS)
def test_k3_rounds_nodes_and_symbols(name, data):
    """K3 == BCE::code mode 1 (bce.cpp:1236-1374): per-round node lists, symbol tuples, K4 model outputs."""
    bwt, off = oracle.bwt_stage(data)
    tr = oracle.trace_encode_from_bwt(bwt, off)
    nodes = tr["nodes"]
    rf = bce_amd.RankFile(bwt=bwt, offset=off)
    bce = bce_amd.BCE()
    try:
        bce.code_begin(rf)
        rounds = int(nodes[:, 0].max()) + 1 if len(nodes) else 0
        for r in range(rounds):
            sel = nodes[nodes[:, 0] == r]
            for p in range(8):
                exp = sel[sel[:, 1] == p][:, 2:5]
                got = bce.code_nodes(rf, p)
                assert got.shape == exp.shape and (got == exp).all(), "round %d plane %d" % (r, p)
            nxt = bce.code_round(rf)
            assert nxt == int((nodes[:, 0] == r + 1).sum())
        exp_syms = expected_symbol_records(tr)
        got_syms = bce.code_symbols(rf, len(exp_syms) + 16)
        assert got_syms.shape == exp_syms.shape
        assert (got_syms == exp_syms).all()
        # K4: (cum, freq, total) per symbol == the oracle's adaptive range-coder ops, per coder in order
        got_ops = bce.code_model(rf, len(exp_syms) + 16)
        ops = tr["ops"]
        for p in range(8):
            mine = ops[ops[:, 0] == p][:, 1:4]
            row = DEFAULT_CFG[p]
            pre = 32 + int((np.diff(np.concatenate([[0], row.astype(np.int64)])) != 0).sum()) + 1
            mine = mine[pre:]
            exp = []
            for (plane, s, k, nesc, esc, slot), (cum, freq, total) in zip(got_syms[got_syms[:, 0] == p].tolist(),
                                                                          got_ops[got_syms[:, 0] == p].tolist()):
                for b in range(nesc):
                    exp.append(((esc >> b) & 1, 1, 2))
                exp.append((cum, freq, total))
            exp = np.array(exp, dtype=np.uint32).reshape(-1, 3)
            assert exp.shape == mine.shape and (exp == mine).all(), "coder %d" % p
    finally:
        rf.close()


@pytest.mark.parametrize("name,data", small_inputs(), ids=IDS)
def test_archive_bit_exact_small(name, data):
    """Whole path (K1..K4 + host coder) == reference archive."""
    assert bce_amd.compress(data) == oracle.compress(data)


@pytest.mark.parametrize("v", load_golden(), ids=lambda v: v["name"])
def test_archive_matches_reference_golden(v):
    """The committed reference hashes, without going through the oracle."""
    a = bce_amd.compress(golden_input(v))
    assert len(a) == v["archive_bytes"]
    assert hashlib.sha256(a).hexdigest() == v["archive_sha256"]


def test_custom_config_and_small_flushes():
    """A scanned-style config (.bcc) and a tiny symbol buffer (many K4 flushes) give the same archive."""
    rnd = np.random.RandomState(3)
    cfg = rnd.randint(0, 6, size=288).astype(np.uint8).tobytes()
    data = oracle.synth_text(4, 200000)
    ref = oracle.compress(data, cfg)
    rf = bce_amd.RankFile(data)
    try:
        assert bce_amd.BCE(cfg).encode(rf) == ref
    finally:
        rf.close()
    rf = bce_amd.RankFile(data)
    try:
        assert bce_amd.BCE(cfg, symbol_capacity=40000).encode(rf) == ref
        assert bce_amd.stats(rf)["flushes"] > 3
    finally:
        rf.close()


def test_rejects_bad_arguments():
    with pytest.raises(bce_amd.BceError):
        bce_amd.compress(b"")
    with pytest.raises(ValueError):
        bce_amd.BCE(b"\x00" * 10)
    rf = bce_amd.RankFile(b"hello world")
    try:
        with pytest.raises(bce_amd.BceError):
            bce_amd.BCE(bytes([9]) * 288).encode(rf)     # context bits > 5 cannot be serialised
    finally:
        rf.close()


def test_node_lists_grow_when_a_round_does_not_fit():
    """The node lists start with n / 8 (+ 4096) nodes each; a round whose children would not fit is not run, the lists are
    doubled, the round's lists moved over, and it runs again (k3_grow_lists; test knob 12 = the divisor).  Lists of 4096 nodes
    on inputs that need tens of thousands (rounds only, knob 1: the depth-first tail would take inputs this small over before
    the lists fill): several doublings, on the way up and in the stepping interface, the same archive."""
    for data in (oracle.synth_rand(2, 300000), oracle.synth_text(9, 2_000_000), oracle.synth_rand(5, 70000) + oracle.synth_text(5, 500000)):
        ctx = bce_amd.api._Ctx(0)
        try:
            ctx.check(ctx.lib.bce_hip_debug_set(ctx.h, 12, 1 << 30), "bce_hip_debug_set")
            ctx.check(ctx.lib.bce_hip_debug_set(ctx.h, 1, 1), "bce_hip_debug_set")     # (no depth-first tail: it would take small inputs over before the lists fill)
            rf = bce_amd.RankFile(data, ctx=ctx)
            assert bce_amd.BCE().encode(rf) == oracle.compress(data)
            st = bce_amd.stats(rf)
            assert st["list_grows"] >= 2 and st["list_nodes"] >= 4096 * 4, st
            # the same context again without the knob: the lists it has are kept, nothing grows
            ctx.check(ctx.lib.bce_hip_debug_set(ctx.h, 12, 0), "bce_hip_debug_set")
            rf2 = bce_amd.RankFile(data, ctx=ctx)
            assert bce_amd.BCE().encode(rf2) == oracle.compress(data)
            assert bce_amd.stats(rf2)["list_grows"] == 0
        finally:
            ctx.close()
    # the default: n / 8 is room enough for text (no growth), random bytes may take one doubling
    rf = bce_amd.RankFile(oracle.synth_text(9, 2_000_000))
    try:
        bce_amd.BCE().encode(rf)
        assert bce_amd.stats(rf)["list_grows"] == 0
    finally:
        rf.close()
    # stepping (bce_hip_enum_round): every round's node lists against the oracle's trace, lists growing under it
    data = oracle.synth_rand(3, 60000)
    bwt, off = oracle.bwt_stage(data)
    nodes = oracle.trace_encode_from_bwt(bwt, off)["nodes"]
    ctx = bce_amd.api._Ctx(0)
    try:
        ctx.check(ctx.lib.bce_hip_debug_set(ctx.h, 12, 1 << 30), "bce_hip_debug_set")
        rf = bce_amd.RankFile(bwt=bwt, offset=off, ctx=ctx)
        bce = bce_amd.BCE()
        bce.code_begin(rf)
        for r in range(int(nodes[:, 0].max()) + 1):
            sel = nodes[nodes[:, 0] == r]
            for p in range(8):
                exp = sel[sel[:, 1] == p][:, 2:5]
                got = bce.code_nodes(rf, p)
                assert got.shape == exp.shape and (got == exp).all(), "round %d plane %d" % (r, p)
            assert bce.code_round(rf) == int((nodes[:, 0] == r + 1).sum())
    finally:
        ctx.close()


def test_symbol_buffer_grows_when_one_round_exceeds_it():
    """A round that emits more symbols than the whole buffer makes the driver enlarge it (k3_grow_symbols)."""
    data = oracle.synth_rand(1, 65536)
    rf = bce_amd.RankFile(data)
    try:
        assert bce_amd.BCE(symbol_capacity=1000).encode(rf) == oracle.compress(data)
        assert bce_amd.stats(rf)["flushes"] > 5
    finally:
        rf.close()
