"""CPU: the per-element arithmetic shared with the HIP kernels (bce_amd/csrc/bce_core.h) and the host
range coder / framing (host_coder.cpp), driven sequentially by tests/core_emul.cpp, against the oracle."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from conftest import ROOT, edge_inputs


@pytest.fixture(scope="module")
def emul():
    out = os.path.join(ROOT, "tests", "_build", "libcore_emul.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    srcs = [os.path.join(ROOT, "tests", "core_emul.cpp"), os.path.join(ROOT, "bce_amd", "csrc", "host_coder.cpp"),
            os.path.join(ROOT, "bce_amd", "csrc", "scan_coder.cpp")]
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", out] + srcs + ["-lpthread"])
    L = C.CDLL(out)
    L.emul_encode_from_bwt.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.emul_rank1.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]
    L.emul_free.argtypes = [C.c_void_p]
    L.emul_check_recip.argtypes = [C.c_uint64, C.c_uint64]
    L.emul_check_recip.restype = C.c_uint64
    L.emul_scan.argtypes = [C.c_void_p, C.c_size_t, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p]
    return L


def run_emul(L, bwt, off, config=None):
    a = np.ascontiguousarray(bwt)
    cfg = np.frombuffer(config, dtype=np.uint8) if config is not None else None
    out, n, nd, sy = C.c_void_p(), C.c_size_t(), C.c_uint64(), C.c_uint64()
    L.emul_encode_from_bwt(a.ctypes.data, len(a), off, cfg.ctypes.data if cfg is not None else None,
                           C.byref(out), C.byref(n), C.byref(nd), C.byref(sy))
    r = C.string_at(out, n.value)
    L.emul_free(out)
    return r, nd.value, sy.value


@pytest.mark.parametrize("name,data", edge_inputs(), ids=[n for n, _ in edge_inputs()])
def test_core_matches_oracle_edges(emul, name, data):
    bwt, off = oracle.bwt_stage(data)
    arch, _, _ = run_emul(emul, bwt, off)
    assert arch == oracle.compress(data)


@pytest.mark.parametrize("gen,seed,n", [("synth_text", 1, 65536), ("synth_rand", 1, 65536), ("synth_text", 2, 300000)])
def test_core_matches_oracle_synth(emul, gen, seed, n):
    data = getattr(oracle, gen)(seed, n)
    bwt, off = oracle.bwt_stage(data)
    arch, nodes, syms = run_emul(emul, bwt, off)
    tr = oracle.trace_encode_from_bwt(bwt, off)
    assert arch == tr["archive"]
    assert nodes == len(tr["nodes"]) == 8 * n - 8
    assert syms == len(tr["syms"])


def test_core_custom_config(emul):
    rnd = np.random.RandomState(3)
    cfg = rnd.randint(0, 6, size=288).astype(np.uint8).tobytes()
    data = oracle.synth_text(4, 50000)
    bwt, off = oracle.bwt_stage(data)
    arch, _, _ = run_emul(emul, bwt, off, cfg)
    assert arch == oracle.compress(data, cfg)


def test_granule_rank_matches_reference_rank(emul):
    data = oracle.synth_rand(11, 10000)
    bwt, _ = oracle.bwt_stage(data)
    bits = oracle.plane_bits(bwt)
    idx = np.concatenate([np.arange(0, 400), np.array([95, 96, 97, 191, 192, 3071, 3072, 9999, 10000]),
                          np.random.RandomState(0).randint(0, 10001, 500)]).astype(np.uint32)
    for plane in range(8):
        out = np.empty(len(idx), dtype=np.uint32)
        a = np.ascontiguousarray(bwt)
        emul.emul_rank1(a.ctypes.data, len(a), plane, idx.ctypes.data, len(idx), out.ctypes.data)
        cum = np.concatenate([[0], np.cumsum(bits[plane])])
        assert (out == cum[idx]).all()


def test_host_coder_reciprocal_division_is_exact(emul):
    """step = (h - l) / total (bce.cpp:527) is done by multiplication in the host coder: must be the exact quotient."""
    assert emul.emul_check_recip(12345, 2000) == 0


def test_context_index_u32_wrap_matches_reference(emul):
    """get_context (bce.cpp:671-677) computes (c1 << bits) / cs in uint32: for c1 >= 2^(32-bits) -- inputs of 2^27
    bytes and more, BASELINE configs 3/4 -- the shift wraps.  The kernels divide with a float
    reciprocal + exact correction (bce_core.h small_quotient, same arithmetic on host and device): check it against
    the oracle's plain uint32 expression on operands that wrap, for every context width 1..5 and k = 2..31."""
    emul.emul_context.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    emul.emul_small_quotient.argtypes = [C.c_uint32, C.c_uint32]
    emul.emul_small_quotient.restype = C.c_uint32
    rs = np.random.RandomState(20)
    for bits in range(0, 6):
        row = np.full(32, bits, dtype=np.uint8)
        cases = []
        for _ in range(4000):
            # cs up to 2^31 - 1 (n < 2^31), c1, c2 < cs; half the cases force c1 into the wrapping range
            cs = int(rs.randint(1 << 27, (1 << 31) - 1)) if rs.rand() < 0.8 else int(rs.randint(2, 1 << 27))
            c1 = int(rs.randint(0, cs))
            if bits and rs.rand() < 0.5 and cs > (1 << (32 - bits)):
                c1 = int(rs.randint(1 << (32 - bits), cs))
            c2 = int(rs.randint(0, cs))
            cases.append((int(rs.randint(2, 32)), c1, c2, cs))
        # edges: exactly at the wrap, one below, cs - 1, powers of two
        for cs in ((1 << 31) - 1, (1 << 30) + 1, (1 << 27) + 5, 1 << 28):
            for c1 in (cs - 1, (1 << (32 - bits)) % cs if bits else 0, max(0, (1 << (32 - bits)) - 1) % cs if bits else 1, cs // 2, 0):
                cases.append((2, c1, cs - 1 - (c1 % 2), cs))
        arr = np.array(cases, dtype=np.uint32)
        out = np.empty((len(cases), 2), dtype=np.uint32)
        emul.emul_context(row.ctypes.data, arr.ctypes.data, len(cases), out.ctypes.data)
        nwrap = 0
        for (k, c1, c2, cs), (ctx, slot) in zip(cases, out):
            want = oracle.context_index(bits, c1, c2, cs)
            assert ctx == want, (bits, k, c1, c2, cs, int(ctx), want)
            assert want < (1 << (2 * bits))
            nwrap += (c1 << bits) >= (1 << 32)
            # slot = first slot of k + ctx  (plane_cfg_init: ctxoff[k] = sum over k' < k of 4^bits)
            assert slot == (k - 2) * (1 << (2 * bits)) + want
        if bits >= 2:          # (c1 < 2^31, so bits = 1 cannot wrap)
            assert nwrap > 500
    # the quotient primitive itself, on wrapped dividends
    for _ in range(20000):
        b = int(rs.randint(1 << 26, (1 << 31) - 1))
        a = int(rs.randint(0, min((1 << 32) - 1, 32 * b - 1)))
        assert emul.emul_small_quotient(a, b) == a // b


@pytest.mark.parametrize("gen,seed,n", [("synth_text", 1, 1 << 20), ("synth_rand", 5, 150000), ("synth_text", 9, 70000)])
def test_scan_coder_sequential_and_threaded_equal_the_oracle(emul, gen, seed, n):
    """`bce -s` host part (scan_coder.cpp): ScanCoder::set / flush and the threaded ScanSet (recording split by ranges of a
    plane's records, optimisation by (coder, k)) give the oracle's 288-byte table and the SAME doubles for the nine
    "Result size" lines -- the order of every floating-point sum is kept.  synth-text 1 MiB is the
    vector whose .bcc hash SURVEY 8c records."""
    import hashlib
    data = getattr(oracle, gen)(seed, n)
    cfg, res = oracle.scan(data)
    bwt, off = oracle.bwt_stage(data)
    syms = np.ascontiguousarray(oracle.trace_encode_from_bwt(bwt, off)["syms"], dtype=np.uint32)   # plane, s, k, c1, c2, cs
    if gen == "synth_text" and n == 1 << 20:
        h = hashlib.sha256(cfg).hexdigest()
        assert h.startswith("7f7c243b") and h.endswith("4cea2f")
    # (threads, buffers, shortest range): the last makes consume() cut each plane's records into up to 16 ranges recorded
    # by different threads even on these small inputs (by default a range has at least 32768 records)
    for threads, chunks, min_range in ((0, 1, None), (1, 1, None), (5, 3, None), (16, 7, None), (16, 1, 50), (7, 5, 300), (3, 2, 1)):
        out = np.zeros(288, dtype=np.uint8)
        r9 = np.zeros(9, dtype=np.float64)
        if min_range is not None:
            os.environ["BCE_HIP_SCAN_MIN_RANGE"] = str(min_range)
        try:
            emul.emul_scan(syms.ctypes.data, len(syms), threads, chunks, out.ctypes.data, r9.ctypes.data)
        finally:
            os.environ.pop("BCE_HIP_SCAN_MIN_RANGE", None)
        assert out.tobytes() == cfg, (threads, chunks, min_range)
        assert list(r9) == res, (threads, chunks, min_range)


def test_repeated_addition_equals_the_loop(emul):
    """finish() adds log(2) once per escape (bce.cpp:739) -- several 10^8 times on binary data.  scan_add_repeated takes
    the additions a binade at a time; the double must be the loop's, for any start, addend and count."""
    import math
    import random
    emul.emul_add_repeated.argtypes = [C.c_double, C.c_double, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    rng = random.Random(7)
    cases = [(0.0, math.log(2), m) for m in (0, 1, 2, 3, 100, 12345, 1 << 20, 3_000_001, 40_000_000, 600_000_000)]
    cases += [(0.0, 0.5, 1 << 22), (0.0, 0.75, 3_000_000), (1.0, 2.0 ** -53, 5000), (1.0, 2.0 ** -54, 5000),
              (1.0, 3 * 2.0 ** -54, 100_000), (0.0, 1.5 * 2.0 ** -30, 1 << 22), (123.456, 1e-3, 2_000_000)]
    for _ in range(60):
        c = rng.choice([math.log(2), rng.random(), rng.random() * 1e-6, rng.uniform(1, 1e6), math.ldexp(rng.randrange(1, 1 << 53), -rng.randrange(40, 70))])
        z = rng.choice([0.0, rng.random(), rng.uniform(0, 1e9), math.ldexp(1.0, rng.randrange(-3, 30))])
        cases.append((z, c, rng.randrange(0, 2_000_000)))
    for z, c, m in cases:
        f, s = C.c_double(), C.c_double()
        ok = emul.emul_add_repeated(z, c, m, C.byref(f), C.byref(s))
        assert ok and f.value == s.value, (z, c, m, f.value, s.value)


def test_scan_coder_under_address_and_ub_sanitizers(tmp_path):
    """scan_coder.cpp keeps its streams, maps, map nodes and buckets in an arena of its own (placement new, a bump
    allocator, nothing freed singly) and records from many threads: the sequential and the threaded paths under ASan +
    UBSan must be clean and give the oracle's table."""
    asan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.exists(asan):
        pytest.skip("no libasan")
    lib = str(tmp_path / "libcore_emul_san.so")
    srcs = [os.path.join(ROOT, "tests", "core_emul.cpp"), os.path.join(ROOT, "bce_amd", "csrc", "host_coder.cpp"),
            os.path.join(ROOT, "bce_amd", "csrc", "scan_coder.cpp")]
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                           "-o", lib] + srcs + ["-lpthread"])
    code = r'''
import ctypes as C, os, sys
sys.path.insert(0, %r)
import numpy as np, oracle
L = C.CDLL(%r)
L.emul_scan.argtypes = [C.c_void_p, C.c_size_t, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p]
for gen, seed, n in (("synth_text", 1, 200000), ("synth_rand", 5, 50000)):
    data = getattr(oracle, gen)(seed, n)
    cfg, res = oracle.scan(data)
    bwt, off = oracle.bwt_stage(data)
    syms = np.ascontiguousarray(oracle.trace_encode_from_bwt(bwt, off)["syms"], dtype=np.uint32)
    for threads, chunks, mr in ((0, 1, None), (5, 3, None), (16, 1, 50), (3, 2, 1)):
        if mr is not None:
            os.environ["BCE_HIP_SCAN_MIN_RANGE"] = str(mr)
        out = np.zeros(288, dtype=np.uint8); r9 = np.zeros(9)
        L.emul_scan(syms.ctypes.data, len(syms), threads, chunks, out.ctypes.data, r9.ctypes.data)
        os.environ.pop("BCE_HIP_SCAN_MIN_RANGE", None)
        assert out.tobytes() == cfg and list(r9) == res, (gen, threads, chunks, mr)
print("SANITIZED_SCAN_OK")
''' % (ROOT, lib)
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="halt_on_error=1:exitcode=98")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "SANITIZED_SCAN_OK" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
