"""CPU: the per-element arithmetic shared with the HIP kernels (bce_amd/csrc/bce_core.h) and the host
range coder / framing (host_coder.cpp), driven sequentially by tests/core_emul.cpp, against the oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle
from conftest import ROOT, edge_inputs


@pytest.fixture(scope="module")
def emul():
    out = os.path.join(ROOT, "tests", "_build", "libcore_emul.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    srcs = [os.path.join(ROOT, "tests", "core_emul.cpp"), os.path.join(ROOT, "bce_amd", "csrc", "host_coder.cpp")]
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", out] + srcs + ["-lpthread"])
    L = C.CDLL(out)
    L.emul_encode_from_bwt.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.emul_rank1.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]
    L.emul_free.argtypes = [C.c_void_p]
    L.emul_check_recip.argtypes = [C.c_uint64, C.c_uint64]
    L.emul_check_recip.restype = C.c_uint64
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
