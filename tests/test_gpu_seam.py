"""GPU: the libdivsufsort seam.  libdivsufsort_hip.so exports divbwt / inverse_bw_transform with the library's own C
signatures (include/divsufsort_hip.h) -- the two calls of the reference, bce.cpp:901 and :1091 -- so that an unmodified
bce.cpp links against it.  Checked through ctypes against the oracle's restatement of the same contract (sentinel suffix
order, U[0] = T[n-1], returned primary index) and against a brute-force suffix sort on small strings."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle
from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def seam():
    L = C.CDLL(os.path.join(ROOT, "bce_amd", "lib", "libdivsufsort_hip.so"))
    L.divbwt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    L.divbwt.restype = C.c_int32
    L.inverse_bw_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
    L.inverse_bw_transform.restype = C.c_int32
    return L


def gpu_divbwt(L, t, in_place=False):
    a = np.frombuffer(bytes(t), dtype=np.uint8).copy()
    u = a if in_place else np.empty_like(a)
    p = L.divbwt(a.ctypes.data, u.ctypes.data, None, len(a))          # A = NULL, as bce.cpp:901 passes
    return u.tobytes(), p


def gpu_inverse(L, u, idx, in_place=False):
    a = np.frombuffer(bytes(u), dtype=np.uint8).copy()
    o = a if in_place else np.empty_like(a)
    rc = L.inverse_bw_transform(a.ctypes.data, o.ctypes.data, None, len(a), idx)
    return o.tobytes(), rc


def brute(t):
    n = len(t)
    sa = sorted(range(n), key=lambda i: t[i:])        # implicit smallest sentinel = plain suffix order
    u, pidx = [t[n - 1]], None
    for i, p in enumerate(sa):
        if p == 0:
            pidx = i + 1
        else:
            u.append(t[p - 1])
    return bytes(u), pidx


def test_small_strings_against_brute_force_and_oracle(seam):
    rs = np.random.RandomState(5)
    cases = [b"a", b"ab", b"ba", b"aa", b"abracadabra", b"banana", b"\x00\x00\x01\x00", b"\xff" * 7, b"ab" * 9, bytes(range(256))]
    for _ in range(150):
        n = int(rs.randint(1, 60))
        cases.append(bytes(rs.choice([0, 1, 97, 98, 255], n).astype(np.uint8)))
    for t in cases:
        u, p = gpu_divbwt(seam, t)
        assert (u, p) == brute(t), t
        assert (u, p) == oracle.divbwt(t)
        back, rc = gpu_inverse(seam, u, p)
        assert rc == 0 and back == t


@pytest.mark.parametrize("gen,seed,n", [("synth_text", 3, 1 << 20), ("synth_rand", 4, 300000), ("synth_text", 9, 3_000_001)])
def test_matches_oracle_divbwt_and_inverts_in_place(seam, gen, seed, n):
    t = getattr(oracle, gen)(seed, n)
    want_u, want_p = oracle.divbwt(t)
    u, p = gpu_divbwt(seam, t, in_place=True)          # bce calls it in place: divbwt(p, p, 0, n - 1)
    assert p == want_p and u == want_u
    back, rc = gpu_inverse(seam, u, p, in_place=True)  # ... and inverse_bw_transform(p, p, nullptr, n, idx)
    assert rc == 0 and back == t
    assert oracle.inverse_bwt(u, p) == t


def test_repetitive_and_periodic_inputs(seam):
    """Suffixes of a periodic string are all distinct once the sentinel ends them: long ties for the doubling rounds."""
    for t in (b"a" * 5000, b"ab" * 4000, b"abc" * 3333 + b"ab", oracle.synth_text(2, 20000) * 6):
        u, p = gpu_divbwt(seam, t)
        assert (u, p) == oracle.divbwt(t)
        back, rc = gpu_inverse(seam, u, p)
        assert rc == 0 and back == t


def test_the_references_own_use_of_the_seam(seam):
    """File::bwt (bce.cpp:896-910) = divbwt on the first n-1 bytes of the rotated file, last byte moved to slot pidx;
    with the seam under it the result must be the BWT stage the oracle computes (and the GPU path's own K1)."""
    data = oracle.synth_text(11, 200000)
    want_bwt, off = oracle.bwt_stage(data)
    rot = data[off + 1:] + data[:off + 1]               # std::rotate(begin, begin + i + 1, end), :884
    u, pidx = gpu_divbwt(seam, rot[:-1])
    bwt = u[:pidx] + rot[-1:] + u[pidx:]                # :902-909
    assert bwt == want_bwt.tobytes()


def test_bad_arguments(seam):
    a = np.zeros(8, dtype=np.uint8)
    assert seam.divbwt(None, a.ctypes.data, None, 8) == -1
    assert seam.divbwt(a.ctypes.data, a.ctypes.data, None, -1) == -1
    assert seam.divbwt(a.ctypes.data, a.ctypes.data, None, 0) == 0
    assert seam.inverse_bw_transform(a.ctypes.data, a.ctypes.data, None, 8, 0) == -1
    assert seam.inverse_bw_transform(a.ctypes.data, a.ctypes.data, None, 8, 9) == -1
    # bytes that are no BWT for this index: several LF cycles
    bad = np.frombuffer(b"abababab", dtype=np.uint8).copy()
    out = np.empty_like(bad)
    assert seam.inverse_bw_transform(bad.ctypes.data, out.ctypes.data, None, 8, 1) in (-1, 0)
