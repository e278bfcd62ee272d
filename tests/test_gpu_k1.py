"""GPU: K1 (rotation sort + BWT, k1_bwt.hip) on inputs aimed at its round-3 machinery: the compacted alphabet and the
64-bit first key, groups around the 2048-element limit of the in-LDS sort (2047 / 2048 / 2049 / larger: the deferred
path), list lengths around the 2048-element chunks a workgroup owns, periodic inputs that never separate, and the old
sorter (BCE_K1_V1) as a second opinion.  Everything against File::rotate + File::bwt of the oracle (bce.cpp:858-910)."""
import os

import numpy as np
import pytest

import bce_amd
import oracle

pytestmark = pytest.mark.gpu


def k1(data):
    rf = bce_amd.RankFile(data, build=False)
    try:
        return bytes(rf.bwt()), rf.offset()
    finally:
        rf.close()


def check(data):
    data = bytes(data)
    bwt, off = oracle.bwt_stage(data)
    got, goff = k1(data)
    assert goff == off
    assert got == bytes(bwt)


def repeats(unit, count, rng, sep=True):
    """`count` copies of `unit`, each followed by a distinct tail: one group of `count` rotations per position of the unit."""
    out = []
    for i in range(count):
        out.append(unit)
        if sep:
            out.append(b"\xff" + int(i).to_bytes(3, "big") + rng.bytes(3))
    return b"".join(out)


@pytest.mark.parametrize("sigma", [1, 2, 3, 4, 5, 16, 17, 127, 128, 129, 255, 256])
def test_alphabet_sizes(sigma):
    """sigma distinct bytes -> ceil(log2 sigma) bits per symbol, 64 / bits symbols per key (16 at most); sigma = 1 never separates."""
    rng = np.random.RandomState(sigma)
    vals = rng.permutation(256)[:sigma].astype(np.uint8)
    for n in (1, 2, 3, 17, 300, 5000, 70001):
        data = vals[rng.randint(0, sigma, n)]
        if sigma > 1 and n >= sigma:
            data[:sigma] = vals                      # every symbol occurs
        check(data)


@pytest.mark.parametrize("count", [2, 63, 64, 65, 1023, 1024, 1025, 2047, 2048, 2049, 2050, 4095, 4096, 4097, 9000])
def test_group_sizes_around_the_lds_limit(count):
    """Groups of exactly `count` rotations that share 40 bytes (then differ): sorted in LDS up to 2048, deferred to the wide
    sort above; 2047 / 2048 / 2049 sit on the boundary of both rules (group <= SEG_CAP, group starts in the chunk)."""
    rng = np.random.RandomState(count)
    unit = rng.bytes(40)
    check(repeats(unit, count, rng))


def test_many_groups_of_mixed_sizes_and_chunk_boundaries():
    """Groups of 2 .. 5000 members in one input, and list lengths of 2048 k - 1, 2048 k, 2048 k + 1 active elements (a
    workgroup owns the groups that START in its 2048-element chunk; the last chunk is ragged)."""
    rng = np.random.RandomState(7)
    parts = []
    for g in (2, 3, 7, 64, 100, 511, 2048, 2049, 3000, 5000, 2, 2047):
        parts.append(repeats(rng.bytes(int(rng.randint(12, 60))), g, rng))
    check(b"".join(parts))
    for total in (2047, 2048, 2049, 4095, 4096, 4097, 6143, 6145):
        # `total` rotations in non-singleton groups: pairs of equal 24-byte units
        units = [rng.bytes(24) for _ in range(total // 2)]
        body = b"".join(u + b"\xfe" + rng.bytes(4) + u + b"\xfd" + rng.bytes(4) for u in units)
        check(body)


def test_long_runs_and_periodic_inputs():
    """Runs (one giant group that shrinks by one per byte: deferred in every round), period-p inputs (rotations p apart are EQUAL:
    h reaches n with groups left), and the mixture an executable is made of."""
    rng = np.random.RandomState(3)
    check(bytes(100000))
    check(b"ab" * 30000)
    check(b"abcabcabd" * 9000)
    check(bytes(70000) + b"\x01" + bytes(50000) + rng.bytes(1000) + bytes(6000))
    check(b"".join(bytes(int(rng.randint(1, 6000))) + rng.bytes(int(rng.randint(1, 40))) for _ in range(300)))
    check((rng.bytes(4099) * 50)[:200001])


def test_old_sorter_agrees(monkeypatch):
    """BCE_K1_V1=1 (the sorter of rounds 1-2) and the new one on the same inputs: the same BWT and offset."""
    rng = np.random.RandomState(11)
    inputs = [oracle.synth_text(5, 1 << 20), repeats(rng.bytes(50), 3000, rng), bytes(50000) + rng.bytes(50000)]
    new = [k1(d) for d in inputs]
    monkeypatch.setenv("BCE_K1_V1", "1")
    old = [k1(d) for d in inputs]
    assert new == old


@pytest.mark.parametrize("knob", [{"BCE_K1_PART_MIN": "0"}, {"BCE_K1_KEYBITS": "16"}, {"BCE_K1_KEYBITS": "24", "BCE_K1_PART_MIN": "0"}],
                         ids=["sorted-scatters", "narrow-first-key", "both"])
def test_rank_scatters_as_sorts_on_small_inputs(knob):
    """(BCE_K1_KEYBITS: a first key of 16 / 24 bits instead of 64 leaves most of the order to the segmented and active rounds:
    the same checks with many more of those.)  Above 2^27 elements K1 writes its ranks by SORTING the (destination, value) pairs instead of scattering them (10^9-byte
    inputs; k1_bwt.hip).  BCE_K1_PART_MIN=0 takes that form at every size: the same inputs as above -- alphabets, groups around
    the LDS limit (the deferred path's buffers are reused by the sort), periodic inputs -- and the libdivsufsort seam (T$: a
    sentinel as one more symbol) in a child process, each against the oracle."""
    import subprocess
    import sys
    code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
import bce_amd, oracle
from tests.test_gpu_k1 import check, repeats
rng = np.random.RandomState(5)
for n in (2, 3, 300, 70001, 1 << 20):
    check(rng.randint(0, 256, n).astype(np.uint8))
    check(rng.randint(97, 101, n).astype(np.uint8))
for count in (2, 2047, 2048, 2049, 9000):
    check(repeats(rng.bytes(40), count, rng))
check(b"ab" * 5000)
check(b"a" * 4097)
check(oracle.synth_text(3, 3_000_000))
data = oracle.synth_text(4, 1_500_000)
assert bytes(bce_amd.compress(data)) == oracle.compress(data)
# the libdivsufsort seam (T$ with the sentinel as one more symbol goes through the same sorter)
import ctypes as C
L = C.CDLL(os.path.join(%r, "bce_amd", "lib", "libdivsufsort_hip.so"))
L.divbwt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]; L.divbwt.restype = C.c_int32
for t in (b"abracadabra", oracle.synth_text(6, 700001), b"ab" * 4000):
    a = np.frombuffer(bytes(t), dtype=np.uint8).copy(); u = np.empty_like(a)
    pidx = L.divbwt(a.ctypes.data, u.ctypes.data, None, len(a))
    wu, wp = oracle.divbwt(bytes(t))
    assert pidx == wp and u.tobytes() == bytes(wu), len(t)
print("SORTED_SCATTERS_OK")
''' % (os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    env = dict(os.environ, **knob)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env,
                       cwd=os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    assert r.returncode == 0 and "SORTED_SCATTERS_OK" in r.stdout, r.stderr[-3000:]
