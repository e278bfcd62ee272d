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
    assert dt < 5.0, "chain skip not taken? encode took %.1f s" % dt
    assert bce_amd.decompress(arch) == data
