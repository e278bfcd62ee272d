"""GPU: size-independent properties at sizes the oracle is too slow for (BASELINE configs)."""
import hashlib

import numpy as np
import pytest
import torch

import bce_amd

pytestmark = pytest.mark.gpu


def dev_input(arr):
    t = torch.from_numpy(arr).to("cuda:0")
    torch.cuda.synchronize()
    return t


@pytest.mark.parametrize("gen,n", [("synth_text", 32 << 20), ("synth_rand", 4 << 20)])
def test_full_pipeline_properties(gen, n):
    data = getattr(bce_amd, gen)(1, n)
    t = dev_input(data)
    arch1, st1 = bce_amd.compress_device(t.data_ptr(), n)
    # every internal node of the 8 binary tries is visited exactly once (primitive input): 8n - 8
    assert st1["nodes"] == 8 * n - 8
    assert 0 < st1["symbols"] <= st1["nodes"]
    # BWT is a permutation of the input and offset names a minimal rotation
    rf = bce_amd.RankFile(n=n, device_ptr=t.data_ptr(), build=False)
    try:
        bwt = rf.bwt()
        assert (np.bincount(bwt, minlength=256) == np.bincount(data, minlength=256)).all()
        off = rf.offset()
        rot = np.concatenate([data[off:], data[:off]])[:4096].tobytes()
        rs = np.random.RandomState(0)
        for q in rs.randint(0, n, 200):
            other = np.concatenate([data[q:q + 4096], data[:max(0, q + 4096 - n)]])[:4096].tobytes()
            assert rot <= other
    finally:
        rf.close()
    # determinism + independence from the flush granularity (model state carried across K4 flushes)
    c = bce_amd.api._Ctx(0)
    try:
        rf2 = bce_amd.RankFile(n=n, device_ptr=t.data_ptr(), ctx=c)
        # (one round must fit in the buffer; random data puts ~1.6n symbols in its widest round)
        arch2 = bce_amd.BCE(symbol_capacity=max(1 << 20, st1["symbols"] // 3)).encode(rf2)
        assert bce_amd.stats(rf2)["flushes"] >= 2
    finally:
        c.close()
    assert hashlib.sha256(arch1).hexdigest() == hashlib.sha256(arch2).hexdigest()
    # encode -> decode round trip at a size the oracle is too slow for: GPU-assisted decoder, and the host decoder
    # where it finishes in seconds
    assert bce_amd.decompress_device(arch1) == data.tobytes()
    if n <= (8 << 20):
        assert bce_amd.decompress(arch1) == data.tobytes()


# BASELINE.json configs[2] size (enwik9 = 10^9 B; synth-text stand-in).  The known answer is the ORACLE's (round 3:
# tools/make_oracle_golden.py ran oracle/bce_oracle.c on this input once in the build container -- 587 s, ~13 GB --
# tests/golden/oracle_fullsize.json, synth-text-1e9).  It is the only end-to-end check of the bits = 3, 4 wraps of
# get_context (bce.cpp:674: c1 >= 2^29 / 2^28 only happens at this size) and of getv's 31-bit path (:374).  The round trip
# through the GPU-assisted decoder stays as the property.
def test_enwik9_size_equals_the_oracle_and_round_trips():
    from conftest import load_fullsize_golden
    v = load_fullsize_golden()["synth-text-1e9"]
    n = v["n"]
    assert n == 1_000_000_000 and n > (1 << 29)
    data = bce_amd.synth_text(1, n)
    assert hashlib.sha256(data.tobytes()).hexdigest() == v["input_sha256"]
    t = dev_input(data)
    arch, st = bce_amd.compress_device(t.data_ptr(), n)
    del t
    torch.cuda.empty_cache()
    assert st["nodes"] == 8 * n - 8
    assert len(arch) == v["archive_bytes"] and hashlib.sha256(arch).hexdigest() == v["archive_sha256"]
    back = bce_amd.decompress_device(arch)
    assert len(back) == n and hashlib.sha256(back).hexdigest() == v["input_sha256"]
