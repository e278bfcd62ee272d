"""CPU: the oracle against the reference's recorded outputs -- this is what pins it."""
import hashlib

import pytest

import oracle
from conftest import golden_input, load_golden


@pytest.mark.parametrize("v", load_golden(), ids=lambda v: v["name"])
def test_oracle_matches_reference_archive(v):
    d = golden_input(v)
    a = oracle.compress(d)
    assert len(a) == v["archive_bytes"]
    assert hashlib.sha256(a).hexdigest() == v["archive_sha256"]
    if "archive_hex" in v:
        assert a.hex() == v["archive_hex"]


def test_oracle_abracadabra_stages():
    v = [x for x in load_golden() if x["name"] == "abracadabra"][0]
    bwt, off = oracle.bwt_stage(b"abracadabra")
    assert bytes(bwt) == v["bwt"].encode() and off == v["offset"]
    tr = oracle.trace_encode_from_bwt(bwt, off)
    assert len(tr["nodes"]) == v["nodes"] and len(tr["syms"]) == v["symbols"] and tr["rounds"] == v["rounds"]
    # first traced nodes of SURVEY section 0: (round, plane, s, x0, x1)
    assert tr["nodes"][0].tolist() == [0, 1, 0, 5, 6]
    assert tr["archive"].hex() == v["archive_hex"]


def test_oracle_bwt_is_cyclic_rotation_bwt():
    """File::rotate + File::bwt == BWT of all cyclic rotations, offset = first minimal rotation."""
    import random
    rnd = random.Random(1)
    for _ in range(200):
        n = rnd.randint(1, 40)
        s = bytes(rnd.choice(b"ab" if rnd.random() < 0.5 else b"abcd") for _ in range(n))
        rots = sorted(range(n), key=lambda i: (s[i:] + s[:i], i))
        expect = bytes(s[(i - 1) % n] for i in rots)
        mn = min(s[i:] + s[:i] for i in range(n))
        first = min(i for i in range(n) if s[i:] + s[:i] == mn)
        bwt, off = oracle.bwt_stage(s)
        assert bytes(bwt) == expect and off == first, s


def test_oracle_rejects_empty():
    with pytest.raises(ValueError):
        oracle.compress(b"")


def test_oracle_scan_matches_reference_config():
    """`bce -s` (ScanCoder, Q11): the .bcc and the archive compressed with it, against SURVEY 8c."""
    import json, os
    from conftest import ROOT
    v = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))["scan_vectors"][0]
    d = oracle.synth_text(v["seed"], v["n"])
    assert hashlib.sha256(d).hexdigest() == v["input_sha256"]
    cfg, res = oracle.scan(d)
    h = hashlib.sha256(cfg).hexdigest()
    assert h.startswith(v["config_sha256_prefix"]) and h.endswith(v["config_sha256_suffix"])
    a = oracle.compress(d, cfg)
    assert len(a) == v["archive_bytes"] and hashlib.sha256(a).hexdigest() == v["archive_sha256"]
    assert len(res) == 9 and res[8] == 0.0


def test_openmp_round_loop_gives_the_same_archive():
    """The 8-thread round loop (the reference's OpenMP build, bce.cpp:1250-1252) is only a schedule."""
    data = oracle.synth_text(9, 300000) + oracle.synth_rand(9, 20000)
    try:
        oracle.set_threads(1)
        a1 = oracle.compress(data)
        oracle.set_threads(8)
        a8 = oracle.compress(data)
    finally:
        oracle.set_threads(1)
    assert a1 == a8


def test_fullsize_golden_file_is_oracle_made_and_reproducible_at_16mib():
    """tests/golden/oracle_fullsize.json (tools/make_oracle_golden.py): the vectors the -m gpu tests pin BASELINE-size
    archives to.  Here, without a GPU: the file is complete, and the oracle reproduces its 16 MiB corpus vectors (the
    10^8-byte ones take 40-65 s each and are left to the tool)."""
    import hashlib

    from conftest import fullsize_input, load_fullsize_golden
    gold = load_fullsize_golden()
    for name in ("synth-text-1e8", "synth-text-1.5e8", "synth-rand-32Mi", "natural-1e8", "binary-1e8"):
        v = gold[name]
        assert len(v["archive_sha256"]) == 64 and v["archive_bytes"] > 0 and v["oracle_seconds_1_thread"] > 0
    assert gold["synth-text-1.5e8"]["n"] > (1 << 27)
    import numpy as np
    checked = 0
    for name in ("natural-16Mi", "binary-16Mi"):
        v = gold[name]
        data = fullsize_input(v)
        if data is None:
            continue                      # another image's files: nothing to compare on this box
        arch = oracle.compress(np.ascontiguousarray(data))
        assert len(arch) == v["archive_bytes"] and hashlib.sha256(arch).hexdigest() == v["archive_sha256"], name
        checked += 1
    print("16 MiB corpus vectors reproduced by the oracle: %d of 2" % checked)
