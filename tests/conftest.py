import hashlib
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "perf: wall-clock limits (kept out of the parity tests; generous, a shared box may still miss them)")


def load_fullsize_golden():
    """Oracle-produced archives at BASELINE sizes (tools/make_oracle_golden.py): name -> vector."""
    with open(os.path.join(ROOT, "tests", "golden", "oracle_fullsize.json")) as f:
        return {v["name"]: v for v in json.load(f)["vectors"]}


def fullsize_input(v):
    """The input of a full-size vector as np.uint8, or None when this box cannot rebuild it (another image's files)."""
    import subprocess

    import numpy as np
    if v["kind"] in ("synth_text", "synth_rand"):
        import bce_amd
        data = getattr(bce_amd, v["kind"])(v["seed"], v["n"])
    elif v["kind"] == "mixed":      # natural corpus || binary corpus, half each (tools/make_oracle_golden.py)
        halves = [fullsize_input({"kind": k, "n": m, "input_sha256": None}) for k, m in (("natural", v["n"] // 2), ("binary", v["n"] - v["n"] // 2))]
        if any(h is None for h in halves):
            return None
        data = np.concatenate(halves)
    else:
        path = "/tmp/bce_%s_%d.bin" % (v["kind"], v["n"])
        if not (os.path.exists(path) and os.path.getsize(path) == v["n"]):
            tool = "make_corpus.py" if v["kind"] == "natural" else "make_binary_corpus.py"
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), "--out", path, "--size", str(v["n"])],
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            if r.returncode != 0:
                return None
        data = np.fromfile(path, dtype=np.uint8)
    if v["input_sha256"] is not None and hashlib.sha256(data.tobytes()).hexdigest() != v["input_sha256"]:
        return None
    return data


def load_golden():
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
        return json.load(f)["vectors"]


def golden_input(v):
    import oracle
    if v["gen"] == "literal":
        return v["literal"].encode()
    d = getattr(oracle, v["gen"])(v["seed"], v["n"])
    assert hashlib.sha256(d).hexdigest() == v["input_sha256"]
    return d


@pytest.fixture(scope="session")
def golden():
    return load_golden()


def edge_inputs():
    """Small inputs covering the edge cases of the path (ragged sizes, periodic, constant planes)."""
    import oracle
    return [
        ("one-byte", b"a"),
        ("two-bytes", b"ab"),
        ("two-equal", b"aa"),
        ("three", b"abc"),
        ("constant-100", b"a" * 100),             # periodic, no roots at all
        ("period2-100", b"ab" * 50),              # periodic (Q8/Q9)
        ("period3-99", b"abc" * 33),
        ("abracadabra", b"abracadabra"),
        ("len-95", oracle.synth_text(5, 95)),     # granule boundaries (96) and chunk boundaries (3072, 2048)
        ("len-96", oracle.synth_text(5, 96)),
        ("len-97", oracle.synth_text(5, 97)),
        ("len-2047", oracle.synth_text(6, 2047)),
        ("len-2048", oracle.synth_rand(6, 2048)),
        ("len-3071", oracle.synth_text(7, 3071)),
        ("len-3072", oracle.synth_rand(7, 3072)),
        ("len-3073", oracle.synth_text(7, 3073)),
        ("all-bytes", bytes(range(256)) * 3),
        ("binary-zeros-tail", oracle.synth_rand(8, 5000) + b"\x00" * 3000),
        ("long-repeat", oracle.synth_text(9, 20000) + oracle.synth_text(9, 6000) + oracle.synth_text(10, 3000)),
    ]
