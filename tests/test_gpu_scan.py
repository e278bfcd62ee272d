"""GPU: `bce -s`: enumeration on the GPU in scan mode + host ScanCoder, against the oracle's
restatement of ScanCoder<31> (bce.cpp:726-834) and the reference's recorded result."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import bce_amd
import oracle
from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("gen,seed,n", [("synth_text", 2, 50000), ("synth_rand", 3, 30000), ("synth_text", 1, 1 << 20)])
def test_scan_config_matches_oracle(gen, seed, n):
    data = getattr(oracle, gen)(seed, n)
    cfg, res = bce_amd.scan(data)
    ocfg, ores = oracle.scan(data)
    assert cfg == ocfg
    assert np.allclose(res, ores, rtol=0, atol=0)          # same additions in the same order: identical doubles


def test_scan_then_compress_matches_reference_golden():
    v = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))["scan_vectors"][0]
    d = oracle.synth_text(v["seed"], v["n"])
    cfg, _ = bce_amd.scan(d)
    h = hashlib.sha256(cfg).hexdigest()
    assert h.startswith(v["config_sha256_prefix"]) and h.endswith(v["config_sha256_suffix"])
    a = bce_amd.compress(d, cfg)
    assert len(a) == v["archive_bytes"] and hashlib.sha256(a).hexdigest() == v["archive_sha256"]
    assert bce_amd.decompress(a) == d


def test_cli_scan(tmp_path):
    data = oracle.synth_text(6, 80000)
    src, cf = tmp_path / "in.txt", tmp_path / "c.bcc"
    src.write_bytes(data)
    exe = os.path.join(ROOT, "bce_amd", "bin", "bce")
    r = subprocess.run([exe, "-s", str(cf), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    assert r.stdout.count("Result size: ") == 9 and "Scanned 80000 B in " in r.stdout
    assert cf.read_bytes() == oracle.scan(data)[0]
