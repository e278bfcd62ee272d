"""GPU: natural data found in the image (the interpreter's own source files and ELF binaries) against the oracle.

The synthetic generators of SURVEY 8c use 28 distinct bytes (text) or all 256 uniformly (rand); real files sit in
between: full ASCII plus UTF-8 text, and executables with long zero runs, tables and repeated code sequences, which
drive the tail / depth-first / chain-skipping paths of the enumeration with data nobody designed.  Inputs are read
from the local filesystem at test time (nothing is stored in the repo); the oracle is computed on the same bytes,
so the test does not depend on the exact file versions."""
import glob
import hashlib
import os
import sys

import numpy as np
import pytest

import bce_amd
import oracle

pytestmark = pytest.mark.gpu


def _python_sources(limit):
    root = os.path.dirname(os.__file__)
    out = bytearray()
    for path in sorted(glob.glob(os.path.join(root, "*.py"))):
        with open(path, "rb") as f:
            out += f.read()
        if len(out) >= limit:
            break
    return bytes(out[:limit])


def _binary(limit, skip=0):
    with open(os.path.realpath(sys.executable), "rb") as f:
        f.seek(skip)
        return f.read(limit)


def _cases():
    src = _python_sources(3 << 20)
    elf = _binary(2 << 20)
    mid = _binary(1 << 20, skip=3 << 20)
    cases = [("python-sources-3MiB", src), ("elf-head-2MiB", elf), ("elf-mid-1MiB", mid)]
    if len(src) > (1 << 20) and len(elf) > (1 << 19):
        cases.append(("mixed-src-elf-src", src[:1 << 20] + elf[:1 << 19] + src[:1 << 19]))
    return [(name, data) for name, data in cases if len(data) > 0]


@pytest.mark.parametrize("name", ["python-sources-3MiB", "elf-head-2MiB", "elf-mid-1MiB", "mixed-src-elf-src"])
def test_real_file_archives_match_oracle(name):
    cases = dict(_cases())
    if name not in cases:
        pytest.skip("input files not present in this image")
    data = cases[name]
    want = oracle.compress(data)
    got = bce_amd.compress(data)
    assert len(got) == len(want)
    assert hashlib.sha256(got).hexdigest() == hashlib.sha256(want).hexdigest()
    assert bce_amd.decompress_device(got) == data
    if len(data) <= (1 << 20):
        assert bce_amd.decompress(got) == data


def test_real_file_scan_then_compress_matches_oracle():
    """BASELINE config 5 in miniature: scan a natural input on the GPU, compress with the scanned table."""
    data = _python_sources(1 << 20)
    if len(data) < 1000:
        pytest.skip("input files not present in this image")
    cfg_want, _ = oracle.scan(data)
    cfg_got, _ = bce_amd.scan(data)
    assert bytes(cfg_got) == bytes(cfg_want)
    want = oracle.compress(data, bytes(cfg_want))
    got = bce_amd.compress(data, bytes(cfg_got))
    assert got == want
    assert bce_amd.decompress_device(got) == data
