"""CPU: libbcehip.so loads and exports every symbol include/bce_hip.h declares (no compute calls)."""
import ctypes as C
import os
import re

import pytest

import bce_amd
from bce_amd import api
from conftest import ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "bce_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bce_hip_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(bce_amd.library_path())
    names = declared_symbols()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), name
    bound = {n for n, _, _ in api.SYMBOLS}
    assert bound == set(names)


def test_stats_struct_of_the_binding_is_the_headers():
    """bce_hip_stats (include/bce_hip.h) field by field == the ctypes structure the binding reads it with: names, order, C types."""
    src = open(os.path.join(ROOT, "include", "bce_hip.h")).read()
    body = src[src.index("typedef struct bce_hip_stats {"):src.index("} bce_hip_stats;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    want = []
    for ctype, names in re.findall(r"\b(uint64_t|uint32_t|double)\s+([^;]+);", body):
        for name in names.split(","):
            want.append((name.strip(), {"uint64_t": C.c_uint64, "uint32_t": C.c_uint32, "double": C.c_double}[ctype]))
    assert len(want) >= 19
    assert [(n, t) for n, t in api.Stats._fields_] == want


def test_divsufsort_seam_library_exports_the_two_reference_calls():
    """include/divsufsort_hip.h: the libdivsufsort entry points bce.cpp:901 / :1091 call, under their original names."""
    src = open(os.path.join(ROOT, "include", "divsufsort_hip.h")).read()
    assert "saidx_t divbwt(const sauchar_t *T, sauchar_t *U, saidx_t *A, saidx_t n);" in src
    assert "saint_t inverse_bw_transform(const sauchar_t *T, sauchar_t *U, saidx_t *A, saidx_t n, saidx_t idx);" in src
    lib = C.CDLL(os.path.join(ROOT, "bce_amd", "lib", "libdivsufsort_hip.so"))
    assert hasattr(lib, "divbwt") and hasattr(lib, "inverse_bw_transform")
    lib.divbwt.restype = C.c_int32
    assert lib.divbwt(None, None, None, 4) == -1           # argument check only: no GPU touched


def test_strerror_and_generators_without_gpu():
    lib = bce_amd.load_library()
    assert lib.bce_hip_strerror(0) == b"ok"
    assert lib.bce_hip_strerror(-5) == b"buffer capacity exceeded"
    import hashlib
    assert hashlib.sha256(bce_amd.synth_rand(1, 65536).tobytes()).hexdigest() == \
        "9fce4bb9a03c0ca0a85425e43a1a8e56056b7cf432d881db380c536ff81eb11e"
    assert hashlib.sha256(bce_amd.synth_text(1, 1 << 20).tobytes()).hexdigest() == \
        "e244773e9f70e60d229868f960128fc6ccdfd29bddecad5395edb549ea2bbd61"


def test_no_cpu_fallback_when_gpu_missing():
    """Without a GPU every compute entry point must fail loudly (never route to a CPU path)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(bce_amd.BceError):
        bce_amd.compress(b"abracadabra")


def test_cli_usage_matches_reference_text():
    import subprocess
    exe = os.path.join(ROOT, "bce_amd", "bin", "bce")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0
    assert r.stdout.startswith("BCE v0.4 Release\nCopyright (C) 2016  Christoph Diegelmann\n")
    assert "  bce -c archive.bce file [config.bcc]\n" in r.stdout
