"""CPU: the host decoder (bce_amd/csrc/decoder.cpp, `bce -d`) against archives of the oracle and the reference."""
import os
import subprocess

import numpy as np
import pytest

import bce_amd
import oracle
from conftest import ROOT, edge_inputs, golden_input, load_golden


@pytest.mark.parametrize("name,data", edge_inputs(), ids=[n for n, _ in edge_inputs()])
def test_decode_roundtrip_edges(name, data):
    """encode -> decode round trip, including the periodic inputs on which the reference decoder fails (Q9)."""
    assert bce_amd.decompress(oracle.compress(data)) == bytes(data)


@pytest.mark.parametrize("gen,seed,n", [("synth_text", 1, 65536), ("synth_rand", 1, 65536), ("synth_text", 3, 1 << 20)])
def test_decode_roundtrip_synth(gen, seed, n):
    data = getattr(oracle, gen)(seed, n)
    assert bce_amd.decompress(oracle.compress(data)) == data


def test_decode_reference_archive_and_custom_config():
    v = [x for x in load_golden() if x["name"] == "abracadabra"][0]
    assert bce_amd.decompress(bytes.fromhex(v["archive_hex"])) == b"abracadabra"     # the reference's own bytes
    cfg = np.random.RandomState(9).randint(0, 6, 288).astype(np.uint8).tobytes()
    data = oracle.synth_text(8, 40000)
    assert bce_amd.decompress(oracle.compress(data, cfg)) == data                    # archives are self-describing


def test_decode_rejects_garbage_without_crashing():
    good = oracle.compress(oracle.synth_text(2, 5000))
    for bad in (b"", b"\x00", b"\x01\x00", good[:7], b"\xff" * 64):
        with pytest.raises(bce_amd.BceError):
            bce_amd.decompress(bad)
    # truncated / bit-flipped archives must return (an error or wrong bytes), never crash
    rs = np.random.RandomState(1)
    for _ in range(30):
        b = bytearray(good)
        b[rs.randint(len(b))] ^= 1 << rs.randint(8)
        try:
            bce_amd.decompress(bytes(b[: len(b) - 2 * rs.randint(0, 3)]))
        except bce_amd.BceError:
            pass


def test_cli_decompress(tmp_path):
    data = oracle.synth_text(4, 20000)
    arc, out = tmp_path / "a.bce", tmp_path / "out.txt"
    arc.write_bytes(oracle.compress(data))
    exe = os.path.join(ROOT, "bce_amd", "bin", "bce")
    # -ds = the host decoder by name: works without a GPU
    r = subprocess.run([exe, "-ds", str(out), str(arc)], capture_output=True, text=True)
    assert r.returncode == 0 and "Decompressed from %d B -> 20000 B in " % arc.stat().st_size in r.stdout
    assert out.read_bytes() == data
    r = subprocess.run([exe, "-ds", str(out), str(tmp_path / "missing")], capture_output=True, text=True)
    assert r.returncode == 255 and "Archive not found." in r.stdout
    # -d = the GPU-assisted decoder: on a machine without a GPU it fails loudly, it never falls back to the host
    import torch
    if not torch.cuda.is_available():
        out.unlink()
        r = subprocess.run([exe, "-d", str(out), str(arc)], capture_output=True, text=True)
        assert r.returncode == 253 and "No usable HIP device" in r.stdout and not out.exists()


def test_cli_container_with_hostile_size_table(tmp_path):
    """A BCEM container's raw sizes are untrusted: wrapping (2^64 - k), oversized (>= 2^31), zero, or merely wrong values
    must end in a clean "Could not read Archive." (exit -2), never in a write outside the output or an abort
    (main.cpp, -d branch).  -ds runs on the host, so this needs no GPU."""
    from bce_amd import container
    exe = os.path.join(ROOT, "bce_amd", "bin", "bce")
    parts = [oracle.synth_text(21, 3000), oracle.synth_text(22, 200)]
    archives = [oracle.compress(p) for p in parts]
    out = tmp_path / "o"
    good = tmp_path / "good.bcem"
    good.write_bytes(container.pack_blocks(archives, [len(p) for p in parts]))
    r = subprocess.run([exe, "-ds", str(out), str(good)], capture_output=True, text=True)
    assert r.returncode == 0 and out.read_bytes() == b"".join(parts)
    tables = [
        [2**64 - 100, 200],            # the sum wraps to 100: block 0 would be written into a 100-byte buffer
        [2**64 - 3000 + 200 - 200, 200],
        [3000, 2**64 - 3000],          # sum wraps to 0
        [2**31, 200],                  # above the encoder's limit
        [2**40, 200],                  # resize would throw
        [0, 200],
        [3001, 200], [2999, 200], [3000, 201],
    ]
    for t in tables:
        out.unlink(missing_ok=True)
        bad = tmp_path / "bad.bcem"
        bad.write_bytes(container.pack_blocks(archives, t))
        r = subprocess.run([exe, "-ds", str(out), str(bad)], capture_output=True, text=True)
        assert r.returncode == 254 and "Could not read Archive." in r.stdout, (t, r.returncode, r.stdout, r.stderr)
        assert not out.exists()
