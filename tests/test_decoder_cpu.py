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


@pytest.fixture(scope="module")
def asan_cli():
    """The CLI + host decoder + host coder as a CPU-only binary under AddressSanitizer and UBSan (the GPU entry points are
    "no device" stubs, tests/asan_stubs.cpp): what `bce -ds` does with untrusted bytes, with every heap access checked."""
    out = os.path.join(ROOT, "tests", "_build", "bce_asan")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    src = [os.path.join(ROOT, "bce_amd", "csrc", f) for f in ("main.cpp", "decoder.cpp", "host_coder.cpp")] + [os.path.join(ROOT, "tests", "asan_stubs.cpp")]
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-o", out] + src + ["-lpthread"])
    return out


def test_sanitized_ds_on_hostile_containers_and_corrupt_archives(asan_cli, tmp_path):
    """Wrapping / oversized / lying container tables (a table of 2^64 - 100 and 200 bytes made the old code write a
    block into a 100-byte buffer) and bit-flipped or truncated archives through `bce -ds` under ASan + UBSan: a clean exit
    code every time, the right bytes when the archive is intact, no sanitizer report."""
    import struct
    from bce_amd import container
    env = dict(os.environ, ASAN_OPTIONS="abort_on_error=0:detect_leaks=0:exitcode=99", UBSAN_OPTIONS="halt_on_error=1:exitcode=98")
    parts = [oracle.synth_text(31, 4000), oracle.synth_text(32, 300), b"a" * 50]
    archives = [oracle.compress(p) for p in parts]
    out = tmp_path / "o"

    def run(path):
        out.unlink(missing_ok=True)
        r = subprocess.run([asan_cli, "-ds", str(out), str(path)], capture_output=True, text=True, env=env)
        assert r.returncode not in (98, 99) and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
        return r

    good = tmp_path / "good.bcem"
    good.write_bytes(container.pack_blocks(archives, [len(p) for p in parts]))
    r = run(good)
    assert r.returncode == 0 and out.read_bytes() == b"".join(parts)
    sizes = [len(p) for p in parts]
    for t in ([2**64 - 100, 200, 50], [4000, 2**64 - 4000, 50], [2**31, 300, 50], [2**40, 300, 50], [0, 300, 50], [4001, 300, 50],
              [4000, 300, 2**63]):
        bad = tmp_path / "bad.bcem"
        bad.write_bytes(container.pack_blocks(archives, t))
        r = run(bad)
        assert r.returncode == 254 and "Could not read Archive." in r.stdout and not out.exists(), (t, r.returncode, r.stdout)
    # a table whose ARCHIVE sizes lie (beyond the file, wrapping)
    blob = bytearray(container.pack_blocks(archives, sizes))
    for alen in (2**64 - 1, len(blob), 2**40):
        b2 = bytearray(blob)
        b2[12 + 8:12 + 16] = struct.pack("<Q", alen)
        bad = tmp_path / "bad2.bcem"
        bad.write_bytes(bytes(b2))
        r = run(bad)
        assert r.returncode != 0 and not out.exists()
    # single archives: truncated and bit-flipped -- any exit code, never a sanitizer report
    rs = np.random.RandomState(5)
    single = archives[0]
    arc = tmp_path / "a.bce"
    arc.write_bytes(single)
    r = run(arc)
    assert r.returncode == 0 and out.read_bytes() == parts[0]
    for _ in range(40):
        b = bytearray(single)
        for _k in range(int(rs.randint(1, 4))):
            b[rs.randint(len(b))] ^= 1 << rs.randint(8)
        arc.write_bytes(bytes(b[: len(b) - 2 * int(rs.randint(0, 5))]))
        run(arc)
