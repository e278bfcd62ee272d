"""GPU: the `bce` command line (bce_amd/bin/bce) -- `bce -c archive file [config]` as the reference's main() (bce.cpp:1403-1427)."""
import os
import subprocess

import numpy as np
import pytest

import bce_amd
import oracle
from conftest import ROOT

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "bce_amd", "bin", "bce")


def test_cli_compress_matches_oracle(tmp_path):
    data = oracle.synth_text(12, 300000)
    src, dst = tmp_path / "in.txt", tmp_path / "out.bce"
    src.write_bytes(data)
    r = subprocess.run([EXE, "-c", str(dst), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("BCE v0.4 Release\n")
    assert "Compressed from 300000 B -> %d B in " % dst.stat().st_size in r.stdout
    assert dst.read_bytes() == oracle.compress(data)


def test_cli_config_and_errors(tmp_path):
    data = oracle.synth_text(13, 100000)
    cfg = np.random.RandomState(5).randint(0, 6, 288).astype(np.uint8).tobytes()
    src, dst, cf = tmp_path / "in.txt", tmp_path / "out.bce", tmp_path / "c.bcc"
    src.write_bytes(data)
    cf.write_bytes(cfg)
    r = subprocess.run([EXE, "-c", str(dst), str(src), str(cf)], capture_output=True, text=True)
    assert r.returncode == 0
    assert dst.read_bytes() == oracle.compress(data, cfg)
    # wrong-size config: message, defaults used (bce.cpp:629-631)
    cf.write_bytes(b"x" * 10)
    r = subprocess.run([EXE, "-c", str(dst), str(src), str(cf)], capture_output=True, text=True)
    assert r.returncode == 0 and "Config not found or wrong size." in r.stdout
    assert dst.read_bytes() == oracle.compress(data)
    # missing / empty input: "Error loading file", exit -1 (bce.cpp:1412-1415; empty input crashes the reference, Q12)
    r = subprocess.run([EXE, "-c", str(dst), str(tmp_path / "nope")], capture_output=True, text=True)
    assert r.returncode == 255 and "Error loading file" in r.stdout
    (tmp_path / "empty").write_bytes(b"")
    r = subprocess.run([EXE, "-c", str(dst), str(tmp_path / "empty")], capture_output=True, text=True)
    assert r.returncode == 255 and "Error loading file" in r.stdout


def test_cli_decompress_gpu_and_host_paths(tmp_path):
    """`bce -d file archive` (GPU-assisted decoder) and `bce -ds` (host decoder) on a reference archive
    (bce.cpp:1428-1472): same bytes, the reference's summary line and exit codes."""
    data = oracle.synth_text(14, 400000) + oracle.synth_rand(14, 30000)
    arc, out1, out2 = tmp_path / "a.bce", tmp_path / "o1", tmp_path / "o2"
    arc.write_bytes(oracle.compress(data))
    r = subprocess.run([EXE, "-d", str(out1), str(arc)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Decompressed from %d B -> %d B in " % (arc.stat().st_size, len(data)) in r.stdout
    assert out1.read_bytes() == data
    r = subprocess.run([EXE, "-ds", str(out2), str(arc)], capture_output=True, text=True)
    assert r.returncode == 0 and out2.read_bytes() == data
    r = subprocess.run([EXE, "-d", str(out1), str(tmp_path / "nope.bce")], capture_output=True, text=True)
    assert r.returncode == 255 and "Archive not found." in r.stdout


def test_cli_block_container(tmp_path):
    """`bce -c3` (an extension): three blocks in a BCEM container, each exactly `bce -c` of that block; `bce -d` and
    `bce -ds` put the file back together; the Python side of the sharded bench reads the same container."""
    from bce_amd import container, sharding
    data = oracle.synth_text(15, 500001)
    src, arc, out = tmp_path / "in.txt", tmp_path / "a.bcem", tmp_path / "o"
    src.write_bytes(data)
    r = subprocess.run([EXE, "-c3", str(arc), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    archives, sizes = container.unpack_blocks(arc.read_bytes())
    assert len(archives) == 3 and sum(sizes) == len(data)
    for b in range(3):
        lo, hi = sharding.block_range(len(data), 3, b)
        assert sizes[b] == hi - lo and archives[b] == oracle.compress(data[lo:hi])
    for flag in ("-d", "-ds"):
        r = subprocess.run([EXE, flag, str(out), str(arc)], capture_output=True, text=True)
        assert r.returncode == 0 and out.read_bytes() == data


def test_cli_block_that_ran_out_of_memory_is_compressed_at_the_end(tmp_path):
    """A context of `bce -cN` that runs out of device memory gives its memory back and leaves its block to one context that
    has the device to itself at the end (main.cpp compress_blocks; BCE_CLI_TEST_NOMEM_BLOCK makes block 2's first attempt
    fail that way): the container is the one an undisturbed run writes."""
    import os
    data = oracle.synth_text(17, 700003)
    src, a1, a2 = tmp_path / "in.txt", tmp_path / "a1.bcem", tmp_path / "a2.bcem"
    src.write_bytes(data)
    r = subprocess.run([EXE, "-c5", str(a1), str(src)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([EXE, "-c5", str(a2), str(src)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, BCE_CLI_TEST_NOMEM_BLOCK="2"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert a1.read_bytes() == a2.read_bytes()


def test_cli_every_context_runs_out_of_memory_and_no_block_is_lost(tmp_path):
    """Round-4 advice: when EVERY worker of `bce -cN` leaves with "out of memory" (another tenant on the device), the blocks
    nobody had taken yet were in no list and the container went out with 0-byte archives, exit code 0.  With
    BCE_CLI_TEST_NOMEM_BLOCK=all every first attempt fails; nine blocks on one GPU means four workers and five blocks that no
    worker ever draws.  The container must still be the undisturbed one -- or the exit code non-zero, never a hollow file."""
    from bce_amd import container, sharding
    data = oracle.synth_text(18, 900007)
    src, a1, a2 = tmp_path / "in.txt", tmp_path / "a1.bcem", tmp_path / "a2.bcem"
    src.write_bytes(data)
    r = subprocess.run([EXE, "-c9", str(a1), str(src)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([EXE, "-c9", str(a2), str(src)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, BCE_CLI_TEST_NOMEM_BLOCK="all"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert a1.read_bytes() == a2.read_bytes()
    archives, sizes = container.unpack_blocks(a2.read_bytes())
    assert len(archives) == 9 and all(len(a) > 0 for a in archives)
    for b in range(9):
        lo, hi = sharding.block_range(len(data), 9, b)
        assert archives[b] == oracle.compress(data[lo:hi])


def test_cli_failed_write_is_not_a_success(tmp_path):
    """The output cannot be written (the directory does not exist): no success line, non-zero exit code -- the fast exit
    right behind the write must not hide it (-c, -cN, -d, -s)."""
    data = oracle.synth_text(19, 50000)
    src, arc = tmp_path / "in.txt", tmp_path / "a.bce"
    src.write_bytes(data)
    arc.write_bytes(oracle.compress(data))
    nowhere = str(tmp_path / "no_such_dir" / "x")
    for args in (["-c", nowhere, str(src)], ["-c2", nowhere, str(src)], ["-d", nowhere, str(arc)], ["-ds", nowhere, str(arc)],
                 ["-s", nowhere, str(src)]):
        r = subprocess.run([EXE] + args, capture_output=True, text=True, timeout=300)
        assert r.returncode == 251, (args, r.returncode, r.stdout, r.stderr)
        assert "Could not write" in r.stdout and "Compressed from" not in r.stdout and "Decompressed from" not in r.stdout


def test_cli_many_blocks_on_few_gpus(tmp_path):
    """`bce -c11` on a box with fewer GPUs than blocks: up to four gated contexts per device take the blocks in turn
    (main.cpp compress_blocks); every block is still exactly `bce -c` of its bytes, whichever context coded it."""
    from bce_amd import container, sharding
    data = oracle.synth_text(16, 1300003)
    src, arc = tmp_path / "in.txt", tmp_path / "a.bcem"
    src.write_bytes(data)
    r = subprocess.run([EXE, "-c11", str(arc), str(src)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    archives, sizes = container.unpack_blocks(arc.read_bytes())
    assert len(archives) == 11 and sum(sizes) == len(data)
    for b in range(11):
        lo, hi = sharding.block_range(len(data), 11, b)
        assert sizes[b] == hi - lo and archives[b] == oracle.compress(data[lo:hi])
    # and back: the blocks are decoded side by side (two contexts per GPU for -d, host threads for -ds)
    out = tmp_path / "o"
    for flag in ("-d", "-ds"):
        r = subprocess.run([EXE, flag, str(out), str(arc)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert out.read_bytes() == data
    # a container whose table lies about a block's size is refused, not written
    blob = bytearray(arc.read_bytes())
    blob[12] ^= 1
    bad = tmp_path / "bad.bcem"
    bad.write_bytes(bytes(blob))
    r = subprocess.run([EXE, "-d", str(out), str(bad)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 254 and "Could not read Archive." in r.stdout
    # wrapping / oversized tables (tests/test_decoder_cpu.py runs the whole list through -ds): the GPU path refuses them too
    import struct
    for raw0 in (2**64 - 100, 2**31, 2**40):
        blob = bytearray(arc.read_bytes())
        blob[12:20] = struct.pack("<Q", raw0)
        bad.write_bytes(bytes(blob))
        r = subprocess.run([EXE, "-d", str(out), str(bad)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 254 and "Could not read Archive." in r.stdout, (raw0, r.stdout, r.stderr)


def test_two_bce_processes_on_one_gpu_take_turns(tmp_path):
    """Two `bce -c` PROCESSES on the same device at the same time: their enumerations are kept apart by the advisory file lock
    behind the device gate (api.hip gate_file: /dev/shm/bce_hip_gate_<PCI address>), everything else overlaps.  Both archives
    must be the oracle's; nothing may hang (the single-launch rounds of two ungated processes could starve each other)."""
    datas = [oracle.synth_text(61, 6_000_000), oracle.synth_text(62, 5_000_000), oracle.synth_rand(63, 1_500_000)]
    procs = []
    for i, d in enumerate(datas):
        (tmp_path / ("in%d" % i)).write_bytes(d)
        procs.append(subprocess.Popen([EXE, "-c", str(tmp_path / ("a%d.bce" % i)), str(tmp_path / ("in%d" % i))], stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0, out + err
    for i, d in enumerate(datas):
        assert (tmp_path / ("a%d.bce" % i)).read_bytes() == oracle.compress(d)


def test_create_sized_prepares_and_is_only_a_hint():
    """bce_hip_create_sized (what `bce -c` calls with the file's size): the flush slots' pinned staging is allocated and touched
    on threads beside the runtime's start-up and registered afterwards.  The hint binds nothing: inputs smaller and larger than
    it, several compressions in a row and a context that is destroyed unused all behave as with bce_hip_create."""
    import ctypes as C

    import numpy as np
    lib = bce_amd.load_library()
    for hint in (0, 1000, 40_000_000, 3_000_000_000):            # (the last one is beyond 2^31: ignored)
        h = C.c_void_p()
        assert lib.bce_hip_create_sized(C.byref(h), 0, hint) == 0
        lib.bce_hip_destroy(h)                                     # unused: the staging threads are joined, their memory freed
    h = C.c_void_p()
    assert lib.bce_hip_create_sized(C.byref(h), 0, 40_000_000) == 0
    try:
        for data in (oracle.synth_text(3, 5000), oracle.synth_text(4, 6_000_000), oracle.synth_rand(5, 3_000_000), oracle.synth_text(3, 5000)):
            a = np.frombuffer(data, dtype=np.uint8)
            n = C.c_size_t()
            assert lib.bce_hip_compress(h, a.ctypes.data, len(a), None, 0, C.byref(n)) == 0
            out = np.empty(n.value, dtype=np.uint8)
            assert lib.bce_hip_archive_copy(h, out.ctypes.data, n.value) == 0
            assert out.tobytes() == oracle.compress(data)
    finally:
        lib.bce_hip_destroy(h)
    assert lib.bce_hip_create_sized(None, 0, 0) != 0


@pytest.mark.timeout(900)
def test_cli_blocks_of_a_file_beyond_the_single_archive_limit(tmp_path):
    """`bce -c` refuses 2^31 bytes and more as the reference does (saidx_t, bce.cpp:901) -- `bce -cN`, the extension, only asks that of
    every BLOCK: a 2.2 * 10^9-byte file goes through `-c2` (two blocks of 1.1 * 10^9 B, two contexts on the one GPU; a context
    that runs out of device memory leaves its block to the end, in -c and in -d alike), each block is the archive `bce -c`
    writes for those bytes (checked through the host-side header and the round trip; the oracle's own per-block answers at
    full size are tests/test_gpu_fullsize.py's), and `bce -d` puts the file back together."""
    import hashlib
    from bce_amd import container
    n = 2_200_000_000
    data = bce_amd.synth_text(5, n)
    src, arc, out = tmp_path / "in.bin", tmp_path / "a.bcem", tmp_path / "o.bin"
    data.tofile(str(src))
    want = hashlib.sha256(data).hexdigest()
    del data
    r = subprocess.run([EXE, "-c", str(arc), str(src)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 255 and "Error loading file" in r.stdout            # the single archive: n < 2^31
    r = subprocess.run([EXE, "-c2", str(arc), str(src)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    blob = arc.read_bytes()
    archives, raws = container.unpack_blocks(blob)
    assert raws == [n // 2, n - n // 2] and all(len(a) > 0 for a in archives)
    del blob, archives
    r = subprocess.run([EXE, "-d", str(out), str(arc)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    h = hashlib.sha256()
    with open(out, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            h.update(chunk)
    assert h.hexdigest() == want
    for p in (src, arc, out):
        p.unlink()
