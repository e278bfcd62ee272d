"""GPU: archives at BASELINE.json's sizes against the ORACLE's archives of the same inputs.

tests/golden/oracle_fullsize.json is written by tools/make_oracle_golden.py, which runs oracle/bce_oracle.c (CPU, single
thread, 40-65 s per 10^8 bytes) and nothing else: every hash below has oracle provenance, none is the GPU path's own
output.  Inputs are regenerated (synthetic generators, or the corpora built from this image's files) and their sha256
checked first; a corpus this box cannot rebuild identically is skipped, the synthetic ones never are."""
import hashlib

import pytest
import torch

import bce_amd
from conftest import fullsize_input, load_fullsize_golden

pytestmark = pytest.mark.gpu
GOLD = load_fullsize_golden()


def _compress(data):
    t = torch.from_numpy(data).to("cuda:0")
    torch.cuda.synchronize()
    arch, st = bce_amd.compress_device(t.data_ptr(), len(data))
    del t
    torch.cuda.empty_cache()
    return arch, st


@pytest.mark.parametrize("name", ["synth-text-1e8", "synth-rand-32Mi", "natural-1e8", "binary-1e8", "natural-16Mi", "binary-16Mi"])
def test_archive_equals_the_oracles(name):
    """BASELINE configs[1] (enwik8-sized) on text, source code, executables and random bytes."""
    v = GOLD[name]
    data = fullsize_input(v)
    if data is None:
        pytest.skip("this box cannot rebuild the %s corpus bit for bit (files of another image)" % v["kind"])
    arch, st = _compress(data)
    assert st["nodes"] == 8 * v["n"] - 8
    assert len(arch) == v["archive_bytes"]
    assert hashlib.sha256(arch).hexdigest() == v["archive_sha256"]


def test_above_2_27_bytes_context_wrap_matches_the_oracle():
    """n = 1.5 * 10^8 > 2^27: AdaptiveCoder::get_context's `c1 << bits` wraps in uint32 (bce.cpp:671-677, SURVEY quirk
    Q1) for the nodes whose N(0w) is 2^27 or more -- the regime of BASELINE configs 3/4 (enwik9).  The archive must
    equal the oracle's, which keeps the reference's uint32 expression; the decoder resolves the same contexts."""
    v = GOLD["synth-text-1.5e8"]
    assert v["n"] > (1 << 27)
    data = fullsize_input(v)
    assert data is not None
    arch, st = _compress(data)
    assert st["nodes"] == 8 * v["n"] - 8
    assert len(arch) == v["archive_bytes"] and hashlib.sha256(arch).hexdigest() == v["archive_sha256"]
    back = bce_amd.decompress_device(arch)
    assert len(back) == v["n"] and hashlib.sha256(back).hexdigest() == v["input_sha256"]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("name", ["natural-1e8", "synth-text-1e8"])
def test_stream_through_three_contexts_equals_the_oracles(name):
    """A stream of full-size inputs through three gated contexts (bce_amd.ContextPool): K1 of one input runs beside the
    enumeration and the model flushes of the others, so blocks of a launch start late and out of step -- which is what
    exposed K1's active-list kernel sharing a word between its per-block counts and its result (k1_plan's block count is
    not monotonic in the element count; 10^8 bytes of source code hit it, text did not).  Every archive of the stream
    must be the oracle's."""
    v = GOLD[name]
    data = fullsize_input(v)
    if data is None:
        pytest.skip("this box cannot rebuild the %s corpus bit for bit (files of another image)" % v["kind"])
    t = torch.from_numpy(data).to("cuda:0")
    torch.cuda.synchronize()
    with bce_amd.ContextPool(3, 0) as pool:
        for _ in range(2):                       # cold contexts first, then warm ones
            res = pool.compress_many([(t.data_ptr(), len(data))] * 6, on_device=True)
            assert [hashlib.sha256(a).hexdigest() for a in res] == [v["archive_sha256"]] * 6
    del t
    torch.cuda.empty_cache()


@pytest.mark.timeout(900)
def test_scanned_table_at_silesia_size_equals_the_oracles():
    """BASELINE configs[4] (Silesia-sized, mixed, tuned AdaptiveCoder): 2 x 10^8 bytes of natural corpus || binary corpus.
    `bce -s` on the GPU path (K1-K3 in scan mode + the threaded host ScanSet) must give the 288 bytes oracle.scan gave,
    and `bce -c` with that table the oracle's archive.  The two are checked apart: the compression uses the ORACLE's
    table from the golden file, so a scan difference cannot hide behind (or cause) an archive difference."""
    v = GOLD["mixed-2e8-scanned"]
    data = fullsize_input(v)
    if data is None:
        pytest.skip("this box cannot rebuild the corpora bit for bit (files of another image)")
    cfg_oracle = bytes.fromhex(v["config_hex"])
    assert hashlib.sha256(cfg_oracle).hexdigest() == v["config_sha256"]
    cfg, sizes = bce_amd.scan(data)
    assert bytes(cfg) == cfg_oracle
    assert sizes == v["scan_result_sizes"]                      # the same doubles: every sum keeps the reference's order (Q11)
    t = torch.from_numpy(data).to("cuda:0")
    torch.cuda.synchronize()
    arch, st = bce_amd.compress_device(t.data_ptr(), len(data), config=cfg_oracle)
    del t
    torch.cuda.empty_cache()
    assert st["nodes"] == 8 * v["n"] - 8
    assert len(arch) == v["archive_bytes"] and hashlib.sha256(arch).hexdigest() == v["archive_sha256"]


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("world", [8, 4])
def test_config4_blocks_of_the_1e9_input_equal_the_oracles(world, tmp_path):
    """BASELINE configs[3] in its stated form, as far as ONE GPU can run it: synth-text v1 seed 1, 10^9 B, cut into `world`
    contiguous blocks by sharding.block_range (what rank r of bench.py --gpus N compresses), every block compressed on the HIP
    path -- through a ContextPool of four contexts, then once more by `bce -cN` on the file -- and compared with the
    oracle-made archive of that block alone (tests/golden/oracle_fullsize.json, synth-text-1e9-bRofN; one block per archive,
    bce.cpp:1151-1157).  The container `bce -cN` writes must be exactly those archives behind the block table, and decode
    back to the input.  What is left for hardware to show is RCCL with more than one rank."""
    import os
    import subprocess

    import numpy as np
    from bce_amd import container, sharding
    from conftest import ROOT
    whole = GOLD["synth-text-1e9"]
    data = fullsize_input(whole)
    assert data is not None and len(data) == 1_000_000_000
    ranges = [sharding.block_range(len(data), world, r) for r in range(world)]
    gold = [GOLD["synth-text-1e9-b%dof%d" % (r, world)] for r in range(world)]
    for (lo, hi), v in zip(ranges, gold):
        assert hi - lo == v["n"] and hashlib.sha256(data[lo:hi].tobytes()).hexdigest() == v["input_sha256"]
    with bce_amd.ContextPool(4, 0) as pool:
        res = pool.compress_many([data[lo:hi] for lo, hi in ranges], with_stats=True)
    for r, ((arch, st), v) in enumerate(zip(res, gold)):
        assert st["nodes"] == 8 * v["n"] - 8, r
        assert len(arch) == v["archive_bytes"] and hashlib.sha256(arch).hexdigest() == v["archive_sha256"], "block %d of %d" % (r, world)
    want_container = container.pack_blocks([bytes(a) for a, _ in res], [hi - lo for lo, hi in ranges])
    del res
    # the command line on the same bytes: `bce -cN` (main.cpp compress_blocks: up to four gated contexts on the one device)
    exe = os.path.join(ROOT, "bce_amd", "bin", "bce")
    src, arc, back = tmp_path / "in.bin", tmp_path / "a.bcem", tmp_path / "back.bin"
    data.tofile(str(src))
    r_ = subprocess.run([exe, "-c%d" % world, str(arc), str(src)], capture_output=True, text=True, timeout=900)
    assert r_.returncode == 0, r_.stdout + r_.stderr
    blob = arc.read_bytes()
    assert blob == want_container
    del want_container
    archives, raws = container.unpack_blocks(blob)
    assert raws == [hi - lo for lo, hi in ranges]
    assert [hashlib.sha256(a).hexdigest() for a in archives] == [v["archive_sha256"] for v in gold]
    del archives, blob
    if world == 8:                                              # and back: the blocks decoded side by side, the file put together
        r_ = subprocess.run([exe, "-d", str(back), str(arc)], capture_output=True, text=True, timeout=900)
        assert r_.returncode == 0, r_.stdout + r_.stderr
        h = hashlib.sha256()
        with open(back, "rb") as f:
            for chunk in iter(lambda: f.read(1 << 24), b""):
                h.update(chunk)
        assert h.hexdigest() == whole["input_sha256"]
    for p in (src, arc, back):
        if p.exists():
            p.unlink()
