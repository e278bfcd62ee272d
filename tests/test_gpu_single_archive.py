"""GPU: ONE reference-identical archive from several contexts / ranks (include/bce_hip.h "ONE archive from several
contexts"; SURVEY section 8e-2's aim): each codes only its planes, the finished streams are put together."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import bce_amd
import oracle
from conftest import ROOT

pytestmark = pytest.mark.gpu


def _encode_with_mask(data, mask, config=None):
    ctx = bce_amd.api._Ctx(0)
    bce_amd.set_plane_mask(ctx, mask)
    rf = bce_amd.RankFile(data, ctx=ctx)
    bce_amd.BCE(config).encode(rf)
    return ctx


@pytest.mark.parametrize("masks", [(0x0F, 0xF0), (0x55, 0xAA), (0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40, 0x80), (0xFF, 0x00)])
def test_streams_of_masked_contexts_make_the_oracles_archive(masks):
    """Contexts that code disjoint sets of planes of the same input: the streams of the others put into the first one
    give the archive of an unmasked run, which is the oracle's."""
    data = oracle.synth_text(11, 400000) + bytes(3000) + oracle.synth_rand(12, 50000)
    want = oracle.compress(data)
    ctxs = [_encode_with_mask(data, m) for m in masks]
    try:
        first = ctxs[0]
        if masks[0] != 0xFF:
            assert bytes(bce_amd.archive_of(first)) != want          # (its own archive lacks the other planes' symbols)
        for c, m in zip(ctxs[1:], masks[1:]):
            for p in range(8):
                if (m >> p) & 1:
                    bce_amd.set_plane_stream(first, p, bce_amd.plane_stream(c, p))
        assert bytes(bce_amd.archive_of(first)) == want
        assert bce_amd.decompress_device(bce_amd.archive_of(first)) == data
    finally:
        for c in ctxs:
            c.close()


def test_masked_context_records_only_its_planes():
    """Since round 4 the enumeration's rounds record no symbol for a plane another context codes (K3Args::pmask): the model,
    its sort and the device-to-host copy shrink with the mask.  The symbol counts of two complementary masks add up to a little
    more than the full run's (the tail's kernels still record every plane) and each is well below it."""
    data = oracle.synth_text(17, 3_000_000)
    full = _encode_with_mask(data, 0xFF)
    lo = _encode_with_mask(data, 0x0F)
    hi = _encode_with_mask(data, 0xF0)
    try:
        n_full, n_lo, n_hi = (bce_amd.api.stats_of(c)["symbols"] for c in (full, lo, hi))
        assert n_lo < n_full and n_hi < n_full
        assert n_full <= n_lo + n_hi < 2 * n_full          # (the tail's share is recorded by both: large on a 3 MB input)
        assert min(n_lo, n_hi) < (n_full * 9) // 10
    finally:
        for c in (full, lo, hi):
            c.close()


def test_mask_is_kept_and_can_be_taken_back():
    data = oracle.synth_text(13, 120000)
    want = oracle.compress(data)
    ctx = bce_amd.api._Ctx(0)
    try:
        bce_amd.set_plane_mask(ctx, 0x03)
        a1 = bce_amd.BCE().encode(bce_amd.RankFile(data, ctx=ctx))
        a2 = bce_amd.BCE().encode(bce_amd.RankFile(data, ctx=ctx))
        assert bytes(a1) == bytes(a2) != want
        bce_amd.set_plane_mask(ctx, 0xFF)
        assert bytes(bce_amd.BCE().encode(bce_amd.RankFile(data, ctx=ctx))) == want
        with pytest.raises(bce_amd.BceError):
            bce_amd.plane_stream(ctx, 8)
    finally:
        ctx.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_make_one_archive_over_gloo(world, tmp_path):
    """sharding.single_archive with `world` ranks on this one GPU (gloo for the gather): rank 0 ends up with the archive
    `bce -c` writes for the whole input -- the oracle's -- with a scanned-style custom config too."""
    code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
import bce_amd, oracle
from bce_amd import sharding
dist.init_process_group(backend="gloo")
rank = dist.get_rank()
ctx = bce_amd.api._Ctx(0)
data = oracle.synth_text(21, 700000) + oracle.synth_rand(22, 40000)
cfg = bytes((i * 7 + 3) %% 6 for i in range(288))
for config in (None, cfg):
    arch = sharding.single_archive(ctx, dist, torch.device("cpu"), data=data, config=config)
    if rank == 0:
        want = oracle.compress(data, config) if config is not None else oracle.compress(data)
        assert bytes(arch) == want, (len(arch), len(want))
    else:
        assert arch is None
# the mask is back to all planes: a plain compression on any rank is the oracle's again
assert bytes(bce_amd.BCE().encode(bce_amd.RankFile(data, ctx=ctx))) == oracle.compress(data)
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print("ONE_ARCHIVE_OK")
''' % ROOT
    script = tmp_path / "one_archive.py"
    script.write_text(code)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "ONE_ARCHIVE_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_bench_single_archive_two_ranks():
    """bench.py --single-archive as torch.distributed.run launches it (2 ranks on this GPU, gloo): the line says so and the
    archive of the whole job is the oracle's archive of the one input."""
    import hashlib
    import json
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--size", "3000000", "--backend", "gloo", "--no-cpu", "--single-archive"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["value"] > 0 and "ONE input" in j["config"]["sharding"]
    assert j["archive_sha256"] == hashlib.sha256(oracle.compress(oracle.synth_text(1, 3000000))).hexdigest()
    assert abs(j["value"] - 3000000 * 2 / (j["ms_per_step"] * 2e-3) / 1e6) < 0.02 * j["value"]      # one input per step, not two
