"""CPU, world_size 2 over gloo: the N>1 path (block split, size exchange + gather of coded streams, container).
The per-block payload is produced by the oracle here (no GPU in this test); on the GPU box bench.py feeds the
same helper with the HIP archives over RCCL."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from bce_amd import container, sharding
    data = oracle.synth_text(5, 300001)
    lo, hi = sharding.block_range(len(data), world, rank)
    mine = oracle.compress(data[lo:hi])
    streams = sharding.gather_streams(mine, dist, torch.device("cpu"))
    if rank == 0:
        raws = [sharding.block_range(len(data), world, r) for r in range(world)]
        blob = container.pack_blocks(streams, [b - a for a, b in raws])
        archives, sizes = container.unpack_blocks(blob)
        ok = sizes == [b - a for a, b in raws] and sum(sizes) == len(data)
        for r, (a, b) in enumerate(raws):
            ok = ok and archives[r] == oracle.compress(data[a:b])
        q.put(ok)
    else:
        assert streams is None
    dist.barrier()
    dist.destroy_process_group()


def test_block_sharding_gather_and_container_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _worker8(rank, world, port, q):
    """World 8, the stated shape of BASELINE configs[3] (one input, eight contiguous blocks, gather to rank 0), on the CPU: the
    oracle stands in for the GPU path, everything around it -- block split, size exchange, padded gather, container, and the
    host decoder reading the container's blocks back -- is the code the 8-GPU run uses."""
    sys.path.insert(0, ROOT)
    import ctypes as C

    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from bce_amd import container, sharding
    data = oracle.synth_text(9, 160003)                     # (ragged: three blocks of 20001, five of 20000)
    lo, hi = sharding.block_range(len(data), world, rank)
    mine = oracle.compress(data[lo:hi])
    for _ in range(2):
        streams = sharding.gather_streams(mine, dist, torch.device("cpu"), copy=False)
    if rank == 0:
        raws = [sharding.block_range(len(data), world, r) for r in range(world)]
        blob = container.pack_blocks(streams, [b - a for a, b in raws])
        archives, sizes = container.unpack_blocks(blob)
        ok = len(archives) == 8 and sizes == [b - a for a, b in raws] and sum(sizes) == len(data)
        import bce_amd
        lib = bce_amd.load_library()
        back = bytearray()
        for r, (a, b) in enumerate(raws):
            ok = ok and archives[r] == oracle.compress(data[a:b])
            arr = np.frombuffer(archives[r], dtype=np.uint8)
            out = np.empty(b - a, dtype=np.uint8)
            n = C.c_size_t()
            ok = ok and lib.bce_hip_decompress(arr.ctypes.data, len(arr), out.ctypes.data, len(out), C.byref(n)) == 0 and n.value == b - a
            back += out.tobytes()
        q.put(bool(ok and bytes(back) == bytes(data)))
    else:
        assert streams is None
    dist.barrier()
    dist.destroy_process_group()


def test_block_sharding_world8_one_input_in_eight_blocks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_block_range_partitions_exactly():
    from bce_amd import sharding
    for n in (1, 7, 8, 100, 1000003):
        for w in (1, 2, 3, 8):
            r = [sharding.block_range(n, w, k) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))


def test_container_roundtrip_and_errors():
    from bce_amd import container
    blob = container.pack_blocks([b"abc", b"", b"0123456789"], [10, 0, 99])
    a, s = container.unpack_blocks(blob)
    assert a == [b"abc", b"", b"0123456789"] and s == [10, 0, 99]
    with pytest.raises(ValueError):
        container.unpack_blocks(b"XXXX" + blob[4:])
    with pytest.raises(ValueError):
        container.unpack_blocks(blob + b"x")


def _ragged_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bce_amd import sharding
    sizes = [0, 1, 70001][:world]
    mine = np.random.RandomState(100 + rank).randint(0, 256, sizes[rank]).astype(np.uint8).tobytes()
    for _ in range(2):                                  # twice: the helper keeps no state between steps
        streams = sharding.gather_streams(mine, dist, torch.device("cpu"))
    if rank == 0:
        want = [np.random.RandomState(100 + r).randint(0, 256, sizes[r]).astype(np.uint8).tobytes() for r in range(world)]
        q.put(streams == want)
    else:
        assert streams is None
    dist.barrier()
    dist.destroy_process_group()


def test_gather_streams_ragged_sizes_world3():
    """An empty stream, a one-byte stream and a long one: the padded gather must hand each back exactly."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ragged_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _run_bench_ranks(world, total, extra=(), timeout=900, backend="gloo"):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per process), rehearsed on ONE GPU with
    the gloo backend.  -> (the JSON line, the BCEM container rank 0 wrote)."""
    import json
    import subprocess
    import tempfile
    port = _free_port()
    with tempfile.TemporaryDirectory() as td:
        cpath = os.path.join(td, "blocks.bcem")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
               "--total-size", str(total), "--backend", backend, "--no-cpu", "--container-out", cpath] + list(extra)
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        with open(cpath, "rb") as f:
            blob = f.read()
    return json.loads(line), blob


def _check_blocks_against_oracle(j, blob, world, total):
    """BASELINE configs[3]'s contract: ONE input cut into `world` contiguous blocks, every block's archive the one the
    reference writes for that block alone (bce.cpp:1151-1157), and the container decodes back to the input."""
    import subprocess
    import tempfile

    import numpy as np
    import oracle
    from bce_amd import container, sharding
    whole = np.frombuffer(oracle.synth_text(1, total), dtype=np.uint8)
    archives, raws = container.unpack_blocks(blob)
    assert len(archives) == world
    for r in range(world):
        lo, hi = sharding.block_range(total, world, r)
        assert raws[r] == hi - lo == j["config"]["bytes_per_gpu"][r]
        assert archives[r] == oracle.compress(whole[lo:hi]), "block %d of %d differs from the oracle's archive of that block" % (r, world)
    assert j["config"]["total_bytes"] == total and j["archive_bytes"] == sum(len(a) for a in archives)
    assert j["container_bytes"] == len(blob)
    with tempfile.TemporaryDirectory() as td:                      # `bce -d` takes the container as it is
        cp, op, ip, c2 = (os.path.join(td, x) for x in ("c.bcem", "out.bin", "in.bin", "cli.bcem"))
        with open(cp, "wb") as f:
            f.write(blob)
        exe = os.path.join(ROOT, "bce_amd", "bin", "bce")
        r = subprocess.run([exe, "-d", op, cp], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:]
        assert np.array_equal(np.fromfile(op, dtype=np.uint8), whole)
        # the same blocks through the CLI: `bce -cN` cuts the file the same way and writes the same container, byte for byte
        whole.tofile(ip)
        r = subprocess.run([exe, "-c%d" % world, c2, ip], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:]
        with open(c2, "rb") as f:
            assert f.read() == blob


@pytest.mark.gpu
def test_bench_two_ranks_one_input_in_two_blocks():
    """N = 2 rehearsal at reduced size: ONE input cut into two contiguous blocks, every rank compresses its block on the
    GPU, rank 0 gathers the HIP archives through sharding.gather_streams -- the code path the nccl backend takes over RCCL --
    and packs the container; every block equals the oracle's archive of it and the container decodes to the input."""
    total = 6_000_001                                           # (odd: the blocks differ by one byte)
    j, blob = _run_bench_ranks(2, total)
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["value"] > 0
    assert len(j["ms_per_step_per_rank"]) == 2 and all(t > 0 for t in j["ms_per_step_per_rank"])
    # what a first 8-GPU run needs to explain itself: backend, world size, gather time and bytes, per-rank host facts
    c = j["comm"]
    assert c["backend"] == "gloo" and c["world"] == 2 and c["gather_ms"] >= 0 and len(c["gather_ms_per_rank"]) == 2
    assert c["gather_bytes"] > 2 * 100000
    assert len(j["per_rank"]) == 2
    for r_ in j["per_rank"]:
        assert r_["coder_busy_ms"] > 0 and r_["k1_ms"] > 0 and r_["cpus"] >= 1 and isinstance(r_["cpu_set"], str) and "numa_pinned" in r_
    _check_blocks_against_oracle(j, blob, 2, total)


@pytest.mark.gpu
def test_bench_four_ranks_one_input_in_four_blocks():
    """N = 4 on the one GPU (the box allows six processes on the card; the driver's N = 8 run is the same code with world 8)."""
    total = 5_000_003
    j, blob = _run_bench_ranks(4, total)
    assert j["n_gpus"] == 4 and j["scaling"] == "strong" and len(j["per_rank"]) == 4
    _check_blocks_against_oracle(j, blob, 4, total)


@pytest.mark.gpu
def test_bench_two_ranks_full_size_blocks_against_the_oracle_goldens():
    """N = 2 at FULL size: synth-text v1 seed 1, 10^9 B, two blocks of 5 x 10^8 B (two contexts on the one GPU).  bench.py itself
    compares every gathered block with the oracle-made known answer (tests/golden/oracle_fullsize.json, synth-text-1e9-bRof2,
    tools/make_oracle_golden.py) -- what the driver's 8-GPU line reports as oracle_golden_blocks."""
    import json
    with open(os.path.join(ROOT, "tests", "golden", "oracle_fullsize.json")) as f:
        gold = {v["name"]: v for v in json.load(f)["vectors"]}
    total = 1_000_000_000
    j, blob = _run_bench_ranks(2, total, timeout=1500)
    assert j["oracle_golden_blocks"] == "identical" and j["oracle_golden_per_block"] == ["identical", "identical"]
    assert j["archive_bytes"] == sum(gold["synth-text-1e9-b%dof2" % r]["archive_bytes"] for r in range(2))
    assert j["config"]["total_bytes"] == total and j["config"]["bytes_per_gpu"] == [500_000_000, 500_000_000]
    from bce_amd import container
    archives, raws = container.unpack_blocks(blob)
    import hashlib
    for r in range(2):
        assert hashlib.sha256(archives[r]).hexdigest() == gold["synth-text-1e9-b%dof2" % r]["archive_sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_over_rccl_on_as_many_gpus_as_the_box_has(world):
    """The real thing, for a box that HAS several GPUs (the pool's boxes have one: skipped there, and the driver's 8-GPU SCALE run
    is this command at full size): bench.py under torch.distributed.run, one rank per GPU, backend nccl = RCCL over xGMI -- every
    gathered block == the oracle's archive of it, the container decodes back to the input."""
    import torch
    if torch.cuda.device_count() < world:
        pytest.skip("%d GPUs on this box, %d ranks need one each (RCCL does not share a device between ranks)" % (torch.cuda.device_count(), world))
    total = 24_000_003
    j, blob = _run_bench_ranks(world, total, backend="nccl")
    assert j["n_gpus"] == world and j["comm"]["backend"] == "nccl" and j["comm"]["world"] == world
    _check_blocks_against_oracle(j, blob, world, total)


@pytest.mark.gpu
def test_bench_four_ranks_full_size_blocks_against_the_oracle_goldens():
    """N = 4 at FULL size (four blocks of 2.5 x 10^8 B, four rank processes on the one GPU: the box allows six): bench.py's own
    per-block verdicts against synth-text-1e9-bRof4, and the container's archives hashed here as well."""
    import hashlib
    import json
    with open(os.path.join(ROOT, "tests", "golden", "oracle_fullsize.json")) as f:
        gold = {v["name"]: v for v in json.load(f)["vectors"]}
    total = 1_000_000_000
    j, blob = _run_bench_ranks(4, total, timeout=1500)
    assert j["oracle_golden_blocks"] == "identical" and j["oracle_golden_per_block"] == ["identical"] * 4
    assert j["config"]["bytes_per_gpu"] == [250_000_000] * 4
    from bce_amd import container
    archives, raws = container.unpack_blocks(blob)
    assert [hashlib.sha256(a).hexdigest() for a in archives] == [gold["synth-text-1e9-b%dof4" % r]["archive_sha256"] for r in range(4)]


def test_block_goldens_cover_the_stated_config():
    """BASELINE configs[3] (enwik9-sized input cut over 2 / 4 / 8 GPUs): an oracle-made archive hash for every block."""
    import json
    from bce_amd import sharding
    with open(os.path.join(ROOT, "tests", "golden", "oracle_fullsize.json")) as f:
        gold = {v["name"]: v for v in json.load(f)["vectors"]}
    for world in (2, 4, 8):
        for r in range(world):
            v = gold["synth-text-1e9-b%dof%d" % (r, world)]
            lo, hi = sharding.block_range(1_000_000_000, world, r)
            assert v["n"] == hi - lo and v["block"] == {"rank": r, "world": world, "of_n": 1_000_000_000}
            assert len(v["archive_sha256"]) == 64 and len(v["input_sha256"]) == 64 and v["archive_bytes"] > 0


@pytest.mark.gpu
def test_gather_streams_over_rccl_on_one_gpu():
    """The nccl (= RCCL) side of gather_streams with device tensors and pinned host buffers, as far as ONE GPU can show
    it: a process group of one rank, streams of changing sizes (the buffers grow and are reused), bytes and views."""
    import subprocess
    port = _free_port()
    code = r'''
import os, sys
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="%d", RANK="0", WORLD_SIZE="1")
import numpy as np, torch, torch.distributed as dist
from bce_amd import sharding
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
rs = np.random.RandomState(3)
for n in (5, 3000000, 0, 70000, 9000000, 1):
    data = rs.randint(0, 256, n).astype(np.uint8).tobytes()
    got = sharding.gather_streams(data, dist, dev)
    assert isinstance(got, list) and len(got) == 1 and got[0] == data, n
    view = sharding.gather_streams(data, dist, dev, copy=False)
    assert bytes(view[0]) == data, n
dist.barrier()
dist.destroy_process_group()
print("RCCL_GATHER_OK")
''' % (ROOT, port)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0 and "RCCL_GATHER_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
