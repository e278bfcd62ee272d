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
