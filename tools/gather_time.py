#!/usr/bin/env python3
"""Host-side cost of sharding.gather_streams per step on ONE GPU (a process group of one rank over RCCL): what a rank pays
besides the wire -- size exchange, staging, H2D, the collective's launch, D2H, views / bytes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29617"), RANK="0", WORLD_SIZE="1")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from bce_amd import sharding  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
data = np.random.RandomState(1).randint(0, 256, 22913144).astype(np.uint8).tobytes()      # the size of the headline archive
for copy in (False, True):
    for _ in range(3):
        sharding.gather_streams(data, dist, dev, copy=copy)
    t0 = time.perf_counter()
    for _ in range(10):
        sharding.gather_streams(data, dist, dev, copy=copy)
    print("gather_streams(copy=%s): %.2f ms per call for %d B" % (copy, (time.perf_counter() - t0) * 100, len(data)))
pinned = torch.empty(len(data), dtype=torch.uint8, pin_memory=True)
pinned.copy_(torch.frombuffer(bytearray(data), dtype=torch.uint8))
for _ in range(3):
    sharding.gather_streams(pinned, dist, dev, copy=False, direct=True)
t0 = time.perf_counter()
for _ in range(10):
    got = sharding.gather_streams(pinned, dist, dev, copy=False, direct=True)
print("gather_streams(pinned tensor, copy=False): %.2f ms per call" % ((time.perf_counter() - t0) * 100))
assert bytes(got[0]) == data
dist.destroy_process_group()
