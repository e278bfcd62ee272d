#!/usr/bin/env python3
"""Where a gather_streams call spends its time (one rank over RCCL): each stage timed with a device synchronisation."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29618"), RANK="0", WORLD_SIZE="1")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
n = 22913144
pinned = torch.empty(n, dtype=torch.uint8, pin_memory=True)
send = torch.empty(n, dtype=torch.uint8, device=dev)
recv = [torch.empty(n, dtype=torch.uint8, device=dev)]
host = torch.empty(n, dtype=torch.uint8, pin_memory=True)
sync = torch.cuda.synchronize
acc = {}


def lap(name, t0):
    sync()
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return time.perf_counter()


for it in range(13):
    if it == 3:
        acc.clear()
    t = time.perf_counter()
    sz = torch.tensor([n], dtype=torch.int64, device=dev); t = lap("size tensor", t)
    allsz = torch.empty(1, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allsz, sz); t = lap("all_gather", t)
    sizes = allsz.cpu().tolist(); t = lap("sizes to host", t)
    send.copy_(pinned, non_blocking=True); t = lap("H2D", t)
    dist.gather(send, recv, dst=0); t = lap("gather", t)
    host.copy_(recv[0], non_blocking=True); t = lap("D2H", t)
    v = memoryview(host.numpy()); t = lap("view", t)
for k, v in acc.items():
    print("%-14s %.3f ms" % (k, v * 100))
dist.destroy_process_group()
