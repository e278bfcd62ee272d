"""Block sharding across the GPUs of one node: one independent block per rank, no data-path collective,
one gather of the coded streams to rank 0 (RCCL over xGMI with backend "nccl", gloo on CPU for tests)."""
import os

import torch


def block_range(n, world, rank):
    """Contiguous block [lo, hi) of rank `rank` when n bytes are split over `world` ranks."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_streams(archive: bytes, dist, device, dst=0):
    """Gather every rank's coded stream to `dst`.  Returns the list of streams on dst, None elsewhere.

    Sizes differ per rank: all_gather the lengths (8 B each), then one padded gather.  Each peer has its own
    xGMI link to the root, so the direct gather uses them concurrently; the payload (~0.2-0.3 x block) is small.
    """
    world, rank = dist.get_world_size(), dist.get_rank()
    a = (torch.frombuffer(bytearray(archive), dtype=torch.uint8) if len(archive) else torch.empty(0, dtype=torch.uint8)).to(device)
    sz = torch.tensor([a.numel()], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(sz) for _ in range(world)]
    dist.all_gather(sizes, sz)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    pad = torch.zeros(mx, dtype=torch.uint8, device=device)
    pad[:a.numel()] = a
    outs = [torch.empty(mx, dtype=torch.uint8, device=device) for _ in range(world)] if rank == dst else None
    dist.gather(pad, outs, dst=dst)
    if rank != dst:
        return None
    return [o[:s].cpu().numpy().tobytes() for o, s in zip(outs, sizes)]


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def format_cpulist(cpus):
    """{0,1,2,3,8} -> "0-3,8" (the kernel's cpulist notation)."""
    out, run = [], []
    for c in sorted(cpus):
        if run and c == run[-1] + 1:
            run.append(c)
        else:
            if run:
                out.append(run)
            run = [c]
    if run:
        out.append(run)
    return ",".join("%d" % r[0] if len(r) == 1 else "%d-%d" % (r[0], r[-1]) for r in out)


def pin_to_local_numa(device_index):
    """Restrict this process (and the threads it starts afterwards: the context's 8 range-coder threads) to the CPUs of
    the NUMA node the GPU hangs off, so that 8 ranks x 9 host threads do not wander over both sockets.  Call before the
    bce_hip context is created.  Returns the CPU set, or None when the topology cannot be read (nothing is changed)."""
    try:
        props = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
        with open("/sys/bus/pci/devices/%s/numa_node" % bdf) as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
            cpus = _parse_cpulist(f.read()) & os.sched_getaffinity(0)
        if len(cpus) < 9:          # too few for 8 coder threads + the driver thread: leave the affinity alone
            return None
        os.sched_setaffinity(0, cpus)
        return cpus
    except Exception:
        return None
