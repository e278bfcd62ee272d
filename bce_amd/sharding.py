"""Block sharding across the GPUs of one node: one independent block per rank, no data-path collective,
one gather of the coded streams to rank 0 (RCCL over xGMI with backend "nccl", gloo on CPU for tests)."""
import os

import numpy as np
import torch


def block_range(n, world, rank):
    """Contiguous block [lo, hi) of rank `rank` when n bytes are split over `world` ranks."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class _GatherBuffers:
    """Send / receive buffers of gather_streams, kept from one call to the next (grow-only): on a GPU the host side is
    pinned, so that a step's copies run at the bus' rate instead of through pageable staging."""

    def __init__(self, device, world):
        self.device, self.world, self.cap = device, world, 0
        self.send = self.recv = self.host_send = self.host_recv = None

    def ensure(self, need, is_dst):
        if need <= self.cap and (self.recv is not None or not is_dst):
            return
        cap = max(need, 2 * self.cap, 1 << 20)
        pin = self.device.type == "cuda"
        self.send = torch.empty(cap, dtype=torch.uint8, device=self.device)
        self.host_send = torch.empty(cap, dtype=torch.uint8, pin_memory=pin)
        if is_dst:
            self.recv = [torch.empty(cap, dtype=torch.uint8, device=self.device) for _ in range(self.world)]
            self.host_recv = torch.empty(self.world * cap, dtype=torch.uint8, pin_memory=pin)
        self.cap = cap


_gather_buffers = {}


def gather_streams(archive, dist, device, dst=0, copy=True, direct=False):
    """Gather every rank's coded stream to `dst`.  Returns the list of streams on dst, None elsewhere.

    Sizes differ per rank: one all_gather of the lengths (8 B each), then one padded gather.  Each peer has its own
    xGMI link to the root, so the direct gather uses them concurrently; the payload (~0.2-0.3 x block) is small.
    The buffers on both sides live on between calls (pinned host memory on a GPU): per step the host does one copy of its
    own stream into the send staging and, on dst, one device-to-host copy per stream at the bus' rate.
    `archive`: bytes-like, or a CPU uint8 tensor; with direct=True the caller vouches that the tensor is pinned (the buffer
    the encoder wrote the archive into) and it goes to the device without the staging copy (asking the tensor itself,
    `is_pinned()`, costs milliseconds on ROCm).  copy=False returns memoryviews into the receive buffer, valid until the next call (a step of a pipeline that hands
    them on at once); copy=True (default) returns bytes.
    """
    world, rank = dist.get_world_size(), dist.get_rank()
    key = (str(device), world)
    bufs = _gather_buffers.get(key)
    if bufs is None:
        bufs = _gather_buffers[key] = _GatherBuffers(device, world)
    n = len(archive)
    sz = torch.tensor([n], dtype=torch.int64, device=device)
    allsz = torch.empty(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(allsz, sz)
    sizes = [int(x) for x in allsz.cpu().tolist()]
    mx = max(max(sizes), 1)
    bufs.ensure(mx, rank == dst)
    if n:
        if direct and isinstance(archive, torch.Tensor) and archive.dtype == torch.uint8 and archive.device.type == "cpu" and archive.is_contiguous():
            bufs.send[:n].copy_(archive.reshape(-1), non_blocking=True)        # (the caller's buffer is sendable as it is: no staging copy)
        else:
            if isinstance(archive, torch.Tensor):
                bufs.host_send[:n].copy_(archive.reshape(-1))
            else:
                bufs.host_send[:n].numpy()[:] = np.frombuffer(archive, dtype=np.uint8)     # one memcpy into the (pinned) staging buffer
            bufs.send[:n].copy_(bufs.host_send[:n], non_blocking=True)
    outs = [r[:mx] for r in bufs.recv] if rank == dst else None
    dist.gather(bufs.send[:mx], outs, dst=dst)          # (what lies beyond a rank's own length is not looked at)
    if rank != dst:
        if device.type == "cuda":                       # the caller may reuse its (pinned) source buffer: the copy out of it is done
            torch.cuda.current_stream(device).synchronize()
        return None
    cap = bufs.cap
    for r in range(world):
        if sizes[r]:
            bufs.host_recv[r * cap:r * cap + sizes[r]].copy_(bufs.recv[r][:sizes[r]], non_blocking=True)
    if device.type == "cuda":
        torch.cuda.current_stream(device).synchronize()
    host = bufs.host_recv.numpy()
    views = [memoryview(host[r * cap:r * cap + sizes[r]]) for r in range(world)]
    return [bytes(v) for v in views] if copy else views


def planes_of(rank, world):
    """The planes rank `rank` of `world` codes in the one-archive mode: p with p % world == rank (the busiest planes of text
    are 0..3: round robin spreads them)."""
    return [p for p in range(8) if p % world == rank]


def single_archive(ctx, dist, device, data=None, device_ptr=None, n=None, config=None, dst=0, timings=None):
    """ONE archive, the one `bce -c` writes, from all ranks together (SURVEY section 8e-2's aim).

    Every rank holds the SAME input and runs the whole GPU part (rotation sort, planes, enumeration, model: a fifth of a
    step); the eight sequential range coders -- the step -- are shared out by plane (planes_of).  Each rank then sends its
    finished plane streams to `dst` in ONE gather (a 64-byte table of word counts, then the streams), `dst` puts them in the
    place of its own and the header is coded from the final sizes.  Returns the archive (bytearray) on dst, None elsewhere.
    No faster than one GPU -- the busiest plane's coder is the step wherever it runs -- but the compressed size is the
    whole-file one, which block sharding cannot give.  timings["gather_s"] accumulates what the mode adds to a compression."""
    from . import api
    world, rank = dist.get_world_size(), dist.get_rank()
    mine = planes_of(rank, world)
    mask = 0
    for p in mine:
        mask |= 1 << p
    api.set_plane_mask(ctx, mask)
    try:
        rf = api.RankFile(data, ctx=ctx) if device_ptr is None else api.RankFile(n=n, device_ptr=device_ptr, ctx=ctx)
        api.BCE(config).encode(rf)
        import ctypes as C
        import time
        t0 = time.perf_counter()
        # the rank's message -- eight u64 word counts, then its streams -- is laid out by the library straight in a send buffer
        # that lives on between calls (pinned on a GPU: the gather sends from it as it is)
        sizes = np.zeros(8, dtype=np.uint64)
        for p in mine:
            w = C.c_size_t()
            ctx.check(ctx.lib.bce_hip_plane_stream_size(ctx.h, p, C.byref(w)), "bce_hip_plane_stream_size")
            sizes[p] = w.value
        total = 64 + 2 * int(sizes.sum())
        key = ("one-archive", str(device))
        msg = _gather_buffers.get(key)
        if msg is None or msg.numel() < total:
            msg = _gather_buffers[key] = torch.empty(max(total + total // 4, 1 << 20), dtype=torch.uint8, pin_memory=device.type == "cuda")
        m = msg.numpy()
        m[:64] = np.frombuffer(sizes.tobytes(), dtype=np.uint8)
        at = 64
        for p in mine:
            ctx.check(ctx.lib.bce_hip_plane_stream_copy(ctx.h, p, m.ctypes.data + at, int(sizes[p])), "bce_hip_plane_stream_copy")
            at += 2 * int(sizes[p])
        got = gather_streams(msg[:total], dist, device, dst=dst, copy=False, direct=True)
        if rank != dst:
            if timings is not None:
                timings["gather_s"] = timings.get("gather_s", 0.0) + time.perf_counter() - t0
            return None
        for r, g in enumerate(got):
            if r == rank:
                continue
            g = np.frombuffer(g, dtype=np.uint8)
            theirs = np.frombuffer(g[:64], dtype=np.uint64)
            at = 64
            for p in planes_of(r, world):
                words = int(theirs[p])
                if at + 2 * words > g.size:
                    raise ValueError("rank %d sent %d bytes, its table asks for more" % (r, g.size))
                ctx.check(ctx.lib.bce_hip_plane_stream_set(ctx.h, p, g.ctypes.data + at, words), "bce_hip_plane_stream_set")
                at += 2 * words
            if at != g.size:
                raise ValueError("rank %d sent %d bytes, its table says %d" % (r, g.size, at))
        arch = api.archive_of(ctx)
        if timings is not None:       # (gather + putting the streams in + laying the archive out: what the mode adds to a compression)
            timings["gather_s"] = timings.get("gather_s", 0.0) + time.perf_counter() - t0
        return arch
    finally:
        api.set_plane_mask(ctx, 0xFF)


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
