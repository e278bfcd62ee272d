#!/usr/bin/env python3
"""bench.py -- headline benchmark of the `bce -c` hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step = one full compression (K1 rotation sort/BWT, K2 planes, K3 enumeration, K4 model, host range
coders, framing) of one input block per GPU, input already resident in HBM, archive bytes ready on the
host at the end.  N > 1 (launched by torch.distributed.run, one rank per GPU): every rank compresses
its own block (weak scaling, no data-path collective) and the coded streams are gathered to rank 0
over RCCL inside the timed step.  Rank 0 prints ONE JSON line.

Workload: BASELINE.json config[1] is enwik8 (10^8 bytes).  The corpus is not available offline; if a
file is given with --file (or $BCE_BENCH_FILE) it is used, otherwise the stand-in is synth-text v1
(SURVEY 8c generator) at 10^8 bytes, seed 1 + rank.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import bce_amd  # noqa: E402
from bce_amd import sharding  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=100_000_000, help="input bytes per GPU (enwik8 = 10^8)")
    ap.add_argument("--workload", default="synth-text", choices=["synth-text", "synth-rand"])
    ap.add_argument("--file", default=os.environ.get("BCE_BENCH_FILE"))
    ap.add_argument("--cpu-sample", type=int, default=48 << 20, help="bytes of the workload the CPU baseline compresses")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-decode", action="store_true", help="skip the untimed decode-and-compare leg (N=1 only)")
    ap.add_argument("--scan-config", action="store_true",
                    help="BASELINE config 5: run `bce -s` on the input first (untimed), compress with the scanned table")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the real multi-GPU path); gloo = rehearsal of the N>1 control flow on one GPU")
    return ap.parse_args()


def make_input(args, rank):
    if args.file:
        data = np.fromfile(args.file, dtype=np.uint8)
        per = len(data) // max(1, args.gpus)
        blk = data[rank * per:(rank + 1) * per] if args.gpus > 1 else data
        return np.ascontiguousarray(blk), "file:%s[%d B, sha256 %s]" % (os.path.basename(args.file), len(data), hashlib.sha256(data.tobytes()).hexdigest()[:16])
    gen = bce_amd.synth_text if args.workload == "synth-text" else bce_amd.synth_rand
    return gen(1 + rank, args.size), "%s-v1 seed %d, %d B per GPU (enwik8-sized stand-in: the corpus is not available offline)" % (args.workload, 1 + rank, args.size)


def cpu_baseline(data, sample_bytes):
    """The oracle (bit-exact CPU restatement of bce -c) timed on this host on a bounded sample: single thread (the
    reference built without OpenMP) and 8 threads (its OpenMP build: one thread per plane, joined every round,
    bce.cpp:1250-1252; suffix sort and plane build stay serial there too).  The faster one is `value`."""
    import oracle
    oracle.build()
    sample = data[:sample_bytes].tobytes()
    runs = {}
    for threads in (1, 8):
        oracle.set_threads(threads)
        t0 = time.time()
        arch = oracle.compress(sample)
        runs[threads] = (time.time() - t0, len(arch), hashlib.sha256(arch).hexdigest())
    oracle.set_threads(1)
    assert runs[1][2] == runs[8][2]
    best = min(runs, key=lambda t: runs[t][0])
    return {"value": round(len(sample) / runs[best][0] / 1e6, 3), "unit": "MB/s", "cores": best, "kind": "port",
            "sample": "first %d B of the workload, oracle/bce_oracle.c: 1 thread %.1f s (%.2f MB/s), 8 OpenMP threads %.1f s (%.2f MB/s), archive %d B" % (
                len(sample), runs[1][0], len(sample) / runs[1][0] / 1e6, runs[8][0], len(sample) / runs[8][0] / 1e6, runs[1][1]),
            "single_thread_value": round(len(sample) / runs[1][0] / 1e6, 3),
            "host_cpus": os.cpu_count()}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend="gloo")
            local = local % max(1, torch.cuda.device_count())     # rehearsal: ranks may share a GPU
    n_gpus = world
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    comm_dev = dev if args.backend == "nccl" else torch.device("cpu")

    data, workload = make_input(args, rank)
    n = len(data)
    t_in = torch.from_numpy(data).to(dev)       # input resident in HBM before the timed region
    torch.cuda.synchronize()
    ctx = bce_amd.api._Ctx(local)
    config, coder_config = None, "default AdaptiveCoder<31> tables"
    if args.scan_config:       # the .bcc a user would have produced beforehand with `bce -s` (not part of `bce -c`)
        t0 = time.perf_counter()
        config, _ = bce_amd.scan(data, device=local)
        coder_config = "table scanned from the input by bce -s (GPU enumeration + host ScanCoder, %.1f s, untimed), sha256 %s" % (
            time.perf_counter() - t0, hashlib.sha256(bytes(config)).hexdigest()[:16])

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    gathered = [None]

    def step():
        arch, st = bce_amd.compress_device(t_in.data_ptr(), n, config=config, ctx=ctx)
        if dist is not None:
            # RCCL gather of the per-block coded streams to rank 0 (size exchange, then padded gather)
            gathered[0] = sharding.gather_streams(arch, dist, comm_dev)
        return arch, st

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    sts = []
    arch = None
    for _ in range(args.steps):
        arch, st = step()
        sts.append(st)
    barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device=comm_dev if dist is not None else dev)
    if dist is not None:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())

    if rank == 0:
        steps = max(1, args.steps)
        st = sts[-1]
        k3_s = sum(s["k3_ms"] for s in sts) / len(sts) / 1e3
        alg_bytes = 384.0 * n + 16.0 * st["symbols"]     # 48 B/node x 8n nodes + 16 B/symbol
        traffic = None    # HBM bytes from PMC counters: measured offline (rocprofv3 --pmc passes), see profiles/
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_f_k3_traffic.json")))
            if tj["bytes_per_gpu"] == n and not args.file and args.workload == "synth-text":
                traffic = tj["traffic_bytes_corrected"]
        except Exception:
            pass
        roof = {"bound": "hbm", "kernel": "K3 interval-count (k3_count2_kernel + k3_tiles_kernel<write> for wide rounds, k3_small_kernel for narrow ones, k3_tail_kernel / k3_dfs_kernel for the ends; all rounds of one compression = one launch unit)",
                "achieved": round(alg_bytes / k3_s / 1e9, 2) if k3_s > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(alg_bytes / k3_s / 1e9 / HBM_PEAK_GBS, 5) if k3_s > 0 else None,
                "traffic": traffic, "algorithmic_bytes": alg_bytes, "k3_ms_per_step": round(k3_s * 1e3, 3),
                "k3_launches_per_step": st["k3_launches"]}
        if dist is not None and gathered[0] is not None:
            # the gathered per-block streams form the multi-block container (bce_amd/container.py)
            from bce_amd import container
            blob = container.pack_blocks(gathered[0], [n] * n_gpus)
            assert container.unpack_blocks(blob)[0][0] == arch
        out = {
            "metric": "MB/s compressed", "value": round(n_gpus * n * steps / dt / 1e6, 3), "unit": "MB/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/u32 integer",
            "data": "synthetic" if not args.file else "file",
            "config": {"workload": workload, "bytes_per_gpu": n, "coder_config": coder_config,
                       "sharding": "one independent block per GPU, RCCL gather of coded streams to rank 0" if n_gpus > 1 else "single block"},
            "archive_bytes": len(arch), "archive_sha256": hashlib.sha256(arch).hexdigest(),
            "ratio": round(len(arch) / n, 5),
            "roofline": roof,
            "breakdown_s": {k: round(st[k], 4) for k in ("t_load", "t_bwt", "t_planes", "t_enum", "t_model", "t_coder", "t_coder_busy")},
            "counts": {"nodes": st["nodes"], "symbols": st["symbols"], "rounds": st["rounds"], "sort_rounds": st["sort_rounds"], "flushes": st["flushes"]},
        }
        if n_gpus == 1 and not args.no_decode and n <= 1_000_000_000:
            # untimed: the archive of the last step through the GPU-assisted decoder (`bce -d`), compared with the input
            t0 = time.perf_counter()
            back = bce_amd.decompress_device(arch, ctx=ctx)
            td = time.perf_counter() - t0
            out["decode"] = {"value": round(n / td / 1e6, 3), "unit": "MB/s", "seconds": round(td, 3),
                             "roundtrip_identical": bool(len(back) == n and hashlib.sha256(back).digest() == hashlib.sha256(data.tobytes()).digest()),
                             "note": "bce_hip_decompress_device: GPU passes + 8 host range decoders; not part of `value`"}
        if n_gpus == 1 and not args.no_cpu:
            cb = cpu_baseline(data, min(args.cpu_sample, n))
            out["cpu_baseline"] = cb
            out["gpu_over_cpu"] = round(out["value"] / cb["value"], 2)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
