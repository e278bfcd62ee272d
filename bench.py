#!/usr/bin/env python3
"""bench.py -- headline benchmark of the `bce -c` hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step = one full compression (K1 rotation sort/BWT, K2 planes, K3 enumeration, K4 model, host range
coders, framing) of one input block per GPU, input already resident in HBM, archive bytes ready on the
host at the end.  Rank 0 prints ONE JSON line.

N = 1: BASELINE.json configs[1], enwik8 (10^8 bytes).  The corpus is not available offline; if a file is given
with --file (or $BCE_BENCH_FILE) it is used, otherwise the stand-in is synth-text v1 (SURVEY 8c generator) at
10^8 bytes, seed 1.

N > 1 (launched by torch.distributed.run, one rank per GPU): BASELINE.json configs[3] as stated -- ONE input,
enwik9-sized (synth-text v1 seed 1, 10^9 bytes; --file if given), cut into N contiguous blocks
(sharding.block_range); every rank compresses its block with no data-path collective and the coded streams
are gathered to rank 0 over RCCL inside the timed step, where they form the BCEM container `bce -cN` writes.
The total work is fixed (strong scaling): value = 10^9 B x steps / time.  Every gathered block is compared on
rank 0 with the ORACLE's archive of that block alone (tests/golden/oracle_fullsize.json, synth-text-1e9-bRofN:
the reference has one block per archive, bce.cpp:1151-1157): `oracle_golden_blocks`.

Besides the headline the line carries (N = 1 only, all untimed with respect to `value`):
  workloads         the same measurement on harder inputs of the same size class: the natural and binary corpora built
                    from this image's own files (tools/make_corpus.py, make_binary_corpus.py) and synth-rand
  value_end_to_end  the headline workload from a HOST buffer (H2D copy inside the timed region)
  cpu_baseline      the oracle on a bounded sample, and the GPU path on the SAME sample: parity_sample_identical
  oracle_golden     whether the headline archive equals the oracle's (tests/golden/oracle_fullsize.json)
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import bce_amd  # noqa: E402
from bce_amd import sharding  # noqa: E402

REF_ENCODE_RATIO = 1.82  # reference encode-stage seconds / oracle encode-stage seconds (BASELINE.md, round 3 calibration)
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
TRAFFIC_FILE = "r05_k3_traffic.json"          # tools/profile.sh + tools/pmc_summary.py (per-launch-unit HBM bytes of K3)
TRAFFIC_FILE_1E9 = "r05_k3_traffic_1e9.json"
FAMILY_FILE = "r05_family_traffic.json"       # the same passes, summed per kernel family (K1, K2, K3, K4)
K3_KERNELS = ("K3 interval-count (k3_count2_kernel + k3_tiles_kernel<write> for wide rounds, k3_small_kernel for narrow ones, "
              "k3_local_kernel / k3_dfs_kernel / k3_tail_kernel for the ends; all rounds of one compression = one launch unit)")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=100_000_000, help="N = 1: input bytes (enwik8 = 10^8)")
    ap.add_argument("--total-size", type=int, default=1_000_000_000,
                    help="N > 1: bytes of the ONE input that is cut into N contiguous blocks (enwik9 = 10^9)")
    ap.add_argument("--container-out", default=None, help="N > 1: rank 0 writes the BCEM container of the last step here (what `bce -d` decodes)")
    ap.add_argument("--workload", default="synth-text", choices=["synth-text", "synth-rand"])
    ap.add_argument("--file", default=os.environ.get("BCE_BENCH_FILE"))
    ap.add_argument("--cpu-sample", type=int, default=48 << 20, help="bytes of the workload the CPU baseline compresses")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-stream", action="store_true", help="skip the two-context stream leg (N=1 only)")
    ap.add_argument("--stream-contexts", type=int, default=4, help="contexts of the stream leg")
    ap.add_argument("--stream-steps", type=int, default=0, help="inputs of the stream leg (default: max(12, --steps); 12 for the extra workloads)")
    ap.add_argument("--no-decode", action="store_true", help="skip the untimed decode-and-compare leg (N=1 only)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-buffer (H2D inside) leg")
    ap.add_argument("--no-cli", action="store_true", help="skip the cold `bce -c / -d / -s` child-process leg (N=1 only)")
    ap.add_argument("--no-workloads", action="store_true", help="skip the extra workloads (natural / binary corpus, synth-rand)")
    ap.add_argument("--no-big", action="store_true", help="skip the 10^9-byte and the scanned 2x10^8-byte workloads (BASELINE configs 3 and 5)")
    ap.add_argument("--scan-config", action="store_true",
                    help="BASELINE config 5: run `bce -s` on the input first (untimed), compress with the scanned table")
    ap.add_argument("--single-archive", action="store_true",
                    help="N > 1: every rank holds the SAME input and codes its share of the eight planes; rank 0 ends with the ONE archive "
                         "`bce -c` writes (sharding.single_archive; whole-file compressed size, no throughput gain). Default: one block per GPU")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the real multi-GPU path); gloo = rehearsal of the N>1 control flow on one GPU")
    return ap.parse_args()


def make_input(args, rank, world):
    """-> (this rank's bytes, description, the whole input or None).  N > 1 without --single-archive: ONE input cut into
    `world` contiguous blocks (BASELINE configs[3]); every rank builds the whole input (the generator is sequential) and
    keeps its block; rank 0 keeps the whole for the per-block checks."""
    gen = bce_amd.synth_text if args.workload == "synth-text" else bce_amd.synth_rand
    one = getattr(args, "single_archive", False)
    if world > 1 and not one:
        if args.file:
            whole = np.fromfile(args.file, dtype=np.uint8)
            desc = "file:%s[%d B]" % (os.path.basename(args.file), len(whole))
        else:
            whole = gen(1, args.total_size)
            desc = "%s-v1 seed 1, %d B (enwik9-sized stand-in: the corpus is not available offline)" % (args.workload, args.total_size)
        lo, hi = sharding.block_range(len(whole), world, rank)
        return np.ascontiguousarray(whole[lo:hi]), desc + ", cut into %d contiguous blocks (this rank: [%d, %d))" % (world, lo, hi), (whole if rank == 0 else None)
    if args.file:
        data = np.fromfile(args.file, dtype=np.uint8)
        return data, "file:%s[%d B, sha256 %s]" % (os.path.basename(args.file), len(data), hashlib.sha256(data.tobytes()).hexdigest()[:16]), None
    return gen(1, args.size), "%s-v1 seed 1, %d B (enwik8-sized stand-in: the corpus is not available offline)" % (args.workload, args.size), None


def golden_table():
    """Oracle-produced archive hashes at full size (tools/make_oracle_golden.py), keyed by input sha256."""
    try:
        with open(os.path.join(ROOT, "tests", "golden", "oracle_fullsize.json")) as f:
            return {v["input_sha256"]: v for v in json.load(f)["vectors"]}
    except Exception:
        return {}


def golden_verdict(table, data, arch):
    """"identical" / "DIFFERENT" against the oracle's archive of the same input, None when the oracle never ran on it."""
    v = table.get(hashlib.sha256(data.tobytes()).hexdigest())
    if v is None:
        return None
    ok = len(arch) == v["archive_bytes"] and hashlib.sha256(arch).hexdigest() == v["archive_sha256"]
    return "identical" if ok else "DIFFERENT"


def roofline(n, sts):
    st = sts[-1]
    k3_s = sum(s["k3_ms"] for s in sts) / len(sts) / 1e3
    alg_bytes = 384.0 * n + 16.0 * st["symbols"]     # 48 B/node x 8n nodes + 16 B/symbol
    return {"bound": "hbm", "kernel": K3_KERNELS,
            "achieved": round(alg_bytes / k3_s / 1e9, 2) if k3_s > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(alg_bytes / k3_s / 1e9 / HBM_PEAK_GBS, 5) if k3_s > 0 else None,
            "traffic": None, "algorithmic_bytes": alg_bytes, "k3_ms_per_step": round(k3_s * 1e3, 3),
            "k3_launches_per_step": st["k3_launches"]}


def roofline_families(n, sts, arch_sha):
    """SURVEY 8d's other kernel families beside K3, per compression: algorithmic bytes (K1 6n, K2 17n, K4 22 x symbols) over the
    family's time in the timed steps -- K1 / K2: host seconds around the stage, which ends with a sync (t_bwt, t_planes: launch
    gaps and read-backs included); K4: HIP events around every flush's kernels + its device-to-host copy (t_model) -- and the
    counter bytes of profiles/FAMILY_FILE (static: separate rocprofv3 --pmc passes on this workload) when that file is this
    archive's."""
    st = sts[-1]
    avg = lambda k: sum(s[k] for s in sts) / len(sts)
    spec = {"K1": (6.0 * n, avg("t_bwt"), "k1_* + radix-sort kernels: rotation sort + BWT (File::rotate / divbwt)"),
            "K2": (17.0 * n, avg("t_planes"), "k2_*: 8 planes + rank granules (RankFile)"),
            "K4": (22.0 * st["symbols"], avg("t_model_kernels"), "k4_* + the slot sort (AdaptiveCoder::set, model half); HIP events around every flush's kernels, "
                                                                "without the device-to-host copies of the records (with them: breakdown_s.t_model)")}
    fam_traffic = {}
    try:
        fj = json.load(open(os.path.join(ROOT, "profiles", FAMILY_FILE)))
        if fj.get("archive_sha256") == arch_sha and fj.get("bytes_per_gpu") == n:
            fam_traffic = fj["families"]
    except Exception:
        pass
    out = {}
    for k, (alg, sec, what) in spec.items():
        t = fam_traffic.get(k)
        out[k] = {"bound": "hbm", "kernels": what, "algorithmic_bytes": alg, "ms_per_step": round(sec * 1e3, 3),
                  "achieved": round(alg / sec / 1e9, 2) if sec > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": round(alg / sec / 1e9 / HBM_PEAK_GBS, 5) if sec > 0 else None,
                  "traffic": t["traffic_bytes"] if t else None,
                  "traffic_over_algorithmic": t["over_algorithmic"] if t else None,
                  "traffic_source": ("profiles/%s (static)" % FAMILY_FILE) if t else None}
    return out


def timed_steps(ctx, t_in, n, steps, warmup, config=None):
    for _ in range(warmup):
        bce_amd.compress_device(t_in.data_ptr(), n, config=config, ctx=ctx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sts, arch = [], None
    for _ in range(steps):
        arch, st = bce_amd.compress_device(t_in.data_ptr(), n, config=config, ctx=ctx)
        sts.append(st)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, arch, sts


def stream_leg(pool, t_in, n, steps, arch, config=None):
    """A STREAM of such inputs (files, the blocks of `bce -cN`) through the pool's gated contexts: the enumerations take
    turns, K1 of one input and the coding tail of another run beside them.  Same input, same archive, every step."""
    nctx = len(pool.ctxs)
    pool.compress_many([(t_in.data_ptr(), n)] * nctx, config=config, on_device=True)     # warm-up: one input per context
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = pool.compress_many([(t_in.data_ptr(), n)] * steps, config=config, on_device=True, with_stats=True)
    ts = time.perf_counter() - t0
    return {"value": round(n * steps / ts / 1e6, 3), "unit": "MB/s", "steps": steps, "ms_per_step": round(ts / steps * 1e3, 2), "contexts": nctx,
            "identical_to_headline": bool(all(bytes(a) == bytes(arch) for a, _ in res)),
            "k3_ms_per_step": round(sum(s_["k3_ms"] for _, s_ in res) / steps, 3),
            "coder_busy_ms": round(max(s_["t_coder_busy"] for _, s_ in res) * 1e3, 1)}


def timed_decode(arch, ctx, data):
    """Seconds of ONE bce_hip_decompress_device call, archive bytes -> a caller's buffer that exists already (allocated and
    touched before the clock starts, as the encoder's input is resident before its clock starts), and whether the bytes
    are the input's."""
    n = len(data)
    buf = np.zeros(n + 64, dtype=np.uint8)
    r0 = bce_amd.stats_of(ctx)["dec_restarts"]
    t0 = time.perf_counter()
    got = bce_amd.decompress_device(arch, ctx=ctx, out=buf)
    td = time.perf_counter() - t0
    timed_decode.restarts = int(bce_amd.stats_of(ctx)["dec_restarts"] - r0)     # decodes started again with larger node lists (0 in a warm context)
    return td, bool(got == n and np.array_equal(buf[:n], np.asarray(data).reshape(-1)))


def extra_workloads(ctx, dev, table, pool=None, stream_steps=12, decode=True):
    """The same measurement (input resident in HBM, 2 steps after 1 warm-up) on harder inputs."""
    out = []
    specs = [("natural corpus v2 (tools/make_corpus.py: this image's Python sources + ROCm headers; long repeats, ~2 M rounds)", "natural", 100_000_000),
             ("binary corpus (tools/make_binary_corpus.py: this image's shared libraries; tables, zero runs)", "binary", 100_000_000),
             ("synth-rand v1 seed 1 (6 symbols per byte)", "synth-rand", 32 << 20)]
    for desc, kind, n in specs:
        try:
            if kind == "synth-rand":
                data = bce_amd.synth_rand(1, n)
            else:
                path = "/tmp/bce_%s_%d.bin" % (kind, n)
                if not (os.path.exists(path) and os.path.getsize(path) == n):
                    tool = "make_corpus.py" if kind == "natural" else "make_binary_corpus.py"
                    subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), "--out", path, "--size", str(n)],
                                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                data = np.fromfile(path, dtype=np.uint8)
            t_in = torch.from_numpy(data).to(dev)
            torch.cuda.synchronize()
            dt, arch, sts = timed_steps(ctx, t_in, n, 2, 1)
            stream = None
            if pool is not None:
                try:
                    stream = stream_leg(pool, t_in, n, stream_steps, arch)
                except Exception as e:
                    stream = {"error": "%s: %s" % (type(e).__name__, e)}
            del t_in
            r = roofline(n, sts)
            st = sts[-1]
            dec = None
            if decode:
                try:
                    timed_decode(arch, ctx, data)               # (warm-up: this workload's decoder buffers, as for the headline)
                    td, same = timed_decode(arch, ctx, data)
                    dec = {"seconds": round(td, 3), "value": round(n / td / 1e6, 2), "unit": "MB/s", "roundtrip_identical": same,
                           "list_restarts": timed_decode.restarts}
                except Exception as e:
                    dec = {"error": "%s: %s" % (type(e).__name__, e)}
            out.append({"workload": desc, "decode": dec, "bytes": n, "input_sha256": hashlib.sha256(data.tobytes()).hexdigest()[:16],
                        "value": round(n * 2 / dt / 1e6, 3), "unit": "MB/s", "ms_per_step": round(dt / 2 * 1e3, 2),
                        "k3_ms": r["k3_ms_per_step"], "roofline_frac": r["frac"], "rounds": st["rounds"], "symbols": st["symbols"],
                        "sort_rounds": st["sort_rounds"], "k1_ms": round(st["t_bwt"] * 1e3, 2), "k4_ms": round(st["t_model"] * 1e3, 2),
                        "coder_busy_ms": round(st["t_coder_busy"] * 1e3, 2),
                        "archive_bytes": len(arch), "archive_sha256": hashlib.sha256(arch).hexdigest(),
                        "oracle_golden": golden_verdict(table, data, arch), "stream": stream})
        except Exception as e:   # a corpus that cannot be built on this box must not take the headline down
            out.append({"workload": desc, "error": "%s: %s" % (type(e).__name__, e)})
    return out


def big_workload(local, dev, table, n=1_000_000_000, stream_contexts=3):
    """BASELINE configs[2] stand-in: 10^9 bytes of synth-text v1 seed 1 on ONE GPU, everything resident in HBM (SA and
    rank arrays 36 n, node lists, symbol records).  1 warm-up + 1 timed step in a context of its own (closed afterwards:
    its buffers are ~70 GB).  The archive is pinned to the ORACLE's (tests/golden/oracle_fullsize.json, synth-text-1e9:
    the only size at which bits = 3, 4 of get_context wrap, bce.cpp:674)."""
    desc = "synth-text v1 seed 1, 10^9 B (enwik9-sized stand-in, BASELINE configs[2])"
    try:
        data = bce_amd.synth_text(1, n)
        t_in = torch.from_numpy(data).to(dev)
        torch.cuda.synchronize()
        ctx = bce_amd.api._Ctx(local)
        try:
            dt, arch, sts = timed_steps(ctx, t_in, n, 1, 1)
        finally:
            ctx.close()
        # a stream of such inputs through three gated contexts (~86 GB each: what fits beside the input; a context that runs out
        # of device memory gives its idle stages' buffers back, api.hip ctx_trim)
        stream = None
        if stream_contexts:
            try:
                with bce_amd.ContextPool(stream_contexts, local) as pool:
                    stream = stream_leg(pool, t_in, n, 2 * stream_contexts, arch)
            except Exception as e:
                stream = {"error": "%s: %s" % (type(e).__name__, e)}
        del t_in
        torch.cuda.empty_cache()
        r = roofline(n, sts)
        st = sts[-1]
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_FILE_1E9)))
            if tj["bytes_per_gpu"] == n and tj.get("archive_sha256") == hashlib.sha256(arch).hexdigest():
                traffic = {"bytes": tj["traffic_bytes"], "raw_counters": tj["traffic_bytes_raw"], "algorithmic": r["algorithmic_bytes"],
                           "over_algorithmic": round(tj["traffic_bytes"] / r["algorithmic_bytes"], 3),
                           "source": "profiles/%s (static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes at this size; read-side factor %s)" % (TRAFFIC_FILE_1E9, tj.get("fetch_factor"))}
        except Exception:
            pass
        return {"workload": desc, "bytes": n, "input_sha256": hashlib.sha256(data.tobytes()).hexdigest()[:16], "steps": 1, "warmup": 1,
                "roofline_traffic": traffic,
                "value": round(n / dt / 1e6, 3), "unit": "MB/s", "ms_per_step": round(dt * 1e3, 2),
                "k3_ms": r["k3_ms_per_step"], "roofline_frac": r["frac"], "roofline_achieved_GBs": r["achieved"],
                "rounds": st["rounds"], "symbols": st["symbols"], "sort_rounds": st["sort_rounds"],
                "k1_ms": round(st["t_bwt"] * 1e3, 2), "k2_ms": round(st["t_planes"] * 1e3, 2), "k4_ms": round(st["t_model"] * 1e3, 2),
                "coder_busy_ms": round(st["t_coder_busy"] * 1e3, 2),
                "archive_bytes": len(arch), "archive_sha256": hashlib.sha256(arch).hexdigest(),
                "oracle_golden": golden_verdict(table, data, arch), "stream": stream}
    except Exception as e:
        return {"workload": desc, "error": "%s: %s" % (type(e).__name__, e)}


def scanned_workload(local, dev, table, n=200_000_000):
    """BASELINE configs[4] stand-in (Silesia-sized, mixed content, tuned AdaptiveCoder): natural corpus || binary corpus,
    2 x 10^8 bytes; `bce -s` first (GPU enumeration in scan mode + the host ScanSet, timed on its own), then `bce -c` with
    that table.  Both the table and the archive are pinned to the oracle's (mixed-2e8-scanned)."""
    desc = "natural corpus || binary corpus, 2x10^8 B, table scanned by bce -s (Silesia-sized stand-in, BASELINE configs[4])"
    try:
        parts = []
        for kind, m in (("natural", n // 2), ("binary", n - n // 2)):
            path = "/tmp/bce_%s_%d.bin" % (kind, m)
            if not (os.path.exists(path) and os.path.getsize(path) == m):
                tool = "make_corpus.py" if kind == "natural" else "make_binary_corpus.py"
                subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), "--out", path, "--size", str(m)],
                               check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            parts.append(np.fromfile(path, dtype=np.uint8))
        data = np.concatenate(parts)
        del parts
        sha = hashlib.sha256(data.tobytes()).hexdigest()
        gold = table.get(sha)
        t0 = time.perf_counter()
        config, _sizes = bce_amd.scan(data, device=local)
        t_scan = time.perf_counter() - t0
        t_in = torch.from_numpy(data).to(dev)
        torch.cuda.synchronize()
        ctx = bce_amd.api._Ctx(local)
        try:
            dt, arch, sts = timed_steps(ctx, t_in, n, 2, 1, config=bytes(config))
        finally:
            ctx.close()
        del t_in
        torch.cuda.empty_cache()
        r = roofline(n, sts)
        st = sts[-1]
        ok = None
        if gold is not None:
            ok = ("identical" if len(arch) == gold["archive_bytes"] and hashlib.sha256(arch).hexdigest() == gold["archive_sha256"] else "DIFFERENT")
        return {"workload": desc, "bytes": n, "input_sha256": sha[:16],
                "scan_seconds": round(t_scan, 3), "scan_MBps": round(n / t_scan / 1e6, 2),
                "config_sha256": hashlib.sha256(bytes(config)).hexdigest(),
                "config_equals_oracle_scan": (bytes(config).hex() == gold["config_hex"]) if gold is not None and "config_hex" in gold else None,
                "value": round(n * 2 / dt / 1e6, 3), "unit": "MB/s", "ms_per_step": round(dt / 2 * 1e3, 2),
                "k3_ms": r["k3_ms_per_step"], "roofline_frac": r["frac"], "rounds": st["rounds"], "symbols": st["symbols"],
                "sort_rounds": st["sort_rounds"], "k1_ms": round(st["t_bwt"] * 1e3, 2), "k4_ms": round(st["t_model"] * 1e3, 2),
                "coder_busy_ms": round(st["t_coder_busy"] * 1e3, 2),
                "archive_bytes": len(arch), "archive_sha256": hashlib.sha256(arch).hexdigest(), "oracle_golden": ok}
    except Exception as e:
        return {"workload": desc, "error": "%s: %s" % (type(e).__name__, e)}


def cli_leg(data):
    """Runs FIRST, before this process creates any GPU context of its own (a second process' idle context on the device
    made the child's 10^8-byte run take 0.6 s instead of 0.27).  -> (dict, the archive the CLI wrote).
    What a drop-in CLI user sees (`n / wall seconds of -c`, file read -> archive on disk; the reference times
    its whole -c branch, bce.cpp:1404,1419-1422): wall seconds of `bce_amd/bin/bce` in a FRESH child process each time --
    process start, HIP runtime initialisation, every allocation of a cold context, file read and write included -- for -c
    (5 runs), -d and -s on the headline input, and -c on a 1000-byte file (the fixed cost).  The archive the CLI writes must
    be the headline archive, and -d must give the input back."""
    import statistics
    import tempfile
    exe = os.path.join(ROOT, "bce_amd", "bin", "bce")
    if not os.path.exists(exe):
        return {"error": "bce_amd/bin/bce has not been built"}, None
    n = len(data)
    with tempfile.TemporaryDirectory() as td:
        fin, farc, fback, fcfg, fsmall = (os.path.join(td, x) for x in ("in.bin", "out.bce", "back.bin", "c.bcc", "small.bin"))
        np.asarray(data).tofile(fin)
        np.asarray(data[:1000]).tofile(fsmall)

        def wall(args):
            time.sleep(0.4)       # (the driver tears the previous process' 18 GB of device memory down in the background; a user's one run does not queue behind one)
            t0 = time.perf_counter()
            r = subprocess.run([exe] + args, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            return time.perf_counter() - t0, r.returncode

        tc = sorted(wall(["-c", farc, fin]) for _ in range(5))
        cli_arch = open(farc, "rb").read()
        td_, rc_d = wall(["-d", fback, farc])
        back = np.fromfile(fback, dtype=np.uint8) if rc_d == 0 else None
        ts_, rc_s = wall(["-s", fcfg, fin])
        tsm = sorted(wall(["-c", farc + ".small", fsmall]) for _ in range(5))
        return {"compress_seconds": round(tc[0][0], 3), "compress_seconds_median": round(statistics.median(t for t, _ in tc), 3),
                "value": round(n / tc[0][0] / 1e6, 2), "value_median": round(n / statistics.median(t for t, _ in tc) / 1e6, 2), "unit": "MB/s",
                "archive_identical_to_headline": None, "all_runs_ok": all(rc == 0 for _, rc in tc),
                "decompress_seconds": round(td_, 3), "decompress_roundtrip_identical": bool(back is not None and np.array_equal(back, np.asarray(data).reshape(-1))),
                "scan_seconds": round(ts_, 3), "scan_ok": rc_s == 0 and os.path.getsize(fcfg) == 288,
                "compress_1000_B_seconds": round(tsm[0][0], 3), "compress_1000_B_seconds_median": round(statistics.median(t for t, _ in tsm), 3),
                "note": "wall seconds of a fresh `bce` process per call (min of 5 and median for -c): process start, HIP initialisation, cold-context "
                        "allocations, file read and archive write included; `value` of the line excludes all of these (input resident in HBM, warm context)"}, cli_arch


def cpu_baseline(data, sample_bytes, ctx, dev):
    """The oracle (bit-exact CPU restatement of bce -c) timed on this host on a bounded sample: single thread (the
    reference built without OpenMP) and 8 threads (its OpenMP build: one thread per plane, joined every round,
    bce.cpp:1250-1252; suffix sort and plane build stay serial there too).  The faster one is `value`.  The GPU path
    compresses the SAME sample: the two archives must be byte-identical, and gpu_over_cpu compares like with like."""
    import oracle
    oracle.build()
    sample = np.ascontiguousarray(data[:sample_bytes])
    runs, stages = {}, {}
    for threads in (1, 8):
        oracle.set_threads(threads)
        t0 = time.time()
        arch = oracle.compress(sample)
        runs[threads] = (time.time() - t0, len(arch), hashlib.sha256(arch).hexdigest())
        stages[threads] = oracle.stage_seconds()
    oracle.set_threads(1)
    assert runs[1][2] == runs[8][2]
    best = min(runs, key=lambda t: runs[t][0])
    n = len(sample)
    t_in = torch.from_numpy(sample).to(dev)
    torch.cuda.synchronize()
    dt, garch, _ = timed_steps(ctx, t_in, n, 2, 1)
    identical = bool(bytes(garch) == arch)
    cpu_v = n / runs[best][0] / 1e6
    gpu_v = n * 2 / dt / 1e6
    return {"value": round(cpu_v, 3), "unit": "MB/s", "cores": best, "kind": "port",
            "sample": "first %d B of the workload, oracle/bce_oracle.c: 1 thread %.1f s (%.2f MB/s), 8 OpenMP threads %.1f s (%.2f MB/s), archive %d B" % (
                n, runs[1][0], n / runs[1][0] / 1e6, runs[8][0], n / runs[8][0] / 1e6, runs[1][1]),
            "single_thread_value": round(n / runs[1][0] / 1e6, 3),
            "stage_seconds": {k: round(v, 3) for k, v in stages[best].items()},
            # BASELINE.md calibration: the reference's encode stage takes 1.82x the oracle's (2.69 s vs 1.48 s, 8 MiB synth-text, 1 thread)
            "reference_equivalent_value": round(n / (runs[best][0] + (REF_ENCODE_RATIO - 1.0) * stages[best]["encode"]) / 1e6, 3),
            "reference_equivalent_note": "oracle time with its encode stage scaled by %.2f = bce.cpp's `Encode:` timer / the oracle's, calibrated in the build container (BASELINE.md)" % REF_ENCODE_RATIO,
            "host_cpus": os.cpu_count(),
            "parity_sample_identical": identical,
            "gpu_value_same_sample": round(gpu_v, 3),
            "gpu_over_cpu_same_sample": round(gpu_v / cpu_v, 2)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        print("bench.py: --gpus %d ignored, WORLD_SIZE is %d (one rank per GPU)" % (args.gpus, world), file=sys.stderr)
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
    cli, cli_arch = None, None
    if world == 1 and not args.no_cli:
        # the cold `bce` child processes run before this process touches the GPU (see cli_leg)
        try:
            cli_data = np.fromfile(args.file, dtype=np.uint8) if args.file else (bce_amd.synth_text if args.workload == "synth-text" else bce_amd.synth_rand)(1, args.size)
            cli, cli_arch = cli_leg(cli_data)
            del cli_data
        except Exception as e:
            cli = {"error": "%s: %s" % (type(e).__name__, e)}
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    comm_dev = dev if args.backend == "nccl" else torch.device("cpu")
    pinned = sharding.pin_to_local_numa(local)   # the rank's 8 coder threads stay on the GPU's NUMA node

    data, workload, whole = make_input(args, rank, world)
    n = len(data)
    sizes = [n]
    if dist is not None:
        total_n = args.total_size if not args.file else None
        if args.single_archive:
            sizes = [n] * world
        else:
            szt = torch.tensor([n], dtype=torch.int64, device=comm_dev)
            allsz = [torch.zeros_like(szt) for _ in range(world)]
            dist.all_gather(allsz, szt)
            sizes = [int(x.item()) for x in allsz]
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
    comm_s = [0.0]
    # The archive is laid out in a buffer that exists before the clock starts (the C ABI writes into the caller's buffer; the
    # input is resident before the clock starts, too): pinned, so that the gather sends from it without a staging copy.
    arch_t = torch.empty(n + n // 8 + 65536, dtype=torch.uint8, pin_memory=True)
    arch_np = arch_t.numpy()

    one = bool(args.single_archive and dist is not None)
    one_arch = [None]

    def step():
        if one:
            tm = {}
            a1 = sharding.single_archive(ctx, dist, comm_dev, device_ptr=t_in.data_ptr(), n=n, config=config, timings=tm)
            st1 = bce_amd.api.stats_of(ctx)
            comm_s[0] += tm.get("gather_s", 0.0)      # streams out, gather, streams in, archive laid out: what the step costs beyond this rank's own compression
            if a1 is not None:
                one_arch[0] = a1
            return a1, st1
        arch, st = bce_amd.compress_device(t_in.data_ptr(), n, config=config, ctx=ctx, out=arch_np)
        if dist is not None:
            # RCCL gather of the per-block coded streams to rank 0 (size exchange, then padded gather)
            tg = time.perf_counter()
            src = arch_t[:len(arch)] if isinstance(arch, memoryview) else arch
            gathered[0] = sharding.gather_streams(src, dist, comm_dev, copy=False, direct=isinstance(arch, memoryview))   # (views of the receive buffer: packed after the loop)
            comm_s[0] += time.perf_counter() - tg
        return arch, st

    for _ in range(args.warmup):
        step()
    barrier()
    comm_s[0] = 0.0
    t0 = time.perf_counter()
    sts = []
    arch = None
    for _ in range(args.steps):
        arch, st = step()
        sts.append(st)
    barrier()
    dt_local = time.perf_counter() - t0
    if one:
        arch = one_arch[0] if one_arch[0] is not None else b""
    arch = bytes(arch)                          # (out of the reused buffer: later legs compress again)
    tt = torch.tensor([dt_local], dtype=torch.float64, device=comm_dev if dist is not None else dev)
    per_rank = [dt_local]
    if dist is not None:
        allt = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)
        per_rank = [float(x.item()) for x in allt]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    # what a SCALE line needs to attribute a loss: per-rank gather time (inside the step), busiest coder thread, CPU set
    mine = {"gather_ms": round(comm_s[0] / max(1, args.steps) * 1e3, 3), "coder_busy_ms": round(sts[-1]["t_coder_busy"] * 1e3, 2) if sts else None,
            "k1_ms": round(sts[-1]["t_bwt"] * 1e3, 2) if sts else None, "cpus": len(os.sched_getaffinity(0)),
            "cpu_set": sharding.format_cpulist(os.sched_getaffinity(0)), "numa_pinned": pinned is not None}
    per_rank_info = [mine]
    if dist is not None:
        per_rank_info = [None] * world
        dist.all_gather_object(per_rank_info, mine)

    if rank == 0:
        steps = max(1, args.steps)
        st = sts[-1]
        roof = roofline(n, sts)
        table = golden_table()
        # HBM bytes from the PMC counters are collected offline (tools/profile.sh: separate rocprofv3 --pmc passes), not in this
        # run; the static file is only quoted when it was measured on THIS workload with THIS result (same archive hash)
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)))
            if tj["bytes_per_gpu"] == n and not args.file and args.workload == "synth-text" and n_gpus == 1:
                if tj.get("archive_sha256") == hashlib.sha256(arch).hexdigest():
                    roof["traffic"] = tj["traffic_bytes"]
                    roof["traffic_raw_counters"] = tj["traffic_bytes_raw"]
                    roof["traffic_source"] = ("profiles/%s (static: measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on this workload, not in this run; "
                                              "read-side factor %s from profiles/%s)" % (TRAFFIC_FILE, tj.get("fetch_factor"), tj.get("calibration_file")))
                else:
                    roof["traffic_source"] = "profiles/%s REFUSED: its archive_sha256 is not this run's" % TRAFFIC_FILE
        except Exception:
            pass
        blocks_verdict, blocks_detail, container_bytes = None, None, None
        if dist is not None and gathered[0] is not None and not one:
            # the gathered per-block streams form the multi-block container (bce_amd/container.py, = what `bce -cN` writes);
            # every block is checked against the ORACLE's archive of that block alone (rank 0 holds the whole input)
            from bce_amd import container
            blob = container.pack_blocks(gathered[0], sizes)
            assert bytes(container.unpack_blocks(blob)[0][0]) == bytes(arch)
            container_bytes = len(blob)
            if args.container_out:
                with open(args.container_out, "wb") as f:
                    f.write(blob)
            del blob
            if config is None and whole is not None:
                blocks_detail = []
                for r in range(world):
                    lo, hi = sharding.block_range(len(whole), world, r)
                    blocks_detail.append(golden_verdict(table, whole[lo:hi], gathered[0][r]))
                blocks_verdict = ("identical" if all(v == "identical" for v in blocks_detail) else
                                  "DIFFERENT" if any(v == "DIFFERENT" for v in blocks_detail) else None)
        total_bytes = n if (one or dist is None) else sum(sizes)
        total_arch = len(arch) if (one or dist is None or gathered[0] is None) else int(sum(len(g) for g in gathered[0]))
        out = {
            "metric": "MB/s compressed", "value": round(total_bytes * steps / dt / 1e6, 3), "unit": "MB/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / steps * 1e3, 2),
            "higher_is_better": True, "scaling": "strong" if n_gpus > 1 else "weak", "vs_baseline": None, "dtype": "u8/u32 integer",
            "data": "synthetic" if not args.file else "file",
            "config": {"workload": workload, "total_bytes": total_bytes, "bytes_per_gpu": sizes if n_gpus > 1 else n, "coder_config": coder_config,
                       "sharding": ("ONE input on every GPU, the eight plane coders shared out by plane, finished streams gathered to rank 0: one archive, the one `bce -c` writes"
                                    if one else "ONE input cut into %d contiguous blocks, one per GPU, no data-path collective; RCCL gather of the coded streams to rank 0 "
                                                "(BASELINE configs[3]: enwik9 block-sharded)" % n_gpus) if n_gpus > 1 else "single block"},
            "archive_bytes": total_arch, "archive_sha256": hashlib.sha256(arch).hexdigest(),
            "archive_sha256_of": "rank 0's block" if (n_gpus > 1 and not one) else "the archive",
            "oracle_golden": golden_verdict(table, data, arch) if config is None else None,
            "oracle_golden_blocks": blocks_verdict, "oracle_golden_per_block": blocks_detail, "container_bytes": container_bytes,
            "ratio": round(total_arch / total_bytes, 5),
            "roofline": roof,
            "roofline_families": roofline_families(n, sts, hashlib.sha256(arch).hexdigest()) if n_gpus == 1 else None,
            "breakdown_s": {k: round(st[k], 4) for k in ("t_load", "t_bwt", "t_planes", "t_enum", "t_model", "t_coder", "t_coder_busy")},
            "counts": {"nodes": st["nodes"], "symbols": st["symbols"], "rounds": st["rounds"], "sort_rounds": st["sort_rounds"], "flushes": st["flushes"]},
            "ms_per_step_per_rank": [round(t / steps * 1e3, 2) for t in per_rank],
            "per_rank": per_rank_info,
        }
        if dist is not None:
            out["comm"] = {"backend": dist.get_backend(), "world": dist.get_world_size(),
                           "collectives_per_step": "all_gather of 8-byte sizes + one padded gather to rank 0 (bce_amd/sharding.py)",
                           "gather_ms": max(r["gather_ms"] for r in per_rank_info),
                           "gather_ms_per_rank": [r["gather_ms"] for r in per_rank_info],
                           "gather_bytes": int(sum(len(g) for g in gathered[0])) if gathered[0] is not None else None,
                           "note": "gather_ms is inside ms_per_step; it includes waiting for the slowest rank's archive"}
        if n_gpus == 1 and not args.no_e2e:
            # SURVEY 8d's "file read -> archive bytes ready": the same workload from a (pageable) HOST buffer, H2D inside
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                a2 = bce_amd.compress(data, config=config, ctx=ctx)
            te = (time.perf_counter() - t0) / 2
            out["value_host_buffer_to_archive"] = round(n / te / 1e6, 3)      # SURVEY 8d's metric as ONE number beside `value` (see metric_scope)
            out["value_end_to_end"] = {"value": round(n / te / 1e6, 3), "unit": "MB/s", "ms_per_step": round(te * 1e3, 2),
                                       "identical_to_headline": bool(bytes(a2) == bytes(arch)),
                                       "note": "host buffer -> archive bytes on the host: the H2D copy of the input is inside the timed region (PCIe); not `value`"}
        if cli is not None:
            if "error" not in cli:
                cli["archive_identical_to_headline"] = bool(cli_arch == bytes(arch)) and cli.pop("all_runs_ok")
                out["value_cli_cold"] = cli["value"]
            out["cli"] = cli
        out["metric_scope"] = ("`value`: input resident in HBM when the clock starts, archive bytes on the host when it stops (this round's bench "
                               "contract: a host-buffer rate is never `value`).  SURVEY 8d's metric -- file / host buffer -> archive bytes, the scope of "
                               "the reference's own timer (bce.cpp:1404-1422) -- stands beside it as `value_host_buffer_to_archive` (warm context, H2D "
                               "inside) and `value_cli_cold` (a fresh `bce -c` process: start-up, allocations, file read and write inside)")
        pool = None
        if n_gpus == 1 and not args.no_stream:
            try:
                pool = bce_amd.ContextPool(max(1, args.stream_contexts), local)
                out["value_stream"] = stream_leg(pool, t_in, n, args.stream_steps or max(12, args.steps), arch, config)
                out["value_stream"]["note"] = ("throughput of a stream of inputs on one GPU: gated contexts take turns for the enumeration, K1 of one "
                                               "input and the coding tail of another run beside it; `value` stays the one-input-at-a-time figure")
            except Exception as e:   # the extra leg must not take the headline down
                out["value_stream"] = {"error": "%s: %s" % (type(e).__name__, e)}
                if pool is not None:
                    pool.close()
                pool = None
        if n_gpus == 1 and not args.no_decode and n <= 1_000_000_000:
            # untimed: the archive of the last step through the GPU-assisted decoder (`bce -d`), compared with the input.
            # (One decode first to warm the context's decoder buffers up, as the compression steps have their warm-up: on a
            # card whose memory an earlier process has just given back the driver clears it first, 27 ms per GB.)
            timed_decode(arch, ctx, data)
            td, same = timed_decode(arch, ctx, data)
            out["decode"] = {"value": round(n / td / 1e6, 3), "unit": "MB/s", "seconds": round(td, 3), "roundtrip_identical": same,
                             "list_restarts": timed_decode.restarts,
                             "note": "bce_hip_decompress_device(archive -> caller's buffer) in a warm context: GPU passes + 8 host range decoders; not part of `value`"}
        if n_gpus == 1 and not args.no_workloads and not args.file and args.workload == "synth-text":
            out["workloads"] = extra_workloads(ctx, dev, table, pool, args.stream_steps or 12, decode=not args.no_decode)
            if pool is not None:
                pool.close()
                pool = None
            if not args.no_big:
                out["workloads"].append(scanned_workload(local, dev, table))     # BASELINE configs[4] stand-in
                out["workloads"].append(big_workload(local, dev, table, stream_contexts=0 if args.no_stream else 3))   # BASELINE configs[2] stand-in (10^9 B)
        if pool is not None:
            pool.close()
        if "workloads" in out:
            # a changed image cannot shrink the parity set silently: workloads whose input has no oracle-made known answer on
            # this box (the natural / binary corpora are rebuilt from the image's own files) are counted
            missing = [w_.get("workload", "?")[:40] for w_ in out["workloads"] if w_.get("oracle_golden") is None]
            out["skipped_vectors"] = len(missing) + (1 if out["oracle_golden"] is None and config is None else 0)
            out["skipped_vector_names"] = missing
        if n_gpus == 1 and not args.no_cpu:
            cb = cpu_baseline(data, min(args.cpu_sample, n), ctx, dev)
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
