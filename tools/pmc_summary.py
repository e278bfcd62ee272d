#!/usr/bin/env python3
"""Summarise tools/profile.sh output: per-kernel FETCH_SIZE / WRITE_SIZE sums (one compression) and K3's HBM
traffic with the gfx950 read correction of MI355X_MICROARCH.md.   python tools/pmc_summary.py OUTDIR TAG [SUFFIX]"""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name.strip()


def sums(outdir, sub, counter):
    acc, calls = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(outdir, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] == counter:
                    k = short(row["Kernel_Name"])
                    acc[k] += float(row["Counter_Value"])
                    calls[k] += 1
    return acc, calls


def family_sums(outdir, sub, counter):
    """Counter sums per kernel FAMILY of one compression: K1 (k1_* and the radix-sort kernels dispatched before K2), K2 (k2_*),
    K3 (k3_*), K4 (k4_* and the radix-sort kernels dispatched after K2: the model's slot sort and the tail's tagged symbols)."""
    rows = []
    for f in glob.glob(os.path.join(outdir, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] == counter:
                    rows.append((int(row["Dispatch_Id"]), short(row["Kernel_Name"]), float(row["Counter_Value"])))
    rows.sort()
    k2_ids = [d for d, k, _ in rows if "k2_" in k]
    k2_first = min(k2_ids) if k2_ids else None
    acc, calls = defaultdict(float), defaultdict(int)
    for d, k, v in rows:
        fam = None
        if "k1_" in k or "rw_" in k: fam = "K1"              # (rw_*: the wide sort of 64-bit keys, K1's alone)
        elif "k2_" in k: fam = "K2"
        elif "k3_" in k or "kd_" in k: fam = "K3"              # (kd_*: the depth-first tail's helpers)
        elif "k4_" in k: fam = "K4"
        elif "rs_" in k or "radix" in k: fam = "K1" if (k2_first is not None and d < k2_first) else "K4"
        if fam:
            acc[fam] += v * 1024.0
            calls[fam] += 1
    return acc, calls


def main():
    outdir, tag = sys.argv[1], sys.argv[2]
    sfx = sys.argv[3] if len(sys.argv) > 3 else ""          # e.g. "_1e9": passes in fetch_1e9/ write_1e9/, log fetch_1e9.log
    fetch, calls = sums(outdir, "fetch" + sfx, "FETCH_SIZE")
    write, _ = sums(outdir, "write" + sfx, "WRITE_SIZE")
    rows = sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, 0) + write.get(k, 0)))
    with open(os.path.join(outdir, "%s_pmc_fetch_write%s.csv" % (tag, sfx)), "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)       # kernel names hold commas (template arguments): quoted
        w.writerow(["kernel", "dispatches", "FETCH_SIZE_KB_sum", "WRITE_SIZE_KB_sum"])
        for k in rows:
            w.writerow([k, calls.get(k, 0), round(fetch.get(k, 0.0), 1), round(write.get(k, 0.0), 1)])
    k3 = [k for k in rows if "k3_" in k]
    fr = sum(fetch.get(k, 0.0) for k in k3) * 1024.0
    wr = sum(write.get(k, 0.0) for k in k3) * 1024.0
    line = {}
    try:
        with open(os.path.join(outdir, "fetch%s.log" % sfx)) as f:
            line = json.loads([l for l in f if l.startswith("{")][-1])
    except Exception:
        pass
    # read-side factor: FETCH_SIZE's calibration on K3's own access patterns (tools/fetch_calib.sh -> profiles/<tag>_fetch_calibration.json);
    # without it, the guide's x2 for wide coalesced streams (an upper estimate for 12 B / lane node reads)
    factor, calib_file = 2.0, None
    for cand in (os.path.join(outdir, "..", "fetch_calib", "fetch_calibration.json"),
                 os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "%s_fetch_calibration.json" % tag)):
        try:
            cal = json.load(open(cand))["patterns"]
            factor = round(1.0 / cal["calib_nodes12"]["fetch_over_known"], 3)      # K3's HBM reads are the node stream (the rank granules hit in cache)
            calib_file = "%s_fetch_calibration.json" % tag
            break
        except Exception:
            pass
    tj = {
        "workload": line.get("config", {}).get("workload"),
        "bytes_per_gpu": line.get("config", {}).get("bytes_per_gpu"),
        "fetch_factor": factor, "calibration_file": calib_file,
        "traffic_bytes": factor * fr + wr,
        "archive_sha256": line.get("archive_sha256"),
        "kernel": "K3 (" + " + ".join(k3) + ", all rounds of one compression)",
        "fetch_size_bytes_raw": fr, "write_size_bytes": wr,
        "traffic_bytes_raw": fr + wr, "traffic_bytes_corrected": 2.0 * fr + wr,
        "algorithmic_bytes": line.get("roofline", {}).get("algorithmic_bytes"),
        "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reads 1/2 of the bytes of a wide coalesced "
                      "stream -> x2 on the read side; WRITE_SIZE exact.  K3's reads are 12 B/lane coalesced node triples plus "
                      "16 B/lane rank gathers (uncalibrated width), so the corrected figure is an upper estimate and the raw "
                      "one a lower bound.",
        "collected": "tools/profile.sh: rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace, separate "
                     "passes, bench.py --steps 1 --warmup 0 --no-cpu --no-decode --no-workloads",
    }
    with open(os.path.join(outdir, "%s_k3_traffic%s.json" % (tag, sfx)), "w") as f:
        json.dump(tj, f, indent=1)
    # SURVEY 8d's other families, same passes: algorithmic bytes K1 6n, K2 17n, K4 22 x symbols
    ff, fc = family_sums(outdir, "fetch" + sfx, "FETCH_SIZE")
    fw, _ = family_sums(outdir, "write" + sfx, "WRITE_SIZE")
    nb = line.get("config", {}).get("bytes_per_gpu") or 0
    syms = line.get("counts", {}).get("symbols") or 0
    alg = {"K1": 6.0 * nb, "K2": 17.0 * nb, "K3": 384.0 * nb + 16.0 * syms, "K4": 22.0 * syms}
    fam = {"archive_sha256": line.get("archive_sha256"), "bytes_per_gpu": nb, "symbols": syms,
           "note": "per compression; reads corrected x2 (FETCH_SIZE counts half of a wide coalesced stream on gfx950); radix-sort kernels "
                   "belong to K1 before the first K2 kernel and to K4 (slot sort, tagged tail symbols) after it",
           "families": {}}
    for k in ("K1", "K2", "K3", "K4"):
        fam["families"][k] = {"dispatches": fc.get(k, 0), "fetch_size_bytes_raw": ff.get(k, 0.0), "write_size_bytes": fw.get(k, 0.0),
                              "traffic_bytes": 2.0 * ff.get(k, 0.0) + fw.get(k, 0.0), "algorithmic_bytes": alg[k],
                              "over_algorithmic": round((2.0 * ff.get(k, 0.0) + fw.get(k, 0.0)) / alg[k], 3) if alg[k] else None}
    with open(os.path.join(outdir, "%s_family_traffic%s.json" % (tag, sfx)), "w") as f:
        json.dump(fam, f, indent=1)
    print(json.dumps({k: tj[k] for k in ("traffic_bytes_raw", "traffic_bytes_corrected", "algorithmic_bytes")}))


if __name__ == "__main__":
    main()
