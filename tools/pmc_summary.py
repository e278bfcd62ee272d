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


def main():
    outdir, tag = sys.argv[1], sys.argv[2]
    sfx = sys.argv[3] if len(sys.argv) > 3 else ""          # e.g. "_1e9": passes in fetch_1e9/ write_1e9/, log fetch_1e9.log
    fetch, calls = sums(outdir, "fetch" + sfx, "FETCH_SIZE")
    write, _ = sums(outdir, "write" + sfx, "WRITE_SIZE")
    rows = sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, 0) + write.get(k, 0)))
    with open(os.path.join(outdir, "%s_pmc_fetch_write%s.csv" % (tag, sfx)), "w") as f:
        f.write("kernel,dispatches,FETCH_SIZE_KB_sum,WRITE_SIZE_KB_sum\n")
        for k in rows:
            f.write("%s,%d,%.1f,%.1f\n" % (k, calls.get(k, 0), fetch.get(k, 0.0), write.get(k, 0.0)))
    k3 = [k for k in rows if "k3_" in k]
    fr = sum(fetch.get(k, 0.0) for k in k3) * 1024.0
    wr = sum(write.get(k, 0.0) for k in k3) * 1024.0
    line = {}
    try:
        with open(os.path.join(outdir, "fetch%s.log" % sfx)) as f:
            line = json.loads([l for l in f if l.startswith("{")][-1])
    except Exception:
        pass
    tj = {
        "workload": line.get("config", {}).get("workload"),
        "bytes_per_gpu": line.get("config", {}).get("bytes_per_gpu"),
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
    print(json.dumps({k: tj[k] for k in ("traffic_bytes_raw", "traffic_bytes_corrected", "algorithmic_bytes")}))


if __name__ == "__main__":
    main()
