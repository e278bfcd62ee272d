#!/usr/bin/env python3
"""Per wide round of ONE compression: grid and duration of k3_count2 / k3_tiles<write> and the gap to the next dispatch,
from a rocprofv3 kernel trace (the last compression in the trace):   python tools/k3_round_profile.py TRACE.csv"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k3 = [r for r in rows if "k3_" in r["Kernel_Name"] or "kd_" in r["Kernel_Name"]]
# last compression: after the last k1 kernel
last_k1 = max(i for i, r in enumerate(rows) if "k1_" in r["Kernel_Name"])
seq = [r for r in rows[last_k1:] if "k3_" in r["Kernel_Name"]]
t0 = int(seq[0]["Start_Timestamp"])
prev_end = None
tot = {}
gaps = 0.0
for r in seq:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void bce::", "").replace("bce::", "")
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    if gap > 0 and gap < 200:
        gaps += gap
    print("%9.1f us  %-28s grid %7s  %7.1f us  gap before %6.1f us" % ((s - t0) / 1e3, name[:28], r.get("Grid_Size", "?"), (e - s) / 1e3, gap))
    tot[name] = tot.get(name, 0.0) + (e - s) / 1e3
    prev_end = e
print("sums (us):", {k: round(v) for k, v in tot.items()}, "gaps between K3 kernels (< 200 us each):", round(gaps))
