#!/usr/bin/env python3
"""Per-compression K1 / radix kernel times from a rocprofv3 kernel_stats.csv:  python tools/kstat_k1.py CSV RUNS"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
runs = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = 0.0
for r in rows:
    nm = r["Name"]
    if "k1_" in nm or "rs_" in nm or "rw_" in nm:
        ms = float(r["TotalDurationNs"]) / 1e6 / runs
        tot += ms
        print("%8.3f ms  %6.1f calls  avg %9.1f us  %s" % (ms, float(r["Calls"]) / runs, float(r["AverageNs"]) / 1e3, nm[:90]))
print("%8.3f ms  K1 + sort kernels per compression" % tot)
