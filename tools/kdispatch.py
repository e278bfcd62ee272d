#!/usr/bin/env python3
"""Durations of every dispatch of the kernels whose name contains PATTERN, in launch order, from a rocprofv3 kernel_trace.csv:
    python tools/kdispatch.py TRACE.csv PATTERN"""
import csv
import sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(" ".join("%.0f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows), "(us)")
