#!/usr/bin/env python3
"""Cold `bce -d` (a fresh process per run) on the three 10^8-byte workloads: wall seconds, min / median of 5, with the decoder's
   boundary-rank buffer on huge pages (default) and on hipHostMalloc's pages (BCE_DEC_NO_HUGE=1); the output is compared.
   python tools/cli_decode_cold.py [text|natural|binary ...]"""
import os
import statistics
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scan_time import load  # noqa: E402

exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bce_amd", "bin", "bce")


def wall(cmd, env):
    time.sleep(0.4)
    t0 = time.perf_counter()
    subprocess.run(cmd, capture_output=True, env=env, check=True)
    return time.perf_counter() - t0


for kind in sys.argv[1:] or ["text", "natural", "binary"]:
    d = load(kind)
    src = "/tmp/cold_%s.in" % kind
    d.tofile(src)
    subprocess.run([exe, "-c", "/tmp/cold.bce", src], capture_output=True, check=True)
    line = "%-8s bce -d" % kind
    for label, extra in (("huge pages", {}), ("hipHostMalloc", {"BCE_DEC_NO_HUGE": "1"})):
        env = dict(os.environ, **extra)
        ts = sorted(wall([exe, "-d", "/tmp/cold.out", "/tmp/cold.bce"], env) for _ in range(5))
        ok = open("/tmp/cold.out", "rb").read() == open(src, "rb").read()
        line += "   %s %.3f / %.3f s%s" % (label, ts[0], statistics.median(ts), "" if ok else " WRONG OUTPUT")
    print(line + "   (min / median of 5)", flush=True)
