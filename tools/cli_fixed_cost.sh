#!/bin/bash
# fixed cost of one CLI run (process start, HIP initialisation, the context's allocations) against input size
cd "$(dirname "$0")/.."
python3 - <<'PY'
import subprocess, time
import bce_amd
exe = 'bce_amd/bin/bce'
for n in (1000, 1 << 20, 8 << 20, 32 << 20, 100_000_000):
    bce_amd.synth_text(1, n).tofile('/tmp/in.txt')
    best = {}
    for mode, args in (('-c', ['-c', '/tmp/out.bce', '/tmp/in.txt']), ('-d', ['-d', '/tmp/back.txt', '/tmp/out.bce']), ('-s', ['-s', '/tmp/c.bcc', '/tmp/in.txt'])):
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); subprocess.run([exe] + args, capture_output=True); ts.append(time.perf_counter() - t0)
        best[mode] = min(ts)
    print("%10d B: -c %.3f s  -d %.3f s  -s %.3f s" % (n, best['-c'], best['-d'], best['-s']), flush=True)
PY
BCE_HIP_TIMING=1 bce_amd/bin/bce -c /tmp/out.bce /tmp/in.txt 2>&1 | tail -5
