#!/bin/bash
# cold CLI timings on the GPU box: generate inputs, then bce -c / -d / -ds / -s as a user would run them
set -e
cd "$(dirname "$0")/.."
python3 - <<'PY'
import subprocess, time, hashlib, os
import bce_amd
bce_amd.synth_text(1, 100_000_000).tofile('/tmp/in100.txt')
bce_amd.synth_text(1, 8 << 20).tofile('/tmp/in8.txt')
exe = 'bce_amd/bin/bce'
def run(args, label):
    t0 = time.perf_counter()
    r = subprocess.run([exe] + args, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    last = [l for l in r.stdout.replace('\r', '\n').split('\n') if l.strip()][-1]
    print("%-28s wall %.3f s   | %s" % (label, dt, last))
for i in (1, 2):
    run(['-c', '/tmp/out100.bce', '/tmp/in100.txt'], 'bce -c 100 MB (run %d)' % i)
run(['-d', '/tmp/back100.txt', '/tmp/out100.bce'], 'bce -d 100 MB')
print('roundtrip', 'ok' if open('/tmp/back100.txt', 'rb').read() == open('/tmp/in100.txt', 'rb').read() else 'FAILED')
run(['-c', '/tmp/out8.bce', '/tmp/in8.txt'], 'bce -c 8 MiB')
run(['-d', '/tmp/back8.txt', '/tmp/out8.bce'], 'bce -d 8 MiB')
run(['-ds', '/tmp/back8s.txt', '/tmp/out8.bce'], 'bce -ds 8 MiB (host)')
run(['-s', '/tmp/c8.bcc', '/tmp/in8.txt'], 'bce -s 8 MiB')
print('archive sha256', hashlib.sha256(open('/tmp/out100.bce', 'rb').read()).hexdigest()[:16])
PY
