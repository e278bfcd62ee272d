#!/bin/bash
# Cold `bce -c / -d / -s` as a user runs them: wall seconds of a fresh process per input size (5 runs: min / median),
# the stage laps of BCE_CLI_TIMING=1 and the floor of any one-shot HIP program on this box (tools/hip_floor.hip).
cd "$(dirname "$0")/.."
mkdir -p tools/_build
[ -x tools/_build/hip_floor ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/hip_floor.hip -o tools/_build/hip_floor -lpthread
python3 - <<'PY'
import subprocess, time, os, statistics
import bce_amd
exe = 'bce_amd/bin/bce'
sizes = (1000, 1 << 20, 100_000_000)
for n in sizes:
    bce_amd.synth_text(1, n).tofile('/tmp/in_%d.txt' % n)
def wall(cmd, env=None):
    time.sleep(float(os.environ.get("CLI_COLD_GAP", "0.4")))     # let the driver finish tearing the previous process down
    t0 = time.perf_counter(); subprocess.run(cmd, capture_output=True, env=env); return time.perf_counter() - t0
for n in sizes:
    f = '/tmp/in_%d.txt' % n
    line = "%10d B:" % n
    for mode, args in (('-c', ['-c', '/tmp/out.bce', f]), ('-d', ['-d', '/tmp/back.txt', '/tmp/out.bce']), ('-s', ['-s', '/tmp/c.bcc', f])):
        ts = sorted(wall([exe] + args) for _ in range(5))
        line += "  %s %.3f / %.3f s" % (mode, ts[0], statistics.median(ts))
    print(line + "   (min / median of 5)", flush=True)
    ok = open('/tmp/back.txt', 'rb').read() == open(f, 'rb').read()
    print("           roundtrip", "ok" if ok else "FAILED", flush=True)
env = dict(os.environ, BCE_CLI_CLEAN_EXIT="1")
ts = sorted(wall([exe, '-c', '/tmp/out.bce', '/tmp/in_100000000.txt'], env) for _ in range(3))
print("100000000 B with BCE_CLI_CLEAN_EXIT=1 (destroy + runtime exit handlers): -c %.3f / %.3f s" % (ts[0], statistics.median(ts)))
if os.path.exists('tools/_build/hip_floor'):
    for m in ('base', 'base', 'base', 'big', 'many', 'pin', 'vmm'):
        t0 = time.perf_counter(); r = subprocess.run(['tools/_build/hip_floor', m], capture_output=True, text=True); dt = time.perf_counter() - t0
        print(r.stdout.rstrip()); print("  process wall %.3f s" % dt, flush=True)
ts = sorted(wall([exe]) for _ in range(5))
print("bce (usage only, no HIP call) wall %.3f / %.3f s" % (ts[0], statistics.median(ts)))
PY
for n in 1000 100000000; do
  for i in 1 2; do
  echo "== BCE_CLI_TIMING, $n B"
  BCE_CLI_TIMING=1 bce_amd/bin/bce -c /tmp/out.bce /tmp/in_$n.txt 2>&1 | tr '\r' '\n' | grep -a "cli:\|lib:"
  done
done
