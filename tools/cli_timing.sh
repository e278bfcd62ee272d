#!/bin/bash
# cold CLI timings on the GPU box: generate a 100 MB synth-text file, then bce -c / -d / -s
set -e
cd "$(dirname "$0")/.."
python3 - <<'PY'
import bce_amd
bce_amd.synth_text(1, 100_000_000).tofile('/tmp/in100.txt')
bce_amd.synth_text(1, 8 << 20).tofile('/tmp/in8.txt')
PY
for i in 1 2; do
  s=$(date +%s.%N); bce_amd/bin/bce -c /tmp/out100.bce /tmp/in100.txt | tail -1; e=$(date +%s.%N); echo "bce -c 100MB wall: $(python3 -c "print(round($e - $s, 3))") s"
done
s=$(date +%s.%N); bce_amd/bin/bce -c /tmp/out8.bce /tmp/in8.txt | tail -1; e=$(date +%s.%N); echo "bce -c 8MiB wall: $(python3 -c "print(round($e - $s, 3))") s"
s=$(date +%s.%N); bce_amd/bin/bce -d /tmp/back8.txt /tmp/out8.bce | tail -1; e=$(date +%s.%N); echo "bce -d 8MiB wall: $(python3 -c "print(round($e - $s, 3))") s"
cmp /tmp/in8.txt /tmp/back8.txt && echo roundtrip-ok
s=$(date +%s.%N); bce_amd/bin/bce -s /tmp/c8.bcc /tmp/in8.txt | tail -1; e=$(date +%s.%N); echo "bce -s 8MiB wall: $(python3 -c "print(round($e - $s, 3))") s"
sha256sum /tmp/out100.bce | cut -c1-16
