#!/bin/bash
# quick look on the GPU box: one-at-a-time and two-context stream figures for synth-text and the natural / binary corpora
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
[ -f /tmp/nat.bin ] || python3 $ROOT/tools/make_corpus.py --out /tmp/nat.bin --size 100000000 2>/dev/null
[ -f /tmp/bin.bin ] || python3 $ROOT/tools/make_binary_corpus.py --out /tmp/bin.bin --size 100000000 2>/dev/null
for w in text nat bin; do
  if [ $w = text ]; then F=""; else F="--file /tmp/$w.bin"; fi
  timeout -k 10 300 python3 $ROOT/bench.py --steps 6 --warmup 1 --no-cpu --no-decode --no-workloads --no-e2e $F "$@" 2>$OUT/s_$w.err | grep '^{' | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; s=j.get('value_stream')
print('$w', j['value'],'MB/s', j['ms_per_step'],'ms/step  K3',r['k3_ms_per_step'],'ms frac',r['frac'], j['breakdown_s'], j['oracle_golden'])
print('   stream:', s)"
done
