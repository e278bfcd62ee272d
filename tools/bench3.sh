#!/bin/bash
# quick look on the GPU box: bench lines (+ tail pass timings) for the natural / binary corpora and synth-text
#   tools/bench3.sh TAG [extra bench args]
TAG=${1:-x}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
[ -f /tmp/nat.bin ] || python3 $ROOT/tools/make_corpus.py --out /tmp/nat.bin --size 100000000 2>/dev/null
[ -f /tmp/bin.bin ] || python3 $ROOT/tools/make_binary_corpus.py --out /tmp/bin.bin --size 100000000 2>/dev/null
for w in nat bin text; do
  if [ $w = text ]; then F=""; else F="--file /tmp/$w.bin"; fi
  BCE_HIP_DFS_DEBUG=1 timeout -k 10 200 python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-decode --no-workloads --no-e2e --no-stream $F "$@" > $OUT/b_${TAG}_$w.log 2>&1
  echo "== $w"; grep -E "^local pass|^dfs pass|^dfs:" $OUT/b_${TAG}_$w.log | tail -14 | cut -c1-230
  grep '^{' $OUT/b_${TAG}_$w.log | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print(j['value'],'MB/s', j['ms_per_step'],'ms/step  K3',r['k3_ms_per_step'],'ms frac',r['frac'],'launches',r['k3_launches_per_step'], j['breakdown_s'], j['archive_sha256'][:16])"
done
