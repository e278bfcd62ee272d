#!/bin/bash
# A/B of library variants on synth-text (K3 ms): tools/ab.sh var1 var2 ...   ("base" = bce_amd/lib/libbcehip.so)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for v in "$@"; do
  if [ $v = base ]; then unset BCE_HIP_LIB; else export BCE_HIP_LIB=$ROOT/bce_amd/lib/var_$v.so; fi
  timeout -k 10 200 python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-decode --no-workloads --no-e2e --no-stream $AB_ARGS 2>/dev/null | grep '^{' | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('%-10s'%'$v', j['value'],'MB/s K3',r['k3_ms_per_step'],'ms frac',r['frac'], j['breakdown_s']['t_bwt'], j['archive_sha256'][:12])"
done
