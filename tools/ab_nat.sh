#!/bin/bash
# A/B of library variants on the natural corpus: K3 ms (mean of 5 after 1 warm-up) and the local passes' times
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
F=${AB_FILE:-/tmp/bce_natural_100000000.bin}
[ -f /tmp/bce_natural_100000000.bin ] || python3 $ROOT/tools/make_corpus.py --out /tmp/bce_natural_100000000.bin --size 100000000 2>/dev/null
[ -f /tmp/bce_binary_100000000.bin ] || python3 $ROOT/tools/make_binary_corpus.py --out /tmp/bce_binary_100000000.bin --size 100000000 2>/dev/null
for v in "$@"; do
  if [ $v = base ]; then unset BCE_HIP_LIB; else export BCE_HIP_LIB=$ROOT/bce_amd/lib/var_$v.so; fi
  python3 - "$v" "$F" <<'P'
import sys, numpy as np, torch
sys.path.insert(0, '.')
import bce_amd
data = np.fromfile(sys.argv[2], dtype=np.uint8)
t = torch.from_numpy(data).to('cuda:0'); torch.cuda.synchronize()
ctx = bce_amd.api._Ctx(0)
ks = []
for i in range(6):
    arch, st = bce_amd.compress_device(t.data_ptr(), len(data), ctx=ctx)
    ks.append(st['k3_ms'])
print('%-10s K3 ms %s mean %.2f  t_total %.1f ms' % (sys.argv[1], ' '.join('%.1f' % k for k in ks[1:]), sum(ks[1:]) / 5, st['t_total'] * 1e3))
P
done
