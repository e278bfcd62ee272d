#!/bin/bash
# A/B of library variants, end to end: ms per step (mean of 5 after 1 warm-up) on text, natural and binary
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
[ -f /tmp/bce_natural_100000000.bin ] || python3 $ROOT/tools/make_corpus.py --out /tmp/bce_natural_100000000.bin --size 100000000 2>/dev/null
[ -f /tmp/bce_binary_100000000.bin ] || python3 $ROOT/tools/make_binary_corpus.py --out /tmp/bce_binary_100000000.bin --size 100000000 2>/dev/null
for v in "$@"; do
  if [ $v = base ]; then unset BCE_HIP_LIB; else export BCE_HIP_LIB=$ROOT/bce_amd/lib/var_$v.so; fi
  python3 - "$v" <<'P'
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bce_amd
out = []
for name, data in (('text', bce_amd.synth_text(1, 100_000_000)), ('nat', np.fromfile('/tmp/bce_natural_100000000.bin', dtype=np.uint8)), ('bin', np.fromfile('/tmp/bce_binary_100000000.bin', dtype=np.uint8))):
    t = torch.from_numpy(data).to('cuda:0'); torch.cuda.synchronize()
    ctx = bce_amd.api._Ctx(0)
    ts, ks = [], []
    for i in range(6):
        t0 = time.perf_counter()
        arch, st = bce_amd.compress_device(t.data_ptr(), len(data), ctx=ctx)
        ts.append(time.perf_counter() - t0); ks.append(st['k3_ms'])
    ctx.close()
    out.append('%s %.1f ms (K3 %.1f, K1 %.1f)' % (name, sum(ts[1:]) / 5 * 1e3, sum(ks[1:]) / 5, st['t_bwt'] * 1e3))
print('%-8s' % sys.argv[1], ' | '.join(out))
P
done
