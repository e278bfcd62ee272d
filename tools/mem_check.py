#!/usr/bin/env python3
"""Device memory a context holds after one compression: python tools/mem_check.py N [N ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bce_amd
for n in [int(float(x)) for x in sys.argv[1:]]:
    d = bce_amd.synth_text(1, n)
    t = torch.from_numpy(d).to("cuda:0"); torch.cuda.synchronize()
    free0, total = torch.cuda.mem_get_info()
    ctx = bce_amd.api._Ctx(0)
    arch, st = bce_amd.compress_device(t.data_ptr(), n, ctx=ctx)
    free1, _ = torch.cuda.mem_get_info()
    back = bce_amd.decompress_device(arch, ctx=ctx) if n <= 10**8 else None
    free2, _ = torch.cuda.mem_get_info()
    print("n = %d: context holds %.1f GB after -c (%.1f bytes per input byte)%s; card %.0f GB" % (
        n, (free0 - free1) / 1e9, (free0 - free1) / n, ", %.1f GB after -d in the same context" % ((free0 - free2) / 1e9) if back else "", total / 1e9), flush=True)
    ctx.close(); del t
