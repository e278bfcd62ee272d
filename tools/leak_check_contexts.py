#!/usr/bin/env python3
"""Resident memory and number of mappings of the process over 40 contexts created and destroyed, each compressing and decoding four
inputs with the registered-mapping route forced for small buffers (BCE_HIP_REG_MIN) and the decoder's host tail forced: nothing may
grow from one context to the next (end of round 5: 1381 MB and 479 mappings throughout).   python tools/leak_check_contexts.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bce_amd
def rss_mb():
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS"): return int(line.split()[1]) / 1024.0
def maps():
    return sum(1 for _ in open("/proc/self/maps"))
# contexts created and destroyed, slots growing through the registered route, decodes with the host tail
os.environ["BCE_HIP_REG_MIN"] = "65536"
os.environ["BCE_DEC_FORCE_HOST_TAIL"] = "1"
datas = [bce_amd.synth_text(i, n) for i, n in enumerate((300000, 3000000, 900000, 6000000))]
for it in range(40):
    ctx = bce_amd.api._Ctx(0)
    for d in datas:
        arch = bce_amd.BCE().encode(bce_amd.RankFile(d, ctx=ctx))
        out = np.empty(len(d), dtype=np.uint8)
        assert bce_amd.decompress_device(arch, ctx=ctx, out=out) == len(d)
    st = bce_amd.stats_of(ctx)
    ctx.close()
    if it % 8 == 0 or it == 39:
        print("iteration %2d: rss %.0f MB, %d mappings, reg_maps %d unmaps %d" % (it, rss_mb(), maps(), st["reg_maps"], st["reg_unmaps"]), flush=True)
