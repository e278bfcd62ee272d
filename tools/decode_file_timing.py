#!/usr/bin/env python3
"""Encode a file on the GPU, decode the archive with the GPU-assisted decoder, compare, print both times.
    python tools/decode_file_timing.py FILE"""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bce_amd
d = np.fromfile(sys.argv[1], dtype=np.uint8)
t = torch.from_numpy(d).to('cuda:0'); torch.cuda.synchronize()
arch, st = bce_amd.compress_device(t.data_ptr(), len(d))
del t
for i in range(2):
    t0 = time.time(); back = bce_amd.decompress_device(arch); dt = time.time() - t0
    print("decode %.3f s  %.1f MB/s  identical %s  (archive sha %s)" % (dt, len(d) / dt / 1e6, back == d.tobytes(), hashlib.sha256(arch).hexdigest()[:8]), flush=True)
