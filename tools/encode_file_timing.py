#!/usr/bin/env python3
"""Compress a file twice on the GPU (input resident in HBM) and print time, MB/s, K3 time and the archive hash;
with BCE_HIP_DFS_DEBUG=1 the depth-first tail prints its per-pass statistics.
    python tools/encode_file_timing.py FILE"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np   # noqa: E402
import torch         # noqa: E402
import bce_amd       # noqa: E402

d = np.fromfile(sys.argv[1], dtype=np.uint8)
t = torch.from_numpy(d).to("cuda:0")
torch.cuda.synchronize()
for _ in range(2):
    t0 = time.time()
    arch, st = bce_amd.compress_device(t.data_ptr(), len(d))
    dt = time.time() - t0
    print("%.3f s  %.1f MB/s k3 %.1f ms sha %s nodes_ok %s" % (dt, len(d) / dt / 1e6, st["k3_ms"], hashlib.sha256(arch).hexdigest()[:8],
                                                                st["nodes"] == 8 * len(d) - 8))
