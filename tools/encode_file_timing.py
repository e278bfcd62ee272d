import os, sys, time, hashlib
sys.path.insert(0, '/root/repo')
import numpy as np, torch, bce_amd
d = np.fromfile(sys.argv[1], dtype=np.uint8)
t = torch.from_numpy(d).to('cuda:0'); torch.cuda.synchronize()
for i in range(2):
    t0 = time.time(); arch, st = bce_amd.compress_device(t.data_ptr(), len(d)); dt = time.time() - t0
    print("%.3f s  %.1f MB/s k3 %.1f ms sha %s nodes_ok %s" % (dt, len(d) / dt / 1e6, st["k3_ms"], hashlib.sha256(arch).hexdigest()[:8], st["nodes"] == 8 * len(d) - 8))
