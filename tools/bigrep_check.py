"""A whole-file duplicate (4 MiB twice: 33 M rounds): encode vs the oracle, host-decoder round trip.   python tools/bigrep_check.py"""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bce_amd, oracle
half = bce_amd.synth_text(5, 4 << 20)
data = np.concatenate([half, half, np.frombuffer(b'#', dtype=np.uint8)])
rf = bce_amd.RankFile(data)
t = time.time(); arch = bce_amd.BCE().encode(rf); dt = time.time() - t
st = bce_amd.stats(rf); rf.close()
print('n', len(data), 'archive', len(arch), 'encode %.3fs' % dt, {k: st[k] for k in ('rounds', 'nodes', 'sort_rounds', 't_bwt', 't_enum', 't_model', 'k3_ms')})
assert st['nodes'] == 8 * len(data) - 8
t = time.time(); ok = bce_amd.decompress(arch) == data.tobytes(); print('roundtrip', ok, '%.1fs' % (time.time() - t))
t = time.time(); ref = oracle.compress(data.tobytes()); print('oracle equal', ref == arch, '%.1fs' % (time.time() - t))
