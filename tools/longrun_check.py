"""Megabyte runs and tables through the encoder (vs the oracle) and the GPU-assisted decoder, with times: the inputs the
staircase closed forms (k3_dfs.hip) and the decoder's host tail (kd_decode.hip) exist for.
    python tools/longrun_check.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bce_amd, oracle
text = oracle.synth_text(5, 3000000)
cases = {
  "zeros2M": text[:1000000] + bytes(2000000) + text[1000000:],
  "zeros2M+1.5M": text[:1000000] + bytes(2000000) + text[1000000:2000000] + bytes(1500000) + b"\x01" + text[2000000:],
  "table16x100K": text[:1000000] + (b"RECORD\x00\x00\x01\x02\x03\x04\x05\x06\x07\x08" * 100000) + text[1000000:],
  "ab-all": b"ab" * 500000 + b"c",
}
for name, data in cases.items():
    t0 = time.time(); want = oracle.compress(data); to = time.time() - t0
    rf = bce_amd.RankFile(data)
    t0 = time.time(); arch = bce_amd.BCE().encode(rf); tg = time.time() - t0
    st = bce_amd.stats(rf); rf.close()
    t0 = time.time(); back = bce_amd.decompress_device(arch); td = time.time() - t0
    print("%-14s n=%d oracle %.1fs gpu %.3fs (k3 %.1f ms, rounds %d) decode %.2fs  parity %s roundtrip %s" % (name, len(data), to, tg, st["k3_ms"], st["rounds"], td, arch == want, back == data), flush=True)
