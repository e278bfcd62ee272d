#!/usr/bin/env python3
"""Decode an archive file with the GPU-assisted decoder (no encode in the process: for rocprofv3):  python tools/decode_archive_timing.py ARCHIVE [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bce_amd
arch = open(sys.argv[1], "rb").read()
ctx = bce_amd.api._Ctx(0)
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
    t0 = time.time(); back = bce_amd.decompress_device(arch, ctx=ctx); dt = time.time() - t0
    print("decode %.3f s  %.1f MB/s (%d B)" % (dt, len(back) / dt / 1e6, len(back)), flush=True)
