#!/usr/bin/env python3
"""Live-node count per round of BCE::code on a file (stepping interface): at which round does the node count
fall below each power of two?   python tools/decay_profile.py FILE [--size N]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bce_amd   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("file")
    ap.add_argument("--size", type=int, default=0)
    ap.add_argument("--max-rounds", type=int, default=20000)
    a = ap.parse_args()
    data = np.fromfile(a.file, dtype=np.uint8, count=a.size or -1)
    rf = bce_amd.RankFile(data)
    enc = bce_amd.BCE(symbol_capacity=max(1 << 20, 2 * len(data)))   # the stepping interface never flushes
    enc.code_begin(rf)
    peak, r, below, hist = 0, 0, {}, {}
    done = 0
    while r < a.max_rounds:
        live = enc.code_round(rf)
        r += 1
        done += live
        peak = max(peak, live)
        b = max(0, int(live).bit_length() - 1)
        hist.setdefault(b, [0, 0])
        hist[b][0] += 1
        hist[b][1] += live
        if live < peak:
            for sh in range(28, 5, -1):
                if live <= (1 << sh) and sh not in below:
                    below[sh] = (r, done)
        if live == 0:
            break
    print("n %d peak %d rounds stepped %d" % (len(data), peak, r))
    for sh in sorted(below, reverse=True):
        print("  live <= 2^%-2d from round %6d  (%.3f of all nodes started by then)" % (sh, below[sh][0], below[sh][1] / (8.0 * len(data))))
    print("rounds by live-node count:")
    for b in sorted(hist):
        print("  [2^%-2d, 2^%-2d): %5d rounds, %12d nodes" % (b, b + 1, hist[b][0], hist[b][1]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
