#!/usr/bin/env python3
"""A binary corpus from the image's own shared libraries (sorted paths under /usr/lib/x86_64-linux-gnu and
/opt/rocm/lib), truncated to --size bytes: executables, tables, zero runs -- all 8 planes busy.
    python tools/make_binary_corpus.py --out /tmp/bin100.bin --size 100000000"""
import argparse
import hashlib
import os
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--size", type=int, default=100_000_000)
    a = ap.parse_args()
    left, h, nf = a.size, hashlib.sha256(), 0
    with open(a.out, "wb") as out:
        for root in ("/usr/lib/x86_64-linux-gnu", "/opt/rocm/lib"):
            if left <= 0 or not os.path.isdir(root):
                continue
            for name in sorted(os.listdir(root)):
                path = os.path.join(root, name)
                if left <= 0:
                    break
                if os.path.islink(path) or not os.path.isfile(path) or ".so" not in name:
                    continue
                try:
                    with open(path, "rb") as f:
                        b = f.read(min(left, 8 << 20))          # at most 8 MiB of each library
                except OSError:
                    continue
                out.write(b); h.update(b); left -= len(b); nf += 1
    print("binary corpus: %d B from %d files, sha256 %s" % (a.size - left, nf, h.hexdigest()), file=sys.stderr)
    return 0 if left <= 0 else 1


if __name__ == "__main__":
    sys.exit(main())
