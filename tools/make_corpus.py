#!/usr/bin/env python3
"""Build a natural-text corpus from files that ship with this image (no network, nothing stored in the repo):
the .py sources under the interpreter's library and site-packages directories, then the ROCm C/C++ headers, in
sorted path order, concatenated and truncated to --size bytes.  Used as an enwik8-like input for
`bench.py --file` and tools/bigrep_check.py; the bench line records its size and sha256.

    python tools/make_corpus.py --out /tmp/corpus100.bin --size 100000000
"""
import argparse
import hashlib
import os
import sys


def walk(root, exts):
    for d, dirs, files in os.walk(root):
        dirs.sort()
        for f in sorted(files):
            if f.endswith(exts):
                yield os.path.join(d, f)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--size", type=int, default=100_000_000)
    a = ap.parse_args()
    roots = [(os.path.dirname(os.__file__), (".py",)),
             ("/usr/lib/python3/dist-packages", (".py",)),
             ("/usr/local/lib/python3.10/dist-packages", (".py",)),
             ("/opt/rocm/include", (".h", ".hpp"))]
    left = a.size
    h = hashlib.sha256()
    nfiles = 0
    with open(a.out, "wb") as out:
        for root, exts in roots:
            if left <= 0 or not os.path.isdir(root):
                continue
            for path in walk(root, exts):
                try:
                    with open(path, "rb") as f:
                        b = f.read(left)
                except OSError:
                    continue
                out.write(b)
                h.update(b)
                left -= len(b)
                nfiles += 1
                if left <= 0:
                    break
    print("corpus: %d B from %d files, sha256 %s" % (a.size - left, nfiles, h.hexdigest()), file=sys.stderr)
    return 0 if left <= 0 else 1


if __name__ == "__main__":
    sys.exit(main())
