#!/usr/bin/env python3
"""Build a natural-text corpus from files that ship with this image (no network, nothing stored in the repo):
the .py sources of the interpreter's library, of the distribution's dist-packages and of a FIXED list of large
packages of the base image (the full site-packages directory differs between the build container and the GPU boxes,
so walking all of it is not reproducible -- round 1's corpus was), then the ROCm C/C++ headers, in sorted path
order, concatenated and truncated to --size bytes.  Used as an enwik8-like input ("natural v2") by bench.py,
tools/make_oracle_golden.py and the tests; its sha256 is recorded wherever it is used.

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


SITE = "/usr/local/lib/python3.10/dist-packages"
PACKAGES = ("numpy", "pandas", "scipy", "sympy", "torch", "matplotlib", "networkx")   # base image, same on every box


def corpus_roots():
    return ([(os.path.dirname(os.__file__), (".py",)), ("/usr/lib/python3/dist-packages", (".py",))] +
            [(os.path.join(SITE, p), (".py",)) for p in PACKAGES] + [("/opt/rocm/include", (".h", ".hpp"))])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--size", type=int, default=100_000_000)
    a = ap.parse_args()
    roots = corpus_roots()
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
