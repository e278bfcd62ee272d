"""Re-derive every golden vector with the CPU oracle (run from the repo root)."""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import oracle  # noqa: E402


def gen_input(v):
    if v["gen"] == "literal":
        return v["literal"].encode()
    return getattr(oracle, v["gen"])(v["seed"], v["n"])


def main():
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden.json")))
    ok = True
    for v in g["vectors"]:
        d = gen_input(v)
        if "input_sha256" in v:
            assert hashlib.sha256(d).hexdigest() == v["input_sha256"], v["name"]
        a = oracle.compress(d)
        good = len(a) == v["archive_bytes"] and hashlib.sha256(a).hexdigest() == v["archive_sha256"]
        print("%-16s %9d -> %8d  %s" % (v["name"], len(d), len(a), "ok" if good else "MISMATCH"))
        ok &= good
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
