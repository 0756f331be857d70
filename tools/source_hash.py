#!/usr/bin/env python3
"""SHA-256 over the sources the product's device code is built from (every .hip / .h under libeddsa_amd/csrc except the
probe library's, plus the Makefile's device flags): what a set of hardware counters belongs to.  tools/summarize_profile.py
stores it with the counters; bench.py prints counters only when it matches the tree it runs from."""
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_source_files():
    d = os.path.join(ROOT, "libeddsa_amd", "csrc")
    return sorted(f for f in os.listdir(d) if (f.endswith(".hip") or f.endswith(".h")) and f != "probe.hip")


def device_source_hash():
    h = hashlib.sha256()
    for f in device_source_files():
        h.update(f.encode() + b"\0")
        h.update(open(os.path.join(ROOT, "libeddsa_amd", "csrc", f), "rb").read())
        h.update(b"\0")
    flags = re.findall(r"^HIPFLAGS\s*:=.*$", open(os.path.join(ROOT, "Makefile")).read(), flags=re.M)
    h.update("\n".join(flags).encode())
    return h.hexdigest()


if __name__ == "__main__":
    print(device_source_hash())
