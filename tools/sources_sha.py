#!/usr/bin/env python3
"""sha1 (16 hex digits) over the kernel / C-ABI sources under em-spec_amd/csrc.  ONE definition for the three places that must
agree: the library (the Makefile compiles it in: emspec_build_info), bench.py (profile-derived numbers are only quoted
while library, tree and profile carry the same digits) and tools/profile_workload.sh (what a profile was taken on)."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sources_sha(root=ROOT):
    h = hashlib.sha1()
    base = os.path.join(root, "em-spec_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(base, "**", "*"), recursive=True)):
        if os.path.isfile(f) and f.endswith((".hip", ".inc", ".h", ".cpp")):
            h.update(os.path.relpath(f, base).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(sources_sha())
