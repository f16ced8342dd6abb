#!/usr/bin/env python3
"""Per-kernel registers / scratch / LDS from em-spec_amd/csrc/kernels.s (`make -C em-spec_amd/csrc asm`).
usage: kernel_resources.py [substring ...]   (demangled-name filters; none = every kernel)"""
import os
import re
import subprocess
import sys

S = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "em-spec_amd", "csrc", "kernels.s")


def kernels(path=S):
    out, name = [], None
    cur = {}
    for line in open(path, errors="replace"):
        m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", line)
        if m:
            name, cur = m.group(1), {}
            continue
        if name:
            m = re.match(r"\s*\.amdhsa_(next_free_vgpr|private_segment_fixed_size|group_segment_fixed_size|accum_offset)\s+(\d+)", line)
            if m:
                cur[m.group(1)] = int(m.group(2))
            if ".end_amdhsa_kernel" in line:
                out.append((name, cur))
                name = None
    return out


if __name__ == "__main__":
    ks = kernels()
    names = subprocess.run(["c++filt"], input="\n".join(k for k, _ in ks), capture_output=True, text=True).stdout.split("\n")
    for (k, r), d in zip(ks, names):
        d = re.sub(r"\(.*", "", d).replace("void emspec::", "")
        if len(sys.argv) > 1 and not any(f in d for f in sys.argv[1:]):
            continue
        print(f"{d:70s} vgpr {r.get('next_free_vgpr', -1):4d} scratch {r.get('private_segment_fixed_size', -1):5d} B  static-lds {r.get('group_segment_fixed_size', -1)}")
