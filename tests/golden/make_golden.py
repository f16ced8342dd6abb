#!/usr/bin/env python3
"""Generate the committed golden fixtures in tests/golden/.

The reference implementation is unavailable (private source,
/root/reference/README.md:73; no tests or vectors exist), so these
vectors do NOT come from the reference.  They come from the independent numpy
formulation oracle/ref_numpy.py (np.fft.rfft on three explicitly windowed frames)
applied to seeded synthetic audio, and pin the C oracle (and through it the HIP
path) to a second, unrelated implementation of the same published method.

Run:  python tests/golden/make_golden.py      (rewrites tests/golden/*.npz)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "em-spec_amd")]

import ref_numpy as RN  # noqa: E402
from emspec import synth  # noqa: E402

CASES = [  # name, n, hop, frames, stream seed, reassign
    ("n1024_h256_off", 1024, 256, 6, 11, False),
    ("n1024_h256", 1024, 256, 6, 11, True),
    ("n4096_h256", 4096, 256, 4, 12, True),
    ("n16384_h512", 16384, 512, 2, 13, True),
]


def main():
    for name, n, hop, frames, seed, reassign in CASES:
        frame0 = 3
        pcm = synth.stream(seed, n + hop * (frame0 + frames - 1))
        r = RN.reassign_frames(pcm, n, hop, frame0, frames, reassign=reassign)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), pcm=pcm, n=n, hop=hop, frame0=frame0, frames=frames,
                            reassign=reassign, power=r["power"], that=r["that"], khat=r["khat"],
                            col=r["col"], row=r["row"])
        print(name, "ok", os.path.getsize(os.path.join(HERE, name + ".npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
