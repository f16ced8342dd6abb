#!/usr/bin/env python3
"""Long-run parity stress of the fused (LDS ring) kernels: many segments per stream, random segment
lengths (EMSPEC_SEGLEN), random rows / gain / floor, hundreds to thousands of frames, every fused shape.
usage: python tools/fuzz_long.py [cases] [seed]     (needs an MI355X; prints one line per failure)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "em-spec_amd"), os.path.join(ROOT, "oracle")]
import numpy as np
import emspec, oracle as O
from emspec import synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
SHAPES = [(4096, 256), (4096, 512), (4096, 1024), (4096, 300), (4096, 2048), (8192, 512), (8192, 1024),
          (2048, 128), (2048, 256), (2048, 200), (2048, 1000), (1024, 128), (1024, 256), (1024, 100), (1024, 700),
          (16384, 512), (16384, 512), (16384, 1024), (16384, 700), (16384, 4096)]
fails, t0 = 0, time.time()
for ci in range(cases):
    n, hop = SHAPES[int(rng.integers(0, len(SHAPES)))]
    reassign = bool(rng.integers(0, 4))            # mostly on
    rows = int(rng.choice([64, 256, 512, 1024, 1024]))
    S = int(rng.integers(1, 4))
    frames = int(rng.integers(200, 2500))
    if n == 16384:
        frames = int(rng.integers(60, 500))            # the CPU oracle takes ~1 ms per frame at this size
    if n * 1 + hop * frames * S > 4e6:
        frames = max(50, int((4e6 / S - n) / hop))
    L = n + hop * (frames - 1) + int(rng.integers(0, hop))
    seglen = int(rng.choice([0, 0, 64, 65, 100, 128, 257, 400, 1000]))
    if seglen:
        os.environ["EMSPEC_SEGLEN"] = str(seglen)
    else:
        os.environ.pop("EMSPEC_SEGLEN", None)
    kw = dict(rows=rows, gain=float(rng.choice([1.0, 3.5])), power_floor=float(rng.choice([1e-14, 1e-10, 0.0])))
    pcm = synth.streams(S, L, first=int(rng.integers(0, 1000)))
    if rng.integers(0, 3) == 0:
        pcm[:, :: int(rng.integers(300, 5000))] += np.float32(rng.uniform(0.2, 1.5))     # clicks: long-range time reassignment
    desc = f"case {ci}: n={n} hop={hop} rows={rows} re={reassign} S={S} frames={frames} seglen={seglen} {kw}"
    try:
        with emspec.Engine(diag=bool(seglen) or "EMSPEC_SHARED" in os.environ, **kw) as e:    # EMSPEC_SEGLEN / EMSPEC_SHARED are switches of the diagnostic build
            fused = e.fused(n, hop, reassign)
            out = e.batch(pcm, n, hop, reassign, want=("db", "index"))
        cfg = O.make_cfg(n, hop, reassign, **kw)
        odb, _, oidx = O.batch_f32(cfg, pcm, want=("db", "index"))
        err = float(np.max(np.abs(out["db"] - odb)))
        didx = np.abs(out["index"].astype(int) - oidx.astype(int))
        if err >= 8.7e-4 or didx.max() > 1 or np.mean(didx != 0) > 1e-3 or not fused:
            fails += 1
            print("FAIL", desc, f"fused={fused} max|dB|={err:.3e} didx={didx.max()}", flush=True)
    except Exception as ex:
        fails += 1
        print("EXC ", desc, repr(ex), flush=True)
    if ci % 10 == 9:
        print(f"... {ci + 1} cases, {fails} failures, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz_long done: {cases} cases, {fails} failures, seed {seed}")
sys.exit(1 if fails else 0)
