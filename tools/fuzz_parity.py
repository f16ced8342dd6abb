#!/usr/bin/env python3
"""Randomised parity stress: random shapes/configs through the C ABI vs the oracle.
usage: python tools/fuzz_parity.py [cases] [seed]     (needs an MI355X; prints one line per failure)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "em-spec_amd"), os.path.join(ROOT, "oracle")]
import numpy as np
import emspec, oracle as O
from emspec import synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
fails = 0
t0 = time.time()
for ci in range(cases):
    n = int(rng.choice([256, 512, 1024, 2048, 4096, 4096, 4096, 8192, 16384]))
    hop = int(rng.choice([n // 16, n // 16, n // 8, n // 4, n // 2, n, max(1, n // 32), int(rng.integers(1, n + 1))]))
    rows = int(rng.choice([64, 128, 256, 512, 1024, 1024, 2048])) if n >= 1024 else int(rng.choice([64, 128, 256]))
    reassign = bool(rng.integers(0, 2))
    S = int(rng.integers(1, 4))
    D = -(-n // (2 * hop)) if reassign else 0
    maxframes = 40 if D <= 64 else 12
    frames = int(rng.integers(1, maxframes + 1))
    if n * frames * S > 6e6:
        frames = max(1, int(6e6 / (n * S)))
    extra = int(rng.integers(0, max(1, min(hop, 50))))
    L = n + hop * (frames - 1) + extra
    fmin = float(rng.choice([20.0, 35.0, 80.0]))
    fmax = float(rng.choice([24000.0, 18000.0, 8000.0]))
    kw = dict(rows=rows, fmin_hz=fmin, fmax_hz=fmax, gain=float(rng.choice([1.0, 3.5, 0.25])),
              db_range=float(rng.choice([80.0, 58.0])), gate_db=float(rng.choice([-80.0, -65.0])),
              power_floor=float(rng.choice([1e-14, 1e-10, 0.0])))
    kind = rng.integers(0, 4)
    pcm = synth.streams(S, L, first=int(rng.integers(0, 1000)))
    if kind == 1:
        pcm *= np.float32(10.0 ** rng.uniform(-3, 1))
    elif kind == 2:
        pcm[:, :: max(1, int(rng.integers(50, 5000)))] += np.float32(rng.uniform(0.1, 2.0))
    elif kind == 3:
        pcm = (pcm * 0).astype(np.float32) if rng.integers(0, 2) else np.sign(pcm).astype(np.float32) * np.float32(0.7)
    desc = f"case {ci}: n={n} hop={hop} rows={rows} re={reassign} S={S} frames={frames} L={L} kind={kind} {kw}"
    try:
        with emspec.Engine(**kw) as e:
            out = e.batch(pcm, n, hop, reassign, want=("db", "index"))
            nf = min(frames, 4)
            pw, col, row = e.parity_dump(pcm, n, hop, reassign, frames - nf, nf)
            stream_err = 0.0
            if ci % 4 == 0 and frames >= 2:      # the streaming call must reproduce the batched columns
                e.reset()
                got = {}
                for j in range(frames):
                    dbc, c = e.column(pcm[0, j * hop:j * hop + n], hop, reassign)
                    if c >= 0:
                        got[c] = dbc
                while True:
                    try:
                        dbc, c = e.flush()
                    except emspec.EmspecError:
                        break
                    got[c] = dbc
                assert sorted(got) == list(range(frames)), sorted(got)
                stream_err = float(np.max(np.abs(np.stack([got[c] for c in range(frames)]) - out["db"][0])))
        cfg = O.make_cfg(n, hop, reassign, **kw)
        odb, _, oidx = O.batch_f32(cfg, pcm, want=("db", "index"))
        err = float(np.max(np.abs(out["db"] - odb)))
        didx = int(np.abs(out["index"].astype(int) - oidx.astype(int)).max())
        bad = err >= 8.7e-4 or didx > 1 or stream_err >= 8.7e-4
        for s in range(S):
            opw, ocol, orow = O.frames_f32(cfg, pcm[s], frames - nf, nf)
            if not (np.array_equal(col[s], ocol) and np.array_equal(row[s], orow) and np.array_equal(pw[s], opw)):
                bad = True
                desc += f" [dump mismatch stream {s}: col {np.sum(col[s] != ocol)} row {np.sum(row[s] != orow)} pw {np.sum(pw[s] != opw)}]"
        if bad:
            fails += 1
            print("FAIL", desc, f"max|dB|={err:.3e} didx={didx} stream={stream_err:.3e}", flush=True)
    except Exception as ex:
        fails += 1
        print("EXC ", desc, repr(ex), flush=True)
    if ci % 25 == 24:
        print(f"... {ci + 1} cases, {fails} failures, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz done: {cases} cases, {fails} failures, seed {seed}")
sys.exit(1 if fails else 0)
