#!/usr/bin/env python3
"""Randomised parity stress of the EXACT mode: random shapes / configs / signals through the C ABI against the binary64
bit model (oracle/emspec_exact.c) - every comparison is array_equal (dump power / column / row / q, dB bits, palette
index), plus the streaming call against the batch bytes and (column,row) against the independent float64 method.
usage: python tools/fuzz_exact.py [cases] [seed]     (needs an MI355X; prints one line per failure)"""
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
f64_bins = f64_bad = 0
t0 = time.time()
for ci in range(cases):
    n = int(rng.choice([256, 512, 1024, 2048, 4096, 4096, 4096, 8192, 16384]))
    hop = int(rng.choice([n // 16, n // 16, n // 8, n // 4, n // 2, n, max(1, n // 32), int(rng.integers(1, n + 1))]))
    rows = int(rng.choice([64, 68, 100, 128, 256, 512, 1000, 1024, 1024, 2048, 4096]))
    reassign = bool(rng.integers(0, 2))
    S = int(rng.integers(1, 4))
    D = -(-n // (2 * hop)) if reassign else 0
    frames = int(rng.integers(1, (40 if D <= 64 else 12) + 1))
    if n in (1024, 2048, 4096) and rng.integers(0, 2):   # the one-kernel exact path (N = 4096: round 4; 2048 / 1024: round 6): longer walks, several segments (short forced segments below)
        frames = int(rng.integers(1, 260))
    if n * frames * S > 4e6:
        frames = max(1, int(4e6 / (n * S)))
    L = n + hop * (frames - 1) + int(rng.integers(0, max(1, min(hop, 50))))
    kw = dict(rows=rows, fmin_hz=float(rng.choice([20.0, 35.0, 80.0])), fmax_hz=float(rng.choice([24000.0, 18000.0, 8000.0])),
              gain=float(rng.choice([1.0, 3.5, 0.25])), db_range=float(rng.choice([80.0, 58.0])),
              gate_db=float(rng.choice([-80.0, -65.0])), power_floor=float(rng.choice([1e-14, 1e-10, 0.0])))
    kind = rng.integers(0, 4)
    pcm = synth.streams(S, L, first=int(rng.integers(0, 1000)))
    if kind == 1:
        pcm *= np.float32(10.0 ** rng.uniform(-3, 0.5))
    elif kind == 2:
        pcm[:, :: max(1, int(rng.integers(50, 5000)))] += np.float32(rng.uniform(0.1, 2.0))
    elif kind == 3:
        pcm = (pcm * 0).astype(np.float32) if rng.integers(0, 2) else np.sign(pcm).astype(np.float32) * np.float32(0.7)
    desc = f"case {ci}: n={n} hop={hop} rows={rows} re={reassign} S={S} frames={frames} L={L} kind={kind} {kw}"
    if n in (1024, 2048, 4096) and rng.integers(0, 2):
        os.environ["EMSPEC_SEGLEN"] = str(int(rng.integers(2, 140)))     # honoured by libemspec_diag.so only (EMSPEC_FUZZ_DIAG=1)
    else:
        os.environ.pop("EMSPEC_SEGLEN", None)
    try:
        with emspec.Engine(mode=emspec.MODE_EXACT, diag=bool(os.environ.get("EMSPEC_FUZZ_DIAG")), **kw) as e:
            out = e.batch(pcm, n, hop, reassign, want=("db", "index"))
            nf = min(frames, 4)
            pw, col, row, q = e.parity_dump_exact(pcm, n, hop, reassign, frames - nf, nf)
            stream_ok = True
            if ci % 4 == 0 and frames >= 2:
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
                stream_ok = sorted(got) == list(range(frames)) and np.array_equal(
                    np.stack([got[c] for c in range(frames)]).view(np.uint32), out["db"][0].view(np.uint32))
        cfg = O.make_cfg(n, hop, reassign, **kw)
        odb, _, oidx, _ = O.batch_exact(cfg, pcm, want=("db", "index"))
        bad = not stream_ok or not np.array_equal(out["db"].view(np.uint32), odb.view(np.uint32)) or not np.array_equal(out["index"], oidx)
        for s in range(S):
            opw, ocol, orow, oq = O.frames_exact(cfg, pcm[s], frames - nf, nf)
            if not (np.array_equal(col[s], ocol) and np.array_equal(row[s], orow) and np.array_equal(pw[s], opw) and np.array_equal(q[s], oq)):
                bad = True
                desc += f" [dump mismatch stream {s}: col {np.sum(col[s] != ocol)} row {np.sum(row[s] != orow)} pw {np.sum(pw[s] != opw)} q {np.sum(q[s] != oq)}]"
            if reassign and kw["power_floor"] > 0:      # the independent float64 method (its gate is P >= floor and P > 0)
                _, _, _, c64, r64 = O.frames_f64(cfg, pcm[s], frames - nf, nf)
                f64_bad += int(np.sum(col[s] != c64) + np.sum(row[s] != r64))
                f64_bins += 2 * c64.size
        if bad:
            fails += 1
            print("FAIL", desc, f"stream_ok={stream_ok}", flush=True)
    except Exception as ex:
        fails += 1
        print("EXC ", desc, repr(ex), flush=True)
    if ci % 25 == 24:
        print(f"... {ci + 1} cases, {fails} failures, {f64_bad} of {f64_bins} (column,row) values differ from the float64 method, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz_exact done: {cases} cases, {fails} failures, seed {seed}; vs the independent float64 method: {f64_bad} mismatches in {f64_bins} (column,row) values")
sys.exit(1 if fails else 0)
