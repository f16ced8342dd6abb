#!/usr/bin/env python3
"""Randomised parity stress of the LIVE multi-stream entry points (emspec_columns / emspec_push_samples_multi /
emspec_columns_flush / emspec_reset_stream) through the C ABI against the oracle's BATCH columns of the same streams: random mode,
shape, rows, stream count, feeding form, block sizes (sub-hop blocks, blocks longer than the staging block), page-locked or
pageable buffers, and a mid-session restart of one stream.  EXACT mode: dB bits and RGBA array_equal; float32 mode: dB within
8.7e-4, RGBA differing on < 1e-3 of the cells.
usage: python tools/fuzz_live.py [cases] [seed]     (needs an MI355X; prints one line per failure)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "em-spec_amd"), os.path.join(ROOT, "oracle")]
import numpy as np
import emspec, oracle as O
from emspec import synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
fails = 0
t0 = time.time()


def oracle(cfg, pcm, exact):
    if exact:
        db, rgba, _, _ = O.batch_exact(cfg, pcm, want=("db", "rgba"))
    else:
        db, rgba, _ = O.batch_f32(cfg, pcm, want=("db", "rgba"))
    return db, rgba


def same(got_db, got_rgba, odb, orgba, exact):
    if exact:
        return np.array_equal(got_db.view(np.uint32), odb.view(np.uint32)) and np.array_equal(got_rgba, orgba)
    return float(np.max(np.abs(got_db - odb))) < 8.7e-4 and float(np.mean(got_rgba != orgba)) < 1e-3


for ci in range(cases):
    exact = bool(rng.integers(0, 2))
    n = int(rng.choice([256, 512, 1024, 2048, 4096, 4096, 8192, 16384]))
    hop = int(rng.choice([n // 16, n // 8, n // 4, n // 2, n, int(rng.integers(max(1, n // 32), n + 1))]))
    rows = int(rng.choice([64, 128, 256, 512, 1000, 1024, 1024, 2048]))
    reassign = bool(rng.integers(0, 4))
    S = int(rng.integers(1, 9)) if n < 16384 else int(rng.integers(1, 4))
    D = -(-n // (2 * hop)) if reassign else 0
    frames = int(rng.integers(1, 50)) + (D if rng.integers(0, 2) else 0)
    if n * S * frames > 6e6:
        frames = max(1, int(6e6 / (n * S)))
    L = n + hop * (frames - 1)
    form = "frames" if rng.integers(0, 2) else "samples"
    pinned = bool(rng.integers(0, 2))
    who = int(rng.integers(0, S)) if (S > 1 and frames > D + 6 and rng.integers(0, 2)) else -1   # the stream that restarts
    cut = int(rng.integers(1, frames - D - 2)) if who >= 0 else 0                                   # ... when frame `cut` is due
    kw = dict(rows=rows, fmin_hz=float(rng.choice([20.0, 35.0])), fmax_hz=float(rng.choice([24000.0, 16000.0])),
              gain=float(rng.choice([1.0, 3.5])), db_range=float(rng.choice([80.0, 58.0])), gate_db=float(rng.choice([-80.0, -65.0])))
    pcm = synth.streams(S, L, first=int(rng.integers(0, 1000)))
    fresh = synth.streams(1, L, first=int(rng.integers(1000, 2000)))[0]
    desc = f"case {ci}: exact={exact} n={n} hop={hop} rows={rows} re={reassign} S={S} frames={frames} form={form} pinned={pinned} who={who} cut={cut}"
    try:
        cfg = O.make_cfg(n, hop, reassign, **kw)
        odb, orgba = oracle(cfg, pcm, exact)
        # what the restarted stream is fed afterwards: per-frame form frames - cut whole frames; per-sample form the L - cut_at
        # samples that are left (fewer frames: the new stream needs n samples before its first frame is complete)
        cut_at = n + hop * (cut - 1) if who >= 0 else -1          # samples fed when frame cut - 1 is complete
        new_len = (n + hop * (frames - cut - 1)) if form == "frames" else (L - cut_at)
        nn = ((new_len - n) // hop + 1) if (who >= 0 and new_len >= n) else 0
        ndb, nrgba = (oracle(cfg, fresh[None, :new_len], exact) if nn > 0 else (None, None))
        got_db = np.zeros((S, frames, rows), np.float32)
        got_rgba = np.zeros((S, frames, rows, 4), np.uint8)
        new_db, new_rgba = {}, {}
        seen = [set() for _ in range(S)]
        keep = []

        def buf(shape, dt):
            if pinned and int(np.prod(shape)) > 0:
                p = emspec.PinnedArray(shape, dt)
                keep.append(p)
                return p.array
            return np.empty(shape, dt)

        def take(s, c, db, rgba, restarted):
            if restarted:
                new_db[c], new_rgba[c] = db.copy(), rgba.copy()
            else:
                got_db[s, c], got_rgba[s, c] = db, rgba
                seen[s].add(c)
        with emspec.Engine(mode=emspec.MODE_EXACT if exact else emspec.MODE_FAST, **kw) as e:
            if form == "frames":
                fin, odb_b, orgba_b = buf((S, n), np.float32), buf((S, rows), np.float32), buf((S, rows, 4), np.uint8)
                for j in range(frames):
                    if j == cut and who >= 0:
                        e.reset_stream(who)
                    fin[:] = pcm[:, j * hop:j * hop + n]
                    if who >= 0 and j >= cut:
                        fin[who] = fresh[(j - cut) * hop:(j - cut) * hop + n]
                    db, rgba, cols = e.columns(fin, hop, reassign, want_rgba=True, db=odb_b, rgba=orgba_b)
                    for s in range(S):
                        if cols[s] >= 0:
                            take(s, int(cols[s]), db[s], rgba[s], s == who and j >= cut)
            else:
                feed = pcm.copy()
                pos = 0
                restarted = False
                while pos < L:
                    blk = int(rng.choice([hop, hop, max(1, hop // 2), 128, int(rng.integers(1, 3 * n)), int(rng.integers(1, 40 * hop + 1))]))
                    cnt = min(blk, L - pos)
                    if who >= 0 and not restarted and pos <= cut_at < pos + cnt:
                        cnt = cut_at - pos                                   # stop the block at the restart point
                    if cnt == 0:
                        e.reset_stream(who)
                        feed[who, pos:] = fresh[:L - pos]
                        restarted = True
                        continue
                    k = e.push_columns_multi(cnt, n, hop, reassign)
                    sin = buf((S, cnt), np.float32)
                    sin[:] = feed[:, pos:pos + cnt]
                    db, rgba, counts, firsts = e.push_samples_multi(sin, n, hop, reassign, want_rgba=True,
                                                                    db=buf((S, k, rows), np.float32), rgba=buf((S, k, rows, 4), np.uint8))
                    assert int(counts.max(initial=0)) == k
                    for s in range(S):
                        for i in range(int(counts[s])):
                            take(s, int(firsts[s]) + i, db[s, i], rgba[s, i], s == who and restarted)
                    pos += cnt
                    for p in keep:
                        p.close()
                    keep.clear()
            while True:
                try:
                    db, rgba, cols = e.columns_flush(want_rgba=True)
                except emspec.EmspecError:
                    break
                for s in range(S):
                    if cols[s] >= 0:
                        take(s, int(cols[s]), db[s], rgba[s], s == who)
        for p in keep:
            p.close()
        bad = ""
        for s in range(S):
            if s == who:
                upto = max(cut - D, 0)                     # the old audio's columns that were complete at the restart
                ok = set(range(upto)) <= seen[s] and (upto == 0 or same(got_db[s, :upto], got_rgba[s, :upto], odb[s, :upto], orgba[s, :upto], exact))
                ok = ok and sorted(new_db) == list(range(nn)) and (nn == 0 or same(np.stack([new_db[c] for c in range(nn)]),
                                                                                   np.stack([new_rgba[c] for c in range(nn)]), ndb[0], nrgba[0], exact))
            else:
                ok = seen[s] == set(range(frames)) and same(got_db[s], got_rgba[s], odb[s], orgba[s], exact)
            if not ok:
                bad += f" [stream {s}]"
        if bad:
            fails += 1
            print("FAIL", desc, bad, flush=True)
    except Exception as ex:
        fails += 1
        print("EXC ", desc, repr(ex), flush=True)
    if ci % 20 == 19:
        print(f"... {ci + 1} cases, {fails} failures, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz_live done: {cases} cases, {fails} failures, seed {seed}")
sys.exit(1 if fails else 0)
