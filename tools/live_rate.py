#!/usr/bin/env python3
"""Per-call host time of the live multi-stream entry points (emspec_columns / emspec_push_samples_multi) on the GPU box:
S streams x one hop per call, N = 4096 / hop 256 (BASELINE configs[2] in its live form).  Prints one line per variant:
median / p90 microseconds per call and columns/s.  usage: live_rate.py [S] [calls] [filter,filter,...]
(filters: every token must appear in the variant's label, e.g. FAST,samples,pinned,out=db)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "em-spec_amd"),):
    sys.path.insert(0, p)
import emspec  # noqa: E402
from emspec import synth  # noqa: E402


def timed(fn, calls):
    ts = np.empty(calls)
    for i in range(calls):
        t0 = time.perf_counter()
        fn(i)
        ts[i] = time.perf_counter() - t0
    return ts[calls // 10:]          # the first tenth warms up (ring priming, allocations)


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    only = sys.argv[3].split(",") if len(sys.argv) > 3 else []
    n, hop, R = 4096, 256, 1024
    L = n + hop * (calls + 4)
    pcm = synth.streams(min(S, 8), L)
    pcm = np.ascontiguousarray(np.tile(pcm, ((S + 7) // 8, 1))[:S])
    for mode, mname in ((emspec.MODE_FAST, "FAST"), (emspec.MODE_EXACT, "EXACT")):
        for pinned in (True, False):
            for form in ("samples", "frames"):
                for outs in (("db",), ("rgba",), ("db", "rgba")):
                    label = f"{mname:5s} S={S:3d} {form:7s} {'pinned' if pinned else 'pageable':8s} out={'+'.join(outs):7s} "
                    if not all(tok in label + " " for tok in only):
                        continue
                    with emspec.Engine(mode=mode) as e:
                        keep = []
                        def buf(shape, dt):
                            if pinned:
                                p = emspec.PinnedArray(shape, dt)
                                keep.append(p)
                                return p.array
                            return np.empty(shape, dt)
                        if form == "frames":
                            fin = buf((S, n), np.float32)
                            db = buf((S, R), np.float32) if "db" in outs else None
                            rgba = buf((S, R, 4), np.uint8) if "rgba" in outs else None
                            def call(i):
                                fin[:] = pcm[:, i * hop:i * hop + n]
                            def run(i):
                                e.columns(fin, hop, True, want_db=db is not None, want_rgba=rgba is not None, db=db, rgba=rgba)
                            # the copy into the frame block is the caller's (the renderer owns its frames): not timed
                            def step(i):
                                call(i)
                            ts = np.empty(calls)
                            for i in range(calls):
                                step(i)
                                t0 = time.perf_counter()
                                run(i)
                                ts[i] = time.perf_counter() - t0
                            ts = ts[calls // 10:]
                        else:
                            sin = buf((S, hop), np.float32)
                            db = buf((S, 1, R), np.float32) if "db" in outs else None
                            rgba = buf((S, 1, R, 4), np.uint8) if "rgba" in outs else None
                            e.push_samples_multi(pcm[:, :n - hop].copy(), n, hop, True, want_db=False)   # prime: no frame yet
                            ts = np.empty(calls)
                            for i in range(calls):
                                sin[:] = pcm[:, n - hop + i * hop:n + i * hop]
                                t0 = time.perf_counter()
                                e.push_samples_multi(sin, n, hop, True, want_db=db is not None, want_rgba=rgba is not None, db=db, rgba=rgba)
                                ts[i] = time.perf_counter() - t0
                            ts = ts[calls // 10:]
                        med, p90 = np.median(ts) * 1e6, np.percentile(ts, 90) * 1e6
                        print(f"{label}"
                              f"median {med:7.1f} us  p90 {p90:7.1f} us  -> {S / (med * 1e-6):.3e} columns/s", flush=True)
                        for p in keep:
                            p.close()


if __name__ == "__main__":
    main()
