#!/usr/bin/env python3
"""Where a live multi-stream call's kernel time goes (diagnostic build): per-phase wall-clock stamps of the frame kernel,
S streams x one hop per call, N = 4096 / hop 256.  usage: live_phases.py [S] [calls] [exact]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "em-spec_amd"))
import emspec  # noqa: E402
from emspec import synth  # noqa: E402

NAMES = ["descriptor", "samples", "transform", "bins+scatter", "ticket", "finalize"]


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    exact = len(sys.argv) > 3 and sys.argv[3] == "exact"
    n, hop = 4096, 256
    pcm = np.ascontiguousarray(np.tile(synth.streams(min(S, 8), n + hop * (calls + 2)), ((S + 7) // 8, 1))[:S])
    st = emspec.PinnedArray((S, 8), np.uint64)
    sin = emspec.PinnedArray((S, hop), np.float32)
    db = emspec.PinnedArray((S, 1, 1024), np.float32)
    with emspec.Engine(diag=True, mode=emspec.MODE_EXACT if exact else emspec.MODE_FAST) as e:
        lib = emspec.load(diag=True)
        lib.emspec_debug_live_stamps.argtypes = [C.c_void_p, C.c_void_p]
        assert lib.emspec_debug_live_stamps(e._h, C.c_void_p(st.array.ctypes.data)) == 0
        e.push_samples_multi(pcm[:, :n - hop].copy(), n, hop, True, want_db=False)
        rows = []
        for i in range(calls):
            sin.array[:] = pcm[:, n - hop + i * hop:n + i * hop]
            st.array[:] = 0
            e.push_samples_multi(sin.array, n, hop, True, db=db.array)
            a = st.array.astype(np.int64)
            if i >= calls // 10 and np.all(a[:, :7] > 0):
                rows.append(a[:, :7].copy())
        a = np.stack(rows)                                    # [calls][S][7] in 10 ns ticks
        t0 = a[:, :, 0].min(axis=1)[:, None, None]
        print(f"{'EXACT' if exact else 'FAST'}: {S} streams, {len(rows)} calls; microseconds, median over calls and streams")
        d = np.diff(a, axis=2) / 100.0
        for k, nm in enumerate(NAMES):
            print(f"  {nm:14s} {np.median(d[:, :, k]):7.2f} us   (p90 {np.percentile(d[:, :, k], 90):7.2f})")
        print(f"  first workgroup starts -> last one ends: {np.median((a[:, :, 6].max(axis=1) - a[:, :, 0].min(axis=1)) / 100.0):.2f} us; "
              f"entry spread {np.median((a[:, :, 0].max(axis=1) - a[:, :, 0].min(axis=1)) / 100.0):.2f} us")


if __name__ == "__main__":
    main()
