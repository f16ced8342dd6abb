#!/usr/bin/env python3
"""emspec_batch from host buffers on SMALL and MID-SIZE batches (FFT 4096 / hop 256, palette index out; median of 30 calls): what a
call costs when the copies and kernels are short beside the ~0.2 ms the host spends submitting a unit of the pipeline.  The number of
units follows the batch (pipe_units, emspec_api.cpp) since late round 6; before: one unit per stream up to sixteen.
   python tools/host_small_batch_rate.py [libemspec.so] [--exact]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "em-spec_amd")]
import ctypes as C
import numpy as np
import emspec
from emspec import synth

args = [a for a in sys.argv[1:] if not a.startswith("--")]
if args:
    emspec.LIB_PATH = os.path.abspath(args[0])
exact = "--exact" in sys.argv
lib = emspec.load()
e = emspec.Engine(mode=emspec.MODE_EXACT if exact else emspec.MODE_FAST)
n, hop = 4096, 256
for S, L in ((2, 1 << 14), (4, 1 << 16), (8, 1 << 18), (8, 1 << 20), (16, 1 << 20), (16, 1 << 22), (32, 1 << 20)):
    pcm = synth.streams(S, L)
    Cn = emspec.num_columns(L, n, hop)
    res, outs = {}, {}
    for kind in ("pageable", "pinned"):
        if kind == "pinned":
            pin, pix = emspec.PinnedArray(pcm.shape, np.float32), emspec.PinnedArray((S, Cn, e.rows), np.uint8)
            pin.array[...] = pcm
            a, b = pin.array, pix.array
        else:
            a, b = pcm, np.empty((S, Cn, e.rows), np.uint8)
        o = emspec.Out(None, None, C.c_void_p(b.ctypes.data))
        def run():
            assert lib.emspec_batch(e._h, C.c_void_p(a.ctypes.data), S, L, n, hop, 1, C.byref(o)) == 0
        for _ in range(3):
            run()
        ts = []
        for _ in range(30):
            t0 = time.perf_counter()
            run()
            ts.append(time.perf_counter() - t0)
        res[kind] = sorted(ts)[len(ts) // 2]
        outs[kind] = b.copy()
    d = np.abs(outs["pageable"].astype(np.int16) - outs["pinned"].astype(np.int16))
    assert (d.max() == 0) if exact else (d.max() <= 1), "pageable and pinned calls differ"
    mb = (pcm.nbytes + S * Cn * e.rows) / 1e6
    print(f"{'exact' if exact else 'fast '} S={S:2d} L=2^{L.bit_length() - 1} ({S * Cn:7d} columns, {mb:6.1f} MB in+out): pageable {res['pageable'] * 1e6:6.0f} us = "
          f"{S * Cn / res['pageable']:.2e} columns/s, pinned {res['pinned'] * 1e6:6.0f} us = {S * Cn / res['pinned']:.2e}", flush=True)
