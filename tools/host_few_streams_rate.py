#!/usr/bin/env python3
"""emspec_batch from host buffers with FEW streams (BASELINE configs[1] is one stream): 1, 2 and 4 streams of 2^22 and 2^25 samples,
FFT 4096 / hop 256, palette index out, page-locked and ordinary arrays - beside the device-resident rate of the same shape.  With
fewer streams than the pipeline has stages the library cuts a stream's columns into runs (round 6); checks the columns against
the device-resident call's.
   python tools/host_few_streams_rate.py [libemspec.so] [--exact]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "em-spec_amd")]
import ctypes as C
import numpy as np
import torch
import emspec
from emspec import synth

args = [a for a in sys.argv[1:] if not a.startswith("--")]
if args:
    emspec.LIB_PATH = os.path.abspath(args[0])
exact = "--exact" in sys.argv
lib = emspec.load()
e = emspec.Engine(mode=emspec.MODE_EXACT if exact else emspec.MODE_FAST)
n, hop, R = 4096, 256, e.rows


def best_of(fn, reps=5):
    fn()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best


for S, L in ((1, 1 << 22), (1, 1 << 25), (2, 1 << 22), (4, 1 << 22), (4, 1 << 25)):
    Cn = emspec.num_columns(L, n, hop)
    one = synth.streams(1, L)[0]
    pcm = np.stack([one * (1.0 - 0.1 * s) for s in range(S)]).astype(np.float32)
    x = torch.from_numpy(pcm).cuda()
    dix = torch.empty((S, Cn, R), dtype=torch.uint8, device="cuda")
    def dev():
        e.batch_device(x, n, hop, True, index=dix)
        torch.cuda.synchronize()
    t_dev = best_of(dev)
    ref = dix.cpu().numpy()
    line = f"{'exact' if exact else 'fast '} S={S} L=2^{L.bit_length() - 1} ({Cn} columns): device-resident {S * Cn / t_dev:.3e}"
    for kind in ("pinned", "pageable"):
        if kind == "pinned":
            pin, pix = emspec.PinnedArray(pcm.shape, np.float32), emspec.PinnedArray((S, Cn, R), np.uint8)
            pin.array[...] = pcm
            a_in, a_out = pin.array, pix.array
        else:
            a_in, a_out = pcm, np.empty((S, Cn, R), np.uint8)
        a_out[...] = 255
        o = emspec.Out(None, None, C.c_void_p(a_out.ctypes.data))
        def host():
            rc = lib.emspec_batch(e._h, C.c_void_p(a_in.ctypes.data), S, L, n, hop, 1, C.byref(o))
            assert rc == 0, e.last_error()
        t = best_of(host)
        if exact:
            assert np.array_equal(a_out, ref), f"{kind}: differs from the device-resident call"
        else:
            d = np.abs(a_out.astype(np.int16) - ref.astype(np.int16))
            assert d.max() <= 1 and (d != 0).mean() < 1e-4, f"{kind}: differs from the device-resident call"
        line += f" | {kind} {S * Cn / t:.3e} ({t * 1e3:.2f} ms)"
        if kind == "pinned":
            pin.close(); pix.close()
    print(line + " columns/s", flush=True)
