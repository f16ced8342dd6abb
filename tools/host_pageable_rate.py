#!/usr/bin/env python3
"""emspec_batch / emspec_batch_packed from ORDINARY (pageable) host memory - what `new Float32Array` in Node or a numpy array
hands over - on the bench shape (64 streams x 2^22 samples, FFT 4096, hop 256, reassignment ON), beside the same calls from
page-locked memory.  Checks that both give the same bytes.
   python tools/host_pageable_rate.py [libemspec.so] [--exact]
(diagnostic build: EMSPEC_COPY_THREADS=k sets the number of copying threads)"""
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
n, hop, S, L = 4096, 256, 64, 1 << 22
R = e.rows
Cn = emspec.num_columns(L, n, hop)
base = synth.streams(8, L)
pcm = np.concatenate([base * (1.0 - 0.01 * i) for i in range(S // 8)], axis=0).astype(np.float32)
assert pcm.shape == (S, L)


def timed(fn, reps=3):
    fn()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best


def run(pcm_a, Sx, db=None, idx=None):
    o = emspec.Out(C.c_void_p(db.ctypes.data) if db is not None else None, None, C.c_void_p(idx.ctypes.data) if idx is not None else None)
    rc = lib.emspec_batch(e._h, C.c_void_p(pcm_a.ctypes.data), Sx, L, n, hop, 1, C.byref(o))
    assert rc == 0, e.last_error()


mode = "exact" if exact else "fast "
# index out, pageable
idx = np.empty((S, Cn, R), np.uint8)
dt = timed(lambda: run(pcm, S, idx=idx))
print(f"emspec_batch {mode} pageable, uint8 palette index out: {S * Cn / dt:.3e} columns/s ({dt * 1e3:.1f} ms; {pcm.nbytes / dt / 1e9:.1f} GB/s in + {idx.nbytes / dt / 1e9:.1f} GB/s out)", flush=True)
# ... into a FRESH array each call (np.empty: the pages are first touched by the copy-out - what a caller that allocates its result per
# call pays; r05's "pageable 4.8e6" was this)
def fresh():
    run(pcm, S, idx=np.empty((S, Cn, R), np.uint8))
dtf = timed(fresh)
print(f"emspec_batch {mode} pageable, index out into a fresh array each call: {S * Cn / dtf:.3e} columns/s ({dtf * 1e3:.1f} ms)", flush=True)
# the same from page-locked memory
pin_in = emspec.PinnedArray(pcm.shape, np.float32)
pin_idx = emspec.PinnedArray(idx.shape, np.uint8)
pin_in.array[...] = pcm
dtp = timed(lambda: run(pin_in.array, S, idx=pin_idx.array))
print(f"emspec_batch {mode} pinned,   uint8 palette index out: {S * Cn / dtp:.3e} columns/s ({dtp * 1e3:.1f} ms)", flush=True)
if exact:
    assert np.array_equal(idx, pin_idx.array), "pageable and pinned index columns differ"
else:
    d = np.abs(idx.astype(np.int16) - pin_idx.array.astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() < 1e-4, "pageable and pinned index columns differ beyond the float32 mode's run-to-run spread"
# float32 dB out on 16 streams, pageable
Sd = 16
db = np.empty((Sd, Cn, R), np.float32)
dt = timed(lambda: run(pcm[:Sd], Sd, db=db))
print(f"emspec_batch {mode} pageable, float32 dB out ({Sd} streams): {Sd * Cn / dt:.3e} columns/s ({dt * 1e3:.1f} ms; {db.nbytes / dt / 1e9:.1f} GB/s out)", flush=True)
# packed images, pageable
wire = np.empty(S * emspec.wire_bound(Cn, R), np.uint8)
offs = np.zeros(S + 1, np.int64)
def packed():
    rc = lib.emspec_batch_packed(e._h, C.c_void_p(pcm.ctypes.data), S, L, n, hop, 1, C.c_void_p(wire.ctypes.data), C.c_int64(wire.size),
                                 offs.ctypes.data_as(C.c_void_p))
    assert rc == 0, e.last_error()
dt = timed(packed)
print(f"emspec_batch_packed {mode} pageable: {S * Cn / dt:.3e} columns/s ({dt * 1e3:.1f} ms; {int(offs[S]) / (S * Cn):.0f} B per column out)", flush=True)
img = emspec.wire_unpack_host(wire[offs[3]:offs[4]], Cn, R)
assert np.array_equal(img.reshape(Cn, R), idx[3]) or not exact, "packed image of stream 3 differs from the index columns"
