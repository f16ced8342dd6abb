#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point and latency of the streaming call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "em-spec_amd")]
import numpy as np
import emspec
from emspec import synth

e = emspec.Engine()
n, hop = 4096, 256
pcm = np.repeat(synth.streams(1, 1 << 22), 8, axis=0)
def best_of(fn, reps=3):
    """(a process's first calls of a kind run slow - allocations, the copy queues' first use: one warm call, then the best of 3)"""
    fn()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best

e.batch(pcm[:1], n, hop, True)
out = e.batch(pcm, n, hop, True, want=("db",))
C = out["db"].shape[1]
dt = best_of(lambda: e.batch(pcm, n, hop, True, want=("db",), db_out=out["db"]))
print(f"emspec_batch (host buffers, pageable and reused, PCIe in+out): {8 * C / dt:.3e} columns/s "
      f"({dt * 1e3:.1f} ms for 8 streams x {C} columns; {pcm.nbytes / 1e6:.0f} MB in, {out['db'].nbytes / 1e6:.0f} MB out)")
pin_in = emspec.PinnedArray(pcm.shape, np.float32)
pin_out = emspec.PinnedArray(out["db"].shape, np.float32)
pin_in.array[...] = pcm
dt = best_of(lambda: e.batch(pin_in.array, n, hop, True, want=("db",), db_out=pin_out.array))
assert np.max(np.abs(pin_out.array - out["db"])) < 2e-4
print(f"emspec_batch (host buffers from emspec_host_alloc, pinned, PCIe in+out): {8 * C / dt:.3e} columns/s ({dt * 1e3:.1f} ms)")
# the 1-byte palette index instead of float32 dB: a quarter of the bytes back over PCIe (what throughput
# callers should take)
pin_idx = emspec.PinnedArray(out["db"].shape, np.uint8)
lib = emspec.load()
import ctypes as C_
o = emspec.Out(None, None, C_.c_void_p(pin_idx.array.ctypes.data))
def run_idx():
    assert lib.emspec_batch(e._h, C_.c_void_p(pin_in.array.ctypes.data), 8, pcm.shape[1], n, hop, 1, C_.byref(o)) == 0
dt = best_of(run_idx)
print(f"emspec_batch (pinned host buffers, uint8 palette index out instead of float32 dB): {8 * C / dt:.3e} columns/s ({dt * 1e3:.1f} ms; "
      f"{pin_idx.array.nbytes / 1e6:.0f} MB out)")
e.reset()
fr = pcm[0]
for j in range(20):
    e.column(fr[j * hop:j * hop + n], hop, True)
t0 = time.perf_counter()
for j in range(20, 520):
    e.column(fr[j * hop:j * hop + n], hop, True)
dt = (time.perf_counter() - t0) / 500
print(f"emspec_column (streaming, one frame per call: one launch reading and writing page-locked host memory + sync): {dt * 1e6:.1f} us per column "
      f"= {1 / dt:.0f} columns/s per engine (real time needs 187.5/s per stream)")
# where the per-call time goes: (a) the same call through raw ctypes with preallocated arrays (no numpy allocation, no
# wrapper), (b) the fixed cost of one launch + one stream synchronisation, read off the smallest frame (N = 256: the kernel
# itself is ~2 us), (c) the EXACT mode's call (two launches + copies)
import ctypes as C_
lib = emspec.load()
dbo = np.empty(e.rows, np.float32)
col = C_.c_int64(0)
def raw(eng, frame, nn, hh):
    return lib.emspec_column(eng._h, C_.c_void_p(frame.ctypes.data), nn, hh, 1, C_.c_void_p(dbo.ctypes.data), None, eng.rows, C_.byref(col))
e.reset()
frames = [np.ascontiguousarray(fr[j * hop:j * hop + n]) for j in range(520)]
for j in range(20):
    raw(e, frames[j], n, hop)
t0 = time.perf_counter()
for j in range(20, 520):
    raw(e, frames[j], n, hop)
dt_raw = (time.perf_counter() - t0) / 500
e.reset()
small = [np.ascontiguousarray(fr[j * 64:j * 64 + 256]) for j in range(520)]
for j in range(20):
    raw(e, small[j], 256, 64)
t0 = time.perf_counter()
for j in range(20, 520):
    raw(e, small[j], 256, 64)
dt_small = (time.perf_counter() - t0) / 500
print(f"emspec_column through raw ctypes (no wrapper allocations): {dt_raw * 1e6:.1f} us per column at N = 4096; {dt_small * 1e6:.1f} us at N = 256 "
      f"(N = 256: frame launch + the 256-thread finalize launch; = the fixed cost of launching + one stream synchronisation + "
      f"the host copies; a hipGraph replay costs 10-16 us on this stack against 3-5 us for a direct launch - MI355X_MICROARCH.md "
      f"'graph-replay-floor' - so capturing this single launch cannot lower it)")
with emspec.Engine(mode=emspec.MODE_EXACT) as x:
    for j in range(20):
        raw(x, frames[j], n, hop)
    t0 = time.perf_counter()
    for j in range(20, 520):
        raw(x, frames[j], n, hop)
    dt_x = (time.perf_counter() - t0) / 500
print(f"emspec_column, EXACT mode (binary64, u64 ring in HBM; one launch, as the float32 mode since round 6): {dt_x * 1e6:.1f} us per column")
for blk in (128, 512, 2048, 16384, 131072):
    e.reset()
    e.push_samples(fr[:n + 16 * hop], n, hop, True)
    pos, cols = n + 16 * hop, 0
    t0 = time.perf_counter()
    while pos + blk <= min(fr.size, n + 16 * hop + 400 * blk):
        db, _ = e.push_samples(fr[pos:pos + blk], n, hop, True)
        cols += len(db)
        pos += blk
    dt = time.perf_counter() - t0
    print(f"emspec_push_samples (streaming, blocks of {blk} samples = {blk / hop:g} columns per call): "
          f"{dt / max(cols, 1) * 1e6:.1f} us per column = {cols / dt:.3e} columns/s per engine")
e.reset()
