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
e.batch(pcm[:1], n, hop, True)
t0 = time.perf_counter()
out = e.batch(pcm, n, hop, True, want=("db",))
dt = time.perf_counter() - t0
C = out["db"].shape[1]
print(f"emspec_batch (host buffers, pageable, PCIe in+out): {8 * C / dt:.3e} columns/s "
      f"({dt * 1e3:.1f} ms for 8 streams x {C} columns; {pcm.nbytes / 1e6:.0f} MB in, {out['db'].nbytes / 1e6:.0f} MB out)")
pin_in = emspec.PinnedArray(pcm.shape, np.float32)
pin_out = emspec.PinnedArray(out["db"].shape, np.float32)
pin_in.array[...] = pcm
e.batch(pin_in.array, n, hop, True, want=("db",), db_out=pin_out.array)
t0 = time.perf_counter()
e.batch(pin_in.array, n, hop, True, want=("db",), db_out=pin_out.array)
dt = time.perf_counter() - t0
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
run_idx()
t0 = time.perf_counter()
run_idx()
dt = time.perf_counter() - t0
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
