#!/usr/bin/env python3
"""In-kernel shader clock of the fused kernels (MI355X_MICROARCH.md "DVFS give-back" item 6): delta s_memtime / delta
s_memrealtime x 100 MHz, stamped around the frame loop of the diagnostic build (libemspec_diag.so) after >= 2 s of
back-to-back launches of the product kernel on the bench input.  usage: tools/kernel_clock.py [4096|16384] [streams]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "em-spec_amd")]
import numpy as np
import torch

import emspec
from bench import synth_device

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
hop = 256 if n == 4096 else 512
L = 1 << 22
eng = emspec.Engine(diag=True)
lib = emspec.load(diag=True)
dev = torch.device("cuda", 0)
base = synth_device(min(S, 8), L, 0, dev)
pcm = base.repeat((S + base.shape[0] - 1) // base.shape[0], 1)[:S].contiguous()
Cn = emspec.num_columns(L, n, hop)
db = torch.empty((S, Cn, 1024), dtype=torch.float32, device=dev)
idx = torch.empty((S, Cn, 1024), dtype=torch.uint8, device=dev)
t0 = time.perf_counter()
launches = 0
while time.perf_counter() - t0 < 2.5:          # warm the DVFS state on random data, product kernel, back to back
    for _ in range(10):
        eng.batch_device(pcm, n, hop, True, db=db, index=idx)
    torch.cuda.synchronize()
    launches += 10
wall = (time.perf_counter() - t0) / launches
groups, waves = C.c_int64(0), C.c_int32(0)
f = lib.emspec_debug_phase_cycles
f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
              C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
assert f(eng._h, pcm.data_ptr(), S, L, n, hop, 1, db.data_ptr(), idx.data_ptr(), None, C.byref(groups), C.byref(waves)) == 0
cyc = np.zeros((groups.value, waves.value, 8), np.uint64)
assert f(eng._h, pcm.data_ptr(), S, L, n, hop, 1, db.data_ptr(), idx.data_ptr(), cyc.ctypes.data, C.byref(groups), C.byref(waves)) == 0
shader = cyc[:, :, :7].sum(axis=2).astype(np.float64)
ticks = (cyc[:, :, 7] >> np.uint64(32)).astype(np.float64)
ghz = shader / ticks * 0.1
print(f"N={n} hop={hop}, {S} streams: {groups.value} workgroups x {waves.value} waves; product kernel {wall * 1e3:.2f} ms per launch "
      f"({S * Cn / wall:.3g} col/s) over {launches} back-to-back launches")
print(f"in-kernel shader clock (stamped build, per wave: shader cycles of the frame loop / 100 MHz ticks): "
      f"median {np.median(ghz):.3f} GHz, p5 {np.percentile(ghz, 5):.3f}, p95 {np.percentile(ghz, 95):.3f}")
print(f"walk duration per workgroup: median {np.median(ticks) / 100:.1f} us")
