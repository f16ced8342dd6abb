#!/usr/bin/env python3
"""Diagnostic: per-phase shader-clock cycles of the fused EXACT-mode kernel (stamped build of exact_fused4096_kernel in
libemspec_diag.so, emspec_debug_phase_cycles on an EXACT engine).  Per wave and FRAME (one frame = two half-iterations:
one in role F, one in role V).
   python tools/phase_cycles_exact.py [streams] [waves]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "em-spec_amd")]
import numpy as np
import torch

import emspec
from bench import synth_device

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = 1 << 22
n, hop = 4096, 256
eng = emspec.Engine(diag=True, mode=emspec.MODE_EXACT)
lib = emspec.load(diag=True)
dev = torch.device("cuda", 0)
pcm = synth_device(S, L, 0, dev)
Cn = emspec.num_columns(L, n, hop)
db = torch.empty((S, Cn, 1024), dtype=torch.float32, device=dev)
idx = torch.empty((S, Cn, 1024), dtype=torch.uint8, device=dev)
groups = C.c_int64(0)
waves = C.c_int32(0)
f = lib.emspec_debug_phase_cycles
f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
              C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
assert f(eng._h, pcm.data_ptr(), S, L, n, hop, 1, db.data_ptr(), idx.data_ptr(), None, C.byref(groups), C.byref(waves)) == 0
torch.cuda.synchronize()
cyc = np.zeros((groups.value, waves.value, 8), np.uint64)
for _ in range(2):   # second run: clock warm
    assert f(eng._h, pcm.data_ptr(), S, L, n, hop, 1, db.data_ptr(), idx.data_ptr(), cyc.ctypes.data, C.byref(groups), C.byref(waves)) == 0
# exact_fused4096_lr_kernel (round 5: no parking; EMSPEC_EXACT_PARKED=1 stamps round 4's kernel instead)
if os.environ.get("EMSPEC_EXACT_PARKED") == "1":
    names = ["F1 scatter + take column", "F2 park, write A", "F3 passes B C D", "F4 spectrum reads", "F5 unpark",
             "F  sync waits (5)", "V  work (bins, dB, pass A)", "V  barrier waits"]
else:
    names = ["F1 take column + scatter", "F1 write A", "F2 passes B C D (+ dB)", "F3 spectrum reads", "-",
             "F  sync waits (2) + barrier", "V  work (bins, pass A)", "V  barrier wait"]
ticks = (cyc[:, :, 7] >> np.uint64(32)).astype(np.float64)          # 100 MHz
cyc[:, :, 7] &= np.uint64(0xFFFFFFFF)
c = cyc.astype(np.float64)
frames = Cn / (groups.value / S) + 2 * 8 + 4      # half-iterations per workgroup = frames per team x 2
tot = c.sum(axis=2)
print(f"groups {groups.value}, {frames:.0f} half-iterations each; mean cycles per wave {tot.mean():.0f}; "
      f"per frame (2 half-iterations) {2 * tot.mean() / frames:.0f}; in-kernel clock {tot.mean() / ticks.mean() / 10:.3f} GHz")
for i, nm in enumerate(names):
    print(f"  {nm:26s} {100 * c[:, :, i].sum() / tot.sum():5.1f} %   per frame {2 * c[:, :, i].mean() / frames:8.0f}")
if len(sys.argv) > 2:
    print("per-wave mean cycles per frame (rows: wave; cols: the phases above)")
    for w in range(waves.value):
        print(f"  wave {w:2d} " + " ".join(f"{2 * c[:, w, i].mean() / frames:7.0f}" for i in range(8)))
