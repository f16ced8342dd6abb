#!/usr/bin/env python3
"""Diagnostic: per-phase shader-clock cycles of the fused N = 16384 kernel (stamped build in libemspec_diag.so,
emspec_debug_phase_cycles).  usage: tools/phase_cycles_n16384.py [streams] [v]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "em-spec_amd")]
import numpy as np
import torch

import emspec
from bench import synth_device

S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L = 1 << 22
n, hop = 16384, 512
eng = emspec.Engine(diag=True)
lib = emspec.load(diag=True)
dev = torch.device("cuda", 0)
pcm = synth_device(S, L, 0, dev)
Cn = emspec.num_columns(L, n, hop)
db = torch.empty((S, Cn, 1024), dtype=torch.float32, device=dev)
idx = torch.empty((S, Cn, 1024), dtype=torch.uint8, device=dev)
groups, waves = C.c_int64(0), C.c_int32(0)
f = lib.emspec_debug_phase_cycles
f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
              C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
assert f(eng._h, pcm.data_ptr(), S, L, n, hop, 1, db.data_ptr(), idx.data_ptr(), None, C.byref(groups), C.byref(waves)) == 0
torch.cuda.synchronize()
cyc = np.zeros((groups.value, waves.value, 8), np.uint64)
for _ in range(2):
    assert f(eng._h, pcm.data_ptr(), S, L, n, hop, 1, db.data_ptr(), idx.data_ptr(), cyc.ctypes.data, C.byref(groups), C.byref(waves)) == 0
names = ["pass-1 write", "barrier wait", "mid passes+last+rewrite", "bins", "ring out+prefetch", "scatter+next pass 1", "finalize+ring in", "-"]
frames = Cn / (groups.value / S) + 32
tot = cyc[:, :, :7].sum(axis=2).astype(np.float64)    # slot 7: 100 MHz ticks of the walk (high word), not a phase
print(f"groups {groups.value}, {frames:.0f} frames each; mean cycles per wave per frame {tot.mean() / frames:.0f}")
for i, nm in enumerate(names[:7]):
    v = cyc[:, :, i].astype(np.float64)
    print(f"  {nm:26s} {100 * v.sum() / tot.sum():5.1f} %   per frame {v.mean() / frames:8.0f}")
if len(sys.argv) > 2:
    print("per-wave mean cycles per frame (rows: wave, cols: phases)")
    for w in range(waves.value):
        print(f"  wave {w:2d} " + " ".join(f"{cyc[:, w, i].astype(np.float64).mean() / frames:7.0f}" for i in range(7)))
