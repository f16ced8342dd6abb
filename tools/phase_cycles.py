#!/usr/bin/env python3
"""Diagnostic: per-phase shader-clock cycles of the fused kernel (stamped build, emspec_debug_phase_cycles)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "em-spec_amd")]
import numpy as np
import torch

import emspec
from bench import synth_device

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = 1 << 22
n, hop = 4096, 256
eng = emspec.Engine(diag=True)      # the stamped build lives in libemspec_diag.so
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
assert f(eng._h, pcm.data_ptr(), S, L, n, hop, 1, db.data_ptr(), idx.data_ptr(), cyc.ctypes.data, C.byref(groups), C.byref(waves)) == 0
# the product kernel (two teams half an iteration apart, fused_pp.hip.inc) stamps these phases; EMSPEC_FUSED_VARIANT=pp3 / ppt
# (round 3's form of it) the second list, r8 (the lock-step A/B variant) the third
names = ["finalize", "passes B C D", "bins", "scatter+next A", "spectrum reads", "team wait", "write+barriers", "-"]
if os.environ.get("EMSPEC_FUSED_VARIANT", "")[:3] in ("pp3", "ppt"):
    names = ["finalize", "passes B C D", "bins", "scatter", "next pass A", "team wait", "write+barrier", "-"]
if os.environ.get("EMSPEC_FUSED_VARIANT", "").startswith("r8") and os.environ.get("EMSPEC_FUSED_VARIANT") != "r8t":
    names = ["passA+write", "barrier wait", "finalize+passB", "passC r/c", "passD in place", "bins", "scatter+next A", "-"]
tot = cyc[:, :, :7].sum(axis=2).astype(np.float64)    # slot 7 is not a phase: rounds (low word) + 100 MHz ticks (high word)
print(f"groups {groups.value}; mean cycles per wave {tot.mean():.0f} (readcyclecounter units)")
for i, nm in enumerate(names[:7]):
    v = cyc[:, :, i].astype(np.float64)
    print(f"  {nm:16s} {100 * v.sum() / tot.sum():5.1f} %   per-iteration {v.mean() / ((Cn / (groups.value / S) + 16) / 2):8.0f}")
it_ = (Cn / (groups.value / S) + 16) / 2
rounds = (cyc[:, :, 7] & np.uint64(0xFFFFFFFF)).astype(np.float64).mean() / it_
if rounds > 0:      # only the variants that still count them (the default kernel's accumulate loop is assembly now)
    print(f"accumulate rounds per wave per frame (4 calls): {rounds:.2f}")
if len(sys.argv) > 2:
    print("per-wave mean cycles per iteration (rows: wave, cols: phases)")
    it = (Cn / (groups.value / S) + 16) / 2
    for w in range(waves.value):
        print(f"  wave {w:2d} " + " ".join(f"{cyc[:, w, i].astype(np.float64).mean() / it:7.0f}" for i in range(7)))
