#!/usr/bin/env python3
"""When does a small kernel launched on a second stream WHILE the column kernel is running get its
compute units?  (The gather of chunk k is enqueued while chunk k+1 computes.)  Prints, for a normal- and
a high-priority side stream, when a 16-workgroup / 1 ms stand-in kernel enqueued ~1 ms after the column
kernel finishes, next to the column kernel's own duration.  One GPU; no collective."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "em-spec_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import emspec
from bench import synth_device

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
S, L, n, hop = 64, 1 << 22, 4096, 256
eng = emspec.Engine(device=0, diag=True)
lib = emspec.load(diag=True)
lib.emspec_debug_occupy.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
R, Cn = eng.rows, emspec.num_columns(L, n, hop)
pcm = synth_device(S, L, 0, dev)
db = torch.empty((S, Cn, R), dtype=torch.float32, device=dev)
idx = torch.empty((S, Cn, R), dtype=torch.uint8, device=dev)
cur = torch.cuda.current_stream(dev)
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
for name, prio in (("normal", 0), ("high", -1)):
    side = torch.cuda.Stream(device=dev, priority=prio)
    for rep in range(3):
        torch.cuda.synchronize(dev)
        e0, e1, s0, s1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        e0.record(cur)
        eng.batch_device(pcm, n, hop, True, db=db, index=idx, stream=cur)
        e1.record(cur)
        time.sleep(0.001)
        s0.record(side)
        assert lib.emspec_debug_occupy(eng._h, 16, 1000, C.c_void_p(side.cuda_stream)) == 0
        s1.record(side)
        torch.cuda.synchronize(dev)
        print(f"{name:6s} side stream: column kernel {e0.elapsed_time(e1):6.2f} ms; stand-in enqueued ~1 ms in, "
              f"done at {e0.elapsed_time(s1):6.2f} ms (ran {s0.elapsed_time(s1):5.2f} ms after reaching the queue head)", flush=True)
