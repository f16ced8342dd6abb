#!/usr/bin/env python3
"""columns/s of the device-resident batch entry over every supported FFT size (hop = N/16, reassign on,
64 streams x 2^21 samples), with the work-normalised figure columns/s x N log2 N for comparison."""
import os, sys, time, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "em-spec_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import emspec
from bench import synth_device
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
S, L = 64, 1 << 21
eng = emspec.Engine(device=0)
pcm = synth_device(S, L, 0, dev)
print(f"{'N':>6} {'hop':>5} {'path':>8} {'columns':>9} {'ms':>8} {'columns/s':>11} {'x N log2 N':>11}")
for n in (256, 512, 1024, 2048, 4096, 8192, 16384):
    for hop in (n // 16, n // 8):
        C = emspec.num_columns(L, n, hop)
        db = torch.empty((S, C, eng.rows), dtype=torch.float32, device=dev)
        idx = torch.empty((S, C, eng.rows), dtype=torch.uint8, device=dev)
        for _ in range(2):
            eng.batch_device(pcm, n, hop, True, db=db, index=idx)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            eng.batch_device(pcm, n, hop, True, db=db, index=idx)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / reps
        rate = S * C / dt
        print(f"{n:6d} {hop:5d} {'fused' if eng.fused(n, hop, True) else 'generic':>8} {S * C:9d} {dt * 1e3:8.2f} {rate:11.3e} "
              f"{rate * n * math.log2(n):11.3e}", flush=True)
        del db, idx
