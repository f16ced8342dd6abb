#!/usr/bin/env python3
"""Measure every single-GPU BASELINE.json config (device-resident batch entry) and print a markdown table."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "em-spec_amd")]
import torch, emspec
from bench import synth_device

dev = torch.device("cuda", 0)
eng = emspec.Engine()
rows = []
for name, S, n, hop, re, log2L in [
        ("configs[0] shape: 1 stream, FFT 1024, hop 256, reassign OFF", 1, 1024, 256, False, 20),
        ("configs[1]: 1 stream, FFT 4096, hop 256, reassign ON", 1, 4096, 256, True, 22),
        ("configs[2]: 64 streams, FFT 4096, hop 256, reassign ON", 64, 4096, 256, True, 22),
        ("configs[4]: 64 streams, FFT 16384, hop 512, reassign ON", 64, 16384, 512, True, 22),
        ("64 streams, FFT 1024, hop 256, reassign ON", 64, 1024, 256, True, 20),
        ("64 streams, FFT 4096, hop 256, reassign OFF", 64, 4096, 256, False, 22)]:
    L = 1 << log2L
    pcm = synth_device(min(S, 4), L, 0, dev)
    pcm = pcm.repeat((S + 3) // 4, 1)[:S].contiguous()
    C = emspec.num_columns(L, n, hop)
    db = torch.empty((S, C, eng.rows), dtype=torch.float32, device=dev)
    idx = torch.empty((S, C, eng.rows), dtype=torch.uint8, device=dev)
    for _ in range(2):
        eng.batch_device(pcm, n, hop, re, db=db, index=idx)
    torch.cuda.synchronize()
    reps = 5 if S > 1 else 50
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.batch_device(pcm, n, hop, re, db=db, index=idx)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    rows.append((name, S * C, dt * 1e3, S * C / dt, "fused" if eng.fused(n, hop, re) else "records + scatter"))
    del db, idx, pcm
print("| workload | columns per launch | ms | columns/s | path |")
print("|---|---|---|---|---|")
for r in rows:
    print(f"| {r[0]} | {r[1]:,} | {r[2]:.2f} | {r[3]:.3g} | {r[4]} |")
