#!/usr/bin/env python3
"""Standalone cost of the gather's wire kernels on real finished columns (no concurrent compute): pack and expand of
one stream-chunk of the bench workload (32 streams x 16369 columns), HIP events.  usage: tools/gather_cost.py [streams]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "em-spec_amd")]
import torch

import emspec
from bench import synth_device, time_launches

S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n, hop, L = 4096, 256, 1 << 22
dev = torch.device("cuda", 0)
eng = emspec.Engine()
pcm = synth_device(min(S, 8), L, 0, dev).repeat((S + 7) // 8, 1)[:S].contiguous()
C = emspec.num_columns(L, n, hop)
idx = torch.empty((S, C, eng.rows), dtype=torch.uint8, device=dev)
eng.batch_device(pcm, n, hop, True, index=idx)
cols = S * C
wire = torch.empty(emspec.wire_bound(cols, eng.rows), dtype=torch.uint8, device=dev)
back = torch.empty_like(idx)
cur = torch.cuda.current_stream(dev)
nbytes = eng.wire_pack(idx, wire)
t_pack = time_launches(lambda: eng.wire_pack(idx, wire, want_size=False), cur, 10)
t_unpack = time_launches(lambda: eng.wire_unpack(wire, nbytes, back), cur, 10)   # includes a 32-byte header read-back
assert torch.equal(back, idx)
raw = cols * eng.rows
print(f"{cols} columns x {eng.rows} rows: wire image {nbytes / cols:.1f} B/column ({raw / nbytes:.2f}x smaller)")
print(f"pack   {t_pack:.3f} ms  = {raw / t_pack / 1e6:.0f} GB/s of raw columns read")
print(f"expand {t_unpack:.3f} ms  = {raw / t_unpack / 1e6:.0f} GB/s of raw columns written")
