#!/usr/bin/env python3
"""Diagnostic: throughput of frames_kernel in parity-dump mode (FFT + reassign + coalesced stores, no atomics)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "em-spec_amd")]
import torch, emspec
from bench import synth_device
n, hop, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
L = 1 << 20
eng = emspec.Engine(); lib = emspec.load(); dev = torch.device("cuda", 0)
pcm = synth_device(S, L, 0, dev)
Cn = emspec.num_columns(L, n, hop); K = n // 2 + 1
pw = torch.empty((S, Cn, K), dtype=torch.float32, device=dev)
col = torch.empty((S, Cn, K), dtype=torch.int32, device=dev)
row = torch.empty((S, Cn, K), dtype=torch.int32, device=dev)
f = lib.emspec_parity_dump_device
def run():
    assert f(eng._h, pcm.data_ptr(), S, L, n, hop, 1, 0, Cn, pw.data_ptr(), col.data_ptr(), row.data_ptr(), None) == 0
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"N={n} hop={hop}: {S*Cn/dt:.3e} columns/s ({dt*1e3:.2f} ms for {S*Cn} columns, {S*Cn*K*12/dt/1e9:.0f} GB/s of dump stores)")
