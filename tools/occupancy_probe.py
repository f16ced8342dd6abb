#!/usr/bin/env python3
"""How much do the column kernels slow down when another kernel (a stand-in for an RCCL send/recv
kernel) holds a few compute units?  The fused workgroup fills a CU (16 waves x 128 VGPRs, 145 KB LDS),
so a resident communication kernel removes whole CUs from the pool and the batch launch quantises into
more rounds.  Measures ms per step of the bench workload (64 streams) launched as `chunks` stream-chunks
while `thieves` 256-thread workgroups stay resident on a second stream.  One GPU; no collective."""
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
side = torch.cuda.Stream(device=dev)
steps = 6
print(f"{'chunks':>6} {'thieves':>7} {'ms/step':>8}")
for thieves in (0, 8, 16, 32):
    for nch in (1, 2, 4, 8):
        bounds = [(S * i // nch, S * (i + 1) // nch) for i in range(nch)]
        def run(k):
            for _ in range(k):
                for a, b in bounds:
                    eng.batch_device(pcm[a:b], n, hop, True, db=db[a:b], index=idx[a:b], stream=cur)
        run(2)
        torch.cuda.synchronize(dev)
        if thieves:   # resident for the whole timed region (about steps x 30 ms at worst)
            assert lib.emspec_debug_occupy(eng._h, thieves, steps * 30000, C.c_void_p(side.cuda_stream)) == 0
            time.sleep(0.002)
        t0 = time.perf_counter()
        run(steps)
        cur.synchronize()
        dt = (time.perf_counter() - t0) / steps
        torch.cuda.synchronize(dev)
        print(f"{nch:6d} {thieves:7d} {dt * 1e3:8.2f}", flush=True)
