#!/usr/bin/env python3
"""Diagnostic: throughput of the parity dump (emspec_parity_dump_device), HIP events around back-to-back launches.
usage: tools/dump_rate.py <fft> <hop> <streams> [log2 samples]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "em-spec_amd")]
import torch, emspec
from bench import synth_device, time_launches
n, hop, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
L = 1 << (int(sys.argv[4]) if len(sys.argv) > 4 else 20)
eng = emspec.Engine(); lib = emspec.load(); dev = torch.device("cuda", 0)
pcm = synth_device(min(S, 8), L, 0, dev).repeat((S + 7) // 8, 1)[:S].contiguous()
Cn = emspec.num_columns(L, n, hop); K = n // 2 + 1
pw = torch.empty((S, Cn, K), dtype=torch.float32, device=dev)
col = torch.empty((S, Cn, K), dtype=torch.int32, device=dev)
row = torch.empty((S, Cn, K), dtype=torch.int32, device=dev)
cur = torch.cuda.current_stream(dev)
f = lib.emspec_parity_dump_device
def run():
    assert f(eng._h, pcm.data_ptr(), S, L, n, hop, 1, 0, Cn, pw.data_ptr(), col.data_ptr(), row.data_ptr(), C.c_void_p(cur.cuda_stream)) == 0
ms = time_launches(run, cur, 10)
bpc = 4 * hop + 12 * K
print(f"N={n} hop={hop} S={S}: {S*Cn/ms*1e3:.3e} columns/s ({ms:.3f} ms for {S*Cn} columns, {S*Cn*bpc/ms/1e6:.0f} GB/s = {S*Cn*bpc/ms/1e6/80:.1f} % of 8 TB/s)")
