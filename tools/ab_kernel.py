#!/usr/bin/env python3
"""A/B two builds of libemspec.so on the same box: alternates them (child process each), times the column kernel of a
bench workload with HIP events, prints ms per launch.  Box-to-box spread is ~2 %, so variants are compared here.
   python tools/ab_kernel.py libA.so libB.so [--workload batch64|n16384] [--rounds 3]
A library given as path@VARIANT runs with EMSPEC_FUSED_VARIANT=VARIANT, path@NAME=value with that environment variable
(libemspec_diag.so only).
"""
import argparse, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "em-spec_amd")]
import torch
import emspec
emspec.LIB_PATH = %(lib)r
from bench import synth_device, time_launches
n, hop = %(n)d, %(hop)d
dev = torch.device("cuda", 0)
eng = emspec.Engine(mode=emspec.MODE_EXACT if %(exact)d else emspec.MODE_FAST)
S, L = (%(streams)d or (16 if %(exact)d else 64)), 1 << 22
C = emspec.num_columns(L, n, hop)
pcm = synth_device(S, L, 0, dev)
db = torch.empty((S, C, eng.rows), dtype=torch.float32, device=dev)
idx = torch.empty((S, C, eng.rows), dtype=torch.uint8, device=dev)
cur = torch.cuda.current_stream(dev)
if %(dump)d:
    import ctypes as C_
    lib = emspec.load()
    Sd, Ld = 16, 1 << 20
    Cd, K = emspec.num_columns(Ld, n, hop), n // 2 + 1
    sub = pcm[:Sd, :Ld].contiguous()
    pw_ = torch.empty((Sd, Cd, K), dtype=torch.float32, device=dev)
    cl_ = torch.empty((Sd, Cd, K), dtype=torch.int32, device=dev)
    rw_ = torch.empty((Sd, Cd, K), dtype=torch.int32, device=dev)
    def dump():
        assert lib.emspec_parity_dump_device(eng._h, sub.data_ptr(), Sd, Ld, n, hop, 1, 0, Cd, pw_.data_ptr(), cl_.data_ptr(),
                                             rw_.data_ptr(), C_.c_void_p(cur.cuda_stream)) == 0
    eng.batch_device(pcm, n, hop, True, db=db, index=idx, stream=cur)     # warm the clock
    ms = time_launches(dump, cur, 20)
    idx.zero_(); db.zero_()
else:
    ms = time_launches(lambda: eng.batch_device(pcm, n, hop, True, db=db, index=idx, stream=cur), cur, %(reps)d)
torch.cuda.synchronize()
print(f"checksum idx {int(idx.sum(dtype=torch.int64).item())} db {float(db.double().sum().item()):.6e}", file=sys.stderr)
print(f"{ms:.4f}")
'''

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--workload", default="batch64")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--n", type=int, default=0, help="FFT size (overrides --workload)")
ap.add_argument("--hop", type=int, default=0)
ap.add_argument("--exact", action="store_true", help="EXACT mode engine, 16 streams")
ap.add_argument("--streams", type=int, default=0, help="streams (default 64; 16 with --exact)")
ap.add_argument("--dump", action="store_true", help="time the per-bin parity dump (16 streams x 2^20 samples) instead of the batch")
a = ap.parse_args()
n, hop, reps = (16384, 512, 4) if a.workload == "n16384" else (4096, 256, 8)
if a.n:
    n, hop, reps = a.n, a.hop or n // 4, 6
res = {l: [] for l in a.libs}
for r in range(a.rounds):
    for lib in a.libs:
        path, _, variant = lib.partition("@")
        env = None
        if variant:
            env = dict(os.environ)
            for item in variant.split(","):      # path@NAME=value[,NAME2=value2]
                name, eq, val = item.partition("=")
                env.update({name: val} if eq else {"EMSPEC_FUSED_VARIANT": item})
        out = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, lib=os.path.abspath(path), n=n, hop=hop, reps=reps, dump=int(a.dump), exact=int(a.exact), streams=a.streams)],
                             capture_output=True, text=True, timeout=300, env=env)
        if out.returncode != 0:
            sys.exit(out.stderr[-2000:])
        res[lib].append(float(out.stdout.strip().splitlines()[-1]))
        chk = [l for l in out.stderr.splitlines() if l.startswith("checksum")]
        if r == 0 and chk:
            print(f"  {os.path.basename(lib)}: {chk[-1]}")
        print(f"round {r} {os.path.basename(lib)}: {res[lib][-1]:.3f} ms", flush=True)
for lib in a.libs:
    v = res[lib]
    print(f"{os.path.basename(lib):32s} min {min(v):.3f}  mean {sum(v) / len(v):.3f} ms")
