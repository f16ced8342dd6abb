#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-buffer entry points at the bench shape (64 streams x 2^22 samples, N = 4096, hop 256):
emspec_batch (dB out / palette index out, both modes) and emspec_batch_packed, from page-locked buffers, beside the
measured hipMemcpy rates of the same buffers - the roofline of a path whose every byte crosses PCIe.
   python tools/host_pipeline_rates.py [streams]            (needs an MI355X)"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "em-spec_amd")]
import numpy as np
import torch

import emspec
from bench import synth_device

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L, n, hop = 1 << 22, 4096, 256
dev = torch.device("cuda", 0)
Cn = emspec.num_columns(L, n, hop)
res = {"streams": S, "columns": S * Cn}
pin = emspec.PinnedArray((S, L), np.float32)
pin.array[...] = synth_device(S, L, 0, dev).cpu().numpy()
pix = emspec.PinnedArray((S, Cn, 1024), np.uint8)
DIAG = bool(os.environ.get("EMSPEC_PIPE_CHUNKS"))      # the chunk-count switch lives in the diagnostic build
lib = emspec.load(diag=DIAG)
hip = C.CDLL("libamdhip64.so")


def copy_rate(nbytes, h2d):
    d = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    host = pin.array.ctypes.data if h2d else pix.array.ctypes.data
    best = 0.0
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if h2d:
            rc = hip.hipMemcpy(C.c_void_p(d.data_ptr()), C.c_void_p(host), C.c_size_t(nbytes), 1)
        else:
            rc = hip.hipMemcpy(C.c_void_p(host), C.c_void_p(d.data_ptr()), C.c_size_t(nbytes), 2)
        assert rc == 0
        best = max(best, nbytes / (time.perf_counter() - t0) / 1e9)
    return best


nb = min(pin.array.nbytes, pix.array.nbytes)
res["h2d_GBps"] = copy_rate(nb, True)
res["d2h_GBps"] = copy_rate(nb, False)
print(f"hipMemcpy, pinned, {nb / 1e6:.0f} MB: H2D {res['h2d_GBps']:.1f} GB/s, D2H {res['d2h_GBps']:.1f} GB/s", flush=True)



def timed(fn, reps=3):
    fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return float(np.median(t))


for mode, name in ((emspec.MODE_FAST, "fast"), (emspec.MODE_EXACT, "exact")):
    with emspec.Engine(mode=mode, diag=DIAG) as e:
        o = emspec.Out(None, None, C.c_void_p(pix.array.ctypes.data))

        def run_idx():
            assert lib.emspec_batch(e._h, C.c_void_p(pin.array.ctypes.data), S, L, n, hop, 1, C.byref(o)) == 0
        dt = timed(run_idx)
        res[f"{name}_index_out_columns_per_s"] = S * Cn / dt
        gb = (pin.array.nbytes + pix.array.nbytes) / 1e9
        print(f"emspec_batch {name:5s} pinned, uint8 palette index out: {S * Cn / dt:.3e} columns/s ({dt * 1e3:.1f} ms; {pin.array.nbytes / dt / 1e9:.1f} GB/s in + "
              f"{pix.array.nbytes / dt / 1e9:.1f} GB/s out)", flush=True)
        wire = pix.array.reshape(-1)              # reuse the pinned output buffer for the images
        offs = np.zeros(S + 1, np.int64)

        def run_packed():
            assert lib.emspec_batch_packed(e._h, C.c_void_p(pin.array.ctypes.data), S, L, n, hop, 1, C.c_void_p(wire.ctypes.data),
                                           C.c_int64(wire.size), offs.ctypes.data_as(C.c_void_p)) == 0, e._lib.emspec_last_error(e._h)
        dt = timed(run_packed)
        res[f"{name}_packed_columns_per_s"] = S * Cn / dt
        res[f"{name}_packed_bytes_per_column"] = float(offs[-1]) / (S * Cn)
        print(f"emspec_batch_packed {name:5s} pinned: {S * Cn / dt:.3e} columns/s ({dt * 1e3:.1f} ms; {pin.array.nbytes / dt / 1e9:.1f} GB/s in, "
              f"{offs[-1] / (S * Cn):.0f} B per column out)", flush=True)
        if mode == emspec.MODE_FAST:
            t0 = time.perf_counter()
            one = emspec.wire_unpack_host(wire[offs[0]:offs[1]], Cn, 1024)
            du = time.perf_counter() - t0
            res["host_unpack_columns_per_s_one_core"] = Cn / du
            print(f"emspec_wire_unpack_host: {Cn / du:.3e} columns/s on one core", flush=True)
    if mode == emspec.MODE_FAST and S <= 64:
        with emspec.Engine(mode=mode, diag=DIAG) as e:
            pdb = emspec.PinnedArray((min(S, 16), Cn, 1024), np.float32)
            Sd = min(S, 16)
            o = emspec.Out(C.c_void_p(pdb.array.ctypes.data), None, None)

            def run_db():
                assert lib.emspec_batch(e._h, C.c_void_p(pin.array.ctypes.data), Sd, L, n, hop, 1, C.byref(o)) == 0
            dt = timed(run_db)
            res["fast_db_out_columns_per_s"] = Sd * Cn / dt
            print(f"emspec_batch fast  pinned, float32 dB out ({Sd} streams): {Sd * Cn / dt:.3e} columns/s ({dt * 1e3:.1f} ms; {pdb.array.nbytes / dt / 1e9:.1f} GB/s out)", flush=True)
            pdb.close()
# both directions at once (two streams): the ceiling of a pipeline whose copies overlap
da, db_ = torch.empty(nb, dtype=torch.uint8, device=dev), torch.empty(nb, dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
best = 0.0
for _ in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    assert hip.hipMemcpyAsync(C.c_void_p(da.data_ptr()), C.c_void_p(pin.array.ctypes.data), C.c_size_t(nb), 1, C.c_void_p(s1.cuda_stream)) == 0
    assert hip.hipMemcpyAsync(C.c_void_p(pix.array.ctypes.data), C.c_void_p(db_.data_ptr()), C.c_size_t(nb), 2, C.c_void_p(s2.cuda_stream)) == 0
    torch.cuda.synchronize()
    best = max(best, nb / (time.perf_counter() - t0) / 1e9)
res["duplex_GBps"] = best
print(f"hipMemcpyAsync both ways at once, {nb / 1e6:.0f} MB each: {best:.1f} GB/s per direction", flush=True)
del da, db_
print(json.dumps(res))
