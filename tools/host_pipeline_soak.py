#!/usr/bin/env python3
"""Soak of the host-buffer pipeline: alternating emspec_batch (index out) from pinned and from ordinary memory (its helper
threads), one long stream cut into runs of columns, and emspec_batch_packed calls, both modes, checking that every call returns the same bytes as the first one and that device memory does not grow.
   python tools/host_pipeline_soak.py [iterations]         (needs an MI355X)"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "em-spec_amd")]
import numpy as np
import torch

import emspec
from emspec import synth

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
S, L, n, hop = 24, 1 << 20, 4096, 256
Cn = emspec.num_columns(L, n, hop)
pcm = np.repeat(synth.streams(4, L), 6, axis=0) * np.linspace(0.5, 1.0, 24, dtype=np.float32)[:, None]
pin = emspec.PinnedArray((S, L), np.float32)
pin.array[...] = pcm
pix = emspec.PinnedArray((S, Cn, 1024), np.uint8)
pw = emspec.PinnedArray((S * emspec.wire_bound(Cn, 1024),), np.uint8)
lib = emspec.load()
for mode, name in ((emspec.MODE_EXACT, "exact"), (emspec.MODE_FAST, "fast")):
    with emspec.Engine(mode=mode) as e:
        o = emspec.Out(None, None, C.c_void_p(pix.array.ctypes.data))
        offs = np.zeros(S + 1, np.int64)
        ref_idx = ref_wire = ref_long = None
        pg_idx = np.empty((S, Cn, 1024), np.uint8)
        long_pcm = np.ascontiguousarray(pcm[:12].reshape(-1))            # 12 x 2^20 samples as one stream: 49,137 columns = 2 runs
        long_idx = np.empty((emspec.num_columns(long_pcm.size, n, hop), 1024), np.uint8)
        free0 = None
        t0 = time.time()
        worst = 0
        for it in range(iters):
            assert lib.emspec_batch(e._h, C.c_void_p(pin.array.ctypes.data), S, L, n, hop, 1, C.byref(o)) == 0
            if ref_idx is None:
                ref_idx = pix.array.copy()
            elif mode == emspec.MODE_EXACT:
                assert np.array_equal(pix.array, ref_idx), f"{name}: index bytes changed at iteration {it}"
            else:
                d = np.abs(pix.array.astype(np.int16) - ref_idx.astype(np.int16))
                worst = max(worst, int(d.max()))
                assert d.max() <= 1 and np.mean(d != 0) < 1e-4
            # the same batch from ORDINARY memory (the second host thread + the page-touching threads; every third time into a fresh
            # array) and, every eighth iteration, the four streams' samples as ONE long stream cut into runs of columns
            pgi = np.empty((S, Cn, 1024), np.uint8) if it % 3 == 0 else pg_idx
            po = emspec.Out(None, None, C.c_void_p(pgi.ctypes.data))
            assert lib.emspec_batch(e._h, C.c_void_p(pcm.ctypes.data), S, L, n, hop, 1, C.byref(po)) == 0
            if mode == emspec.MODE_EXACT:
                assert np.array_equal(pgi, ref_idx), f"{name}: pageable index bytes differ at iteration {it}"
            else:
                d = np.abs(pgi.astype(np.int16) - ref_idx.astype(np.int16))
                assert d.max() <= 1 and np.mean(d != 0) < 1e-4
            if it % 8 == 0:
                lo = emspec.Out(None, None, C.c_void_p(long_idx.ctypes.data))
                assert lib.emspec_batch(e._h, C.c_void_p(long_pcm.ctypes.data), 1, long_pcm.size, n, hop, 1, C.byref(lo)) == 0
                if ref_long is None:
                    ref_long = long_idx.copy()
                    x = torch.from_numpy(long_pcm[None]).cuda()
                    dix = torch.empty((1, long_idx.shape[0], 1024), dtype=torch.uint8, device="cuda")
                    e.batch_device(x, n, hop, True, index=dix)
                    torch.cuda.synchronize()
                    dd = np.abs(dix.cpu().numpy()[0].astype(np.int16) - ref_long.astype(np.int16))
                    assert (dd.max() == 0) if mode == emspec.MODE_EXACT else (dd.max() <= 1 and np.mean(dd != 0) < 1e-4), "long stream: runs differ from one piece"
                    del x, dix
                elif mode == emspec.MODE_EXACT:
                    assert np.array_equal(long_idx, ref_long), f"long stream: bytes changed at iteration {it}"
            assert lib.emspec_batch_packed(e._h, C.c_void_p(pin.array.ctypes.data), S, L, n, hop, 1, C.c_void_p(pw.array.ctypes.data),
                                           C.c_int64(pw.array.size), offs.ctypes.data_as(C.c_void_p)) == 0
            if mode == emspec.MODE_EXACT:
                img = pw.array[:offs[-1]].copy()
                if ref_wire is None:
                    ref_wire = img
                    back = emspec.wire_unpack_host(pw.array[offs[5]:offs[6]], Cn, 1024)
                    assert np.array_equal(back, ref_idx[5])
                else:
                    assert np.array_equal(img, ref_wire), f"packed bytes changed at iteration {it}"
            if it == 2:
                free0 = torch.cuda.mem_get_info()[0]
        free1 = torch.cuda.mem_get_info()[0]
        e.device_status()
        print(f"{name}: {iters} x (emspec_batch index out, pinned and pageable, + emspec_batch_packed; every eighth: one long stream in runs), {S} streams x 2^20 samples: {time.time() - t0:.1f} s, "
              f"every call the same bytes{'' if mode == emspec.MODE_EXACT else f' (float32 mode: +-{worst} cells)'}, device memory delta "
              f"{(free0 - free1) / 1e6:.1f} MB, no device error word", flush=True)
pin.close(); pix.close(); pw.close()
