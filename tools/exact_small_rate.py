#!/usr/bin/env python3
"""EXACT mode at the small sizes: columns/s of emspec_batch_device on 64 streams, the one-kernel path against the two-kernel
records path (EMSPEC_EXACT_RECORDS=1, diagnostic library; the switch is read once per process, so the tool re-runs itself).
usage: exact_small_rate.py [records]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "em-spec_amd"))


def run():
    import torch
    import emspec
    from emspec import synth
    import numpy as np
    S, R = 64, 1024
    dev = torch.device("cuda", 0)
    for n, hop, L in ((1024, 256, 1 << 20), (1024, 128, 1 << 19), (2048, 256, 1 << 20), (2048, 128, 1 << 19), (4096, 256, 1 << 20)):
        base = synth.streams(8, L)
        pcm = torch.from_numpy(np.ascontiguousarray(np.tile(base, (8, 1)))).to(dev)
        C = emspec.num_columns(L, n, hop)
        idx = torch.empty((S, C, R), dtype=torch.uint8, device=dev)
        db = torch.empty((S, C, R), dtype=torch.float32, device=dev)
        with emspec.Engine(mode=emspec.MODE_EXACT, diag=True) as e:
            fused = e.fused(n, hop, True)
            for _ in range(2):
                e.batch_device(pcm, n, hop, True, db=db, index=idx)
            torch.cuda.synchronize()
            t = []
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                e.batch_device(pcm, n, hop, True, db=db, index=idx)
                b.record()
                torch.cuda.synchronize()
                t.append(a.elapsed_time(b))
            ms = float(np.median(t))
            e.device_status()
        print(f"N={n:5d} hop={hop:4d} fused={int(fused)}  {ms:8.3f} ms  {S * C / ms * 1e3:.3e} columns/s  "
              f"(algorithmic {(4 * hop + 5 * R) * S * C / ms / 1e6:.0f} GB/s)", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        run()
    else:
        for rec in ("0", "1"):
            print(f"--- EMSPEC_EXACT_RECORDS={rec}", flush=True)
            env = dict(os.environ, EMSPEC_EXACT_RECORDS=rec)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)
