#!/usr/bin/env python3
"""Long live session: S streams fed one hop per call for many thousands of hops (the sample rings and the column rings wrap
hundreds of times), EXACT mode, every column's bits compared with the BATCH call on the same audio; one stream is restarted
every `every` hops and must match the batch columns of the audio it was restarted on.  Prints calls, wall time, mismatches and
the device-memory delta.  usage: live_soak.py [S] [hops] [every]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "em-spec_amd"))
import torch  # noqa: E402
import emspec  # noqa: E402
from emspec import synth  # noqa: E402


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    hops = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    every = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
    n, hop, R = 4096, 256, 1024
    D = emspec.latency_columns(n, hop, True)
    L = n + hop * (hops - 1)
    base = synth.streams(4, L)
    pcm = np.ascontiguousarray(np.stack([np.roll(base[s % 4], 977 * s) * np.float32(1.0 - 0.03 * (s % 7)) for s in range(S)]))
    dev = torch.device("cuda", 0)
    free0 = torch.cuda.mem_get_info(dev)[0]
    with emspec.Engine(mode=emspec.MODE_EXACT) as ref:
        x = torch.from_numpy(pcm).to(dev)
        db = torch.empty((S, hops, R), dtype=torch.float32, device=dev)
        ref.batch_device(x, n, hop, True, db=db)
        torch.cuda.synchronize()
        want = db.cpu().numpy().view(np.uint32)
        del db, x
    blk = emspec.PinnedArray((S, hop), np.float32)
    out = emspec.PinnedArray((S, 1, R), np.float32)
    bad = checked = restarts = 0
    who, since = S - 1, None          # the stream that restarts, and the hop its current run started at
    t0 = time.time()
    with emspec.Engine(mode=emspec.MODE_EXACT) as e:
        free1 = torch.cuda.mem_get_info(dev)[0]
        e.push_samples_multi(pcm[:, :n - hop].copy(), n, hop, True, want_db=False)
        run_ref = None
        for j in range(hops):
            if j > 0 and j % every == 0 and j + 40 < hops:
                # restart stream `who` on its own audio from sample 0: its columns must then equal the batch columns 0, 1, ...
                e.reset_stream(who)
                since, restarts = j, restarts + 1
            blk.array[:] = pcm[:, n - hop + j * hop:n + j * hop]
            if since is not None:
                a = (j - since) * hop
                blk.array[who] = pcm[who, a:a + hop]
            _, _, counts, firsts = e.push_samples_multi(blk.array, n, hop, True, db=out.array)
            got = out.array.view(np.uint32)
            for s in range(S):
                if counts[s] != 1:
                    continue
                c = int(firsts[s])
                # a restarted run's column c is complete with frames up to c + D of that run: it equals the long run's column c
                # only while the two saw the same samples, i.e. for every column (the restarted run replays the stream's own start)
                checked += 1
                if not np.array_equal(got[s, 0], want[s, c]):
                    bad += 1
            if j % 5000 == 4999:
                print(f"... {j + 1} hops, {checked} columns checked, {bad} differ, {time.time() - t0:.0f} s", flush=True)
        free2 = torch.cuda.mem_get_info(dev)[0]
    dt = time.time() - t0
    torch.cuda.empty_cache()          # (the reference run's tensors sit in torch's caching allocator until then)
    print(f"live soak: {S} streams x {hops} hops (one call per hop, EXACT, page-locked blocks), {restarts} restarts of one stream: "
          f"{checked} columns compared with the batch call's bits, {bad} differ; {dt:.0f} s wall ({dt / hops * 1e6:.0f} us per hop incl. the "
          f"Python-side compare); device memory while the session lived: {(free1 - free2) / 1e6:.1f} MB taken by it, "
          f"{(free0 - torch.cuda.mem_get_info(dev)[0]) / 1e6:.1f} MB not returned after both engines closed")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
