#!/usr/bin/env python3
"""bench.py — reassigned spectrogram columns/s (4096-pt, hop 256, 48 kHz) on N MI355X.

A "step" is one pass of the hot path over one batch of synthetic audio that is
already resident in HBM: every rank turns its 64 streams x 2^22 samples
(BASELINE.json configs[2]; configs[3] at N=8) into 64 x 16,369 finished columns
(float32 dB + uint8 palette index per cell).  For N > 1 the streams shard
across ranks with no data-path exchange; the only collective is the RCCL
gather of the finished palette-index columns to rank 0 (north_star), issued
per stream-chunk on a side stream so it overlaps the next chunk's compute.

Prints ONE JSON line on rank 0 (contract: see the task statement / DESIGN.md §6).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "em-spec_amd"), os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def _lsr(z, k):
    """logical right shift of int64 tensors (torch's >> is arithmetic)"""
    return (z >> k) & ((1 << (64 - k)) - 1)


def _splitmix64(seed, idx):
    """draws number idx (int64 tensor, 1-based) of splitmix64 seeded with `seed`: int64 arithmetic wraps mod 2^64"""
    def s64(c):
        return c - (1 << 64) if c >= (1 << 63) else c
    z = idx * s64(0x9E3779B97F4A7C15) + s64(seed & ((1 << 64) - 1))
    z = (z ^ _lsr(z, 30)) * s64(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(z, 27)) * s64(0x94D049BB133111EB)
    return z ^ _lsr(z, 31)


def _uniform(seed, idx):
    return _lsr(_splitmix64(seed, idx), 11).double() * (1.0 / (1 << 53))


def synth_device(S, L, first_stream, device, fs=48000.0):
    """The synthetic audio of SURVEY.md §8(d), generated on the device: the same counter-based definition as
    em-spec_amd/emspec/synth.py and em-spec_amd/js/synth.js (stream s: seed 1000+s; splitmix64 uniforms; 8
    sinusoids + chirp + Gaussian noise at -60 dBFS + a click every 24000 samples), float32 in [-1,1]."""
    out = torch.empty((S, L), dtype=torch.float32, device=device)
    t = torch.arange(L, dtype=torch.float64, device=device) / fs
    half = (L + 1) // 2
    pair = torch.arange(half, dtype=torch.int64, device=device)
    for s in range(S):
        seed = 1000 + first_stream + s
        u = _uniform(seed, torch.arange(1, 65, dtype=torch.int64, device=device)).cpu().numpy()
        x = torch.zeros(L, dtype=torch.float64, device=device)
        for i in range(8):
            f = 30.0 * (20000.0 / 30.0) ** float(u[i])
            a = 10.0 ** (-40.0 * float(u[8 + i]) / 20.0)
            x += a * torch.sin(2 * np.pi * f * t + 2 * np.pi * float(u[16 + i]))
        f0 = 200.0 + 4000.0 * float(u[24])
        rate = 4.0e4 * (0.25 + 0.75 * float(u[25]))
        tt = torch.remainder(t, max(1e-3, min(L / fs, (20000.0 - f0) / rate)))
        x += 0.25 * torch.sin(2 * np.pi * (f0 * tt + 0.5 * rate * tt * tt))
        u1 = torch.clamp(_uniform(seed ^ 0x5EED, 64 + 2 * pair + 1), min=1e-300)     # Box-Muller on draws 65, 66, ...
        u2 = _uniform(seed ^ 0x5EED, 64 + 2 * pair + 2)
        r = torch.sqrt(-2.0 * torch.log(u1))
        noise = torch.stack([r * torch.cos(2 * np.pi * u2), r * torch.sin(2 * np.pi * u2)], dim=1).reshape(-1)
        x += 1e-3 * noise[:L]
        x[::24000] += 1.0
        x /= max(1.0, float(x.abs().max()))
        out[s] = x.float()
    return out


def cpu_baseline(n, hop, seconds_target=12.0):
    """Time the CPU oracle (float32 bit model, oracle/emspec_oracle.c) on a bounded sample of the
    same workload, all host cores (OpenMP over streams)."""
    import oracle as O
    from emspec import synth
    cores = O.max_threads()
    cfg = O.make_cfg(n, hop, True)
    # calibrate on a small run, then size the sample for ~seconds_target
    probe = synth.streams(cores, n + hop * 127)
    t0 = time.perf_counter(); O.batch_f32(cfg, probe, want=("db", "index"), threads=cores); dt = time.perf_counter() - t0
    rate = cores * 128 / dt
    cols_per_stream = int(max(256, min(8192, rate * seconds_target / cores)))
    L = n + hop * (cols_per_stream - 1)
    base = synth.streams(1, L)
    pcm = np.stack([np.roll(base[0], 977 * s) for s in range(cores)])
    t0 = time.perf_counter(); O.batch_f32(cfg, pcm, want=("db", "index"), threads=cores); dt = time.perf_counter() - t0
    return {"value": cores * cols_per_stream / dt, "unit": "columns/s", "cores": cores, "kind": "port",
            "sample": f"{cores} streams x {cols_per_stream} columns (N={n}, hop={hop}, reassign on), "
                      f"oracle float32 bit model, OpenMP {cores} threads, {dt:.1f} s"}


def js_baseline(n, hop, seconds=5.0):
    """Plain-JS restatement under node (oracle/js/reassign_ref.js), one thread: the closest proxy for the
    reference's JS/WebAudio CPU path that can exist here (CPU baseline plan item 3)."""
    import shutil
    import subprocess
    node = shutil.which("node")
    if node is None:
        return None
    try:
        out = subprocess.run([node, os.path.join(ROOT, "oracle", "js", "reassign_ref.js"), "bench", str(n), str(hop),
                              str(seconds)], capture_output=True, text=True, timeout=120).stdout
        d = json.loads(out.strip().splitlines()[-1])
        return {"value": d["columns_per_s"], "unit": "columns/s", "cores": 1, "kind": "port",
                "sample": f"{d['columns']} columns of one stream (N={n}, hop={hop}, reassign on), plain JavaScript "
                          f"restatement (three windowed FFTs, float64) under node {d['node']}, {d['seconds']:.1f} s"}
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="batch64", choices=["batch64", "single", "n16384", "n8192", "n1024"])
    ap.add_argument("--streams", type=int, default=0, help="override streams per GPU")
    ap.add_argument("--log2-samples", type=int, default=22)
    ap.add_argument("--chunks", type=int, default=2, help="stream-chunks per step (gather overlap, N>1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-chunks", type=int, default=0, help="N=1: launch per stream-chunk as the N>1 path does (no gather)")
    ap.add_argument("--reassign", type=int, default=1)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N>1 control flow on fewer GPUs than ranks (gather via host tensors)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    if world > 1:
        torch.cuda.set_device(dev_index)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo")
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)

    import emspec
    from emspec import shard
    if args.workload == "n16384":
        n, hop = 16384, 512
    elif args.workload == "n8192":
        n, hop = 8192, 512
    elif args.workload == "n1024":
        n, hop = 1024, 256
    else:
        n, hop = 4096, 256
    S = args.streams or (1 if args.workload == "single" else 64)
    L = 1 << args.log2_samples
    eng = emspec.Engine(device=dev_index)
    R = eng.rows
    C = emspec.num_columns(L, n, hop)
    first_stream, _ = shard.stream_shard(rank, world, world * S)
    pcm = synth_device(S, L, first_stream, dev)
    db = torch.empty((S, C, R), dtype=torch.float32, device=dev)
    # N>1: two index buffers, so the gather of step k's last chunks overlaps step k+1's first kernels
    nbuf = 2 if world > 1 else 1
    idx_bufs = [torch.empty((S, C, R), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    idx = idx_bufs[0]

    nch = max(1, min(args.chunks, S)) if world > 1 else max(1, min(args.force_chunks, S)) if args.force_chunks else 1
    bounds = [(S * i // nch, S * (i + 1) // nch) for i in range(nch)]
    comm_stream = torch.cuda.Stream(device=dev) if world > 1 else None
    gathered = None
    gdev = dev if args.backend == "nccl" else torch.device("cpu")
    if world > 1 and rank == 0:
        gathered = [[torch.empty((b - a, C, R), dtype=torch.uint8, device=gdev) for _ in range(world)] for a, b in bounds]

    cur = torch.cuda.current_stream(dev)
    kev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    sent = [[None] * nch for _ in range(nbuf)]     # per (buffer, chunk): event after its last gather
    nstep = [0]

    def step(timed_i=None):
        p = nstep[0] % nbuf
        nstep[0] += 1
        ibuf = idx_bufs[p]
        for ci, (a, b) in enumerate(bounds):
            if sent[p][ci] is not None:
                cur.wait_event(sent[p][ci])        # the gather that last read this chunk of ibuf
            if timed_i is not None and ci == 0:
                kev[timed_i][0].record(cur)
            eng.batch_device(pcm[a:b], n, hop, bool(args.reassign), db=db[a:b], index=ibuf[a:b], stream=cur)
            if timed_i is not None and ci == nch - 1:
                kev[timed_i][1].record(cur)
            if world > 1:
                ready = torch.cuda.Event()
                ready.record(cur)
                with torch.cuda.stream(comm_stream):
                    comm_stream.wait_event(ready)
                    src = ibuf[a:b] if args.backend == "nccl" else ibuf[a:b].cpu()   # gloo rehearsal: host tensors
                    shard.gather_columns_into(src, gathered[ci] if rank == 0 else None, dst=0)
                    sent[p][ci] = torch.cuda.Event()
                    sent[p][ci].record(comm_stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if world > 1:
        # open the point-to-point connections the gather uses before anything is timed (with --warmup 0 the
        # first gather would otherwise pay for them inside the timed region)
        probe = torch.zeros(16, dtype=torch.uint8, device=gdev)
        shard.gather_columns_into(probe, [torch.empty_like(probe) for _ in range(world)] if rank == 0 else None, dst=0)
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=gdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        cols_per_step = world * S * C
        value = cols_per_step * args.steps / elapsed
        # dominant-kernel roofline: algorithmic bytes per column for this output mode
        # (4*hop in, + 4*R dB out, + R palette-index out)
        bytes_per_col = 4 * hop + 4 * R + R
        kms = [a.elapsed_time(b) for a, b in kev]           # HIP events on the launch stream
        k_avg_ms = float(np.mean(kms))
        achieved = S * C * bytes_per_col / (k_avg_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        if args.workload == "batch64" and S == 64 and args.log2_samples == 22:
            import glob
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))[::-1]:
                try:
                    summ = json.load(open(f))["summary"]
                    traffic, traffic_src = summ["hbm_bytes_per_launch"], os.path.relpath(f, ROOT)
                    break
                except Exception:
                    pass
        # second view of the same launch: VALU issue rate against the chip's measured rate.  The instruction
        # count per launch comes from the committed SQ counters, the chip rate from tools/ubench/valu_rate.hip.
        valu = None
        if traffic is not None:
            try:
                tag = os.path.basename(traffic_src).split("_")[0]
                for ln in open(os.path.join(ROOT, "profiles", f"{tag}_sq_counters.txt")):
                    if ln.startswith("SQ_INSTS_VALU"):
                        insts = float(ln.split()[1])
                        rate = insts / (k_avg_ms * 1e-3) / 1024          # per SIMD (256 CUs x 4)
                        valu = {"valu_insts_per_launch": insts, "achieved_per_simd_per_s": rate,
                                "chip_measured_per_simd_per_s": {"v_fma_f32": 7.4e8, "v_add_f32": 8.7e8},
                                "frac_range": [rate / 8.7e8, rate / 7.4e8],
                                "source": f"profiles/{tag}_sq_counters.txt, profiles/{tag}_valu_rate.txt"}
            except Exception:
                valu = None
        line = {
            "metric": "reassigned spectrogram columns/sec (4096-pt, hop 256, 48 kHz)" if n == 4096 else
                      f"reassigned spectrogram columns/sec ({n}-pt, hop {hop}, 48 kHz)",
            "value": value, "unit": "columns/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{S} concurrent 48 kHz streams per GPU x 2^{args.log2_samples} samples, FFT {n}, "
                                   f"hop {hop}, reassignment ON, {R} log-frequency rows, outputs float32 dB + uint8 "
                                   f"palette index" + (f"; RCCL gather of the palette-index columns to rank 0 in "
                                                       f"{nch} overlapped chunks" if world > 1 else ""),
                       "streams_per_gpu": S, "samples_per_stream": L, "columns_per_step": cols_per_step,
                       "parallelism": f"streams sharded {world} way(s)", "fused_kernel": eng.fused(n, hop, True)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": S * C * bytes_per_col,
                         "bytes_per_column": bytes_per_col, "kernel_ms": k_avg_ms,
                         "note": "algorithmic bytes (4*hop in + 4*R dB + R index out) x columns per launch / "
                                 "HIP-event duration of the column kernel(s) on the launch stream; PMC traffic: profiles/"},
        }
        if valu is not None:
            line["valu_issue"] = valu
        if world == 1 and args.workload == "batch64":
            # BASELINE configs[1] (one stream of 2^22 samples) measured beside the headline, same engine
            one = pcm[:1].contiguous()
            for _ in range(3):
                eng.batch_device(one, n, hop, True, db=db[:1], index=idx[:1], stream=cur)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            reps = 50
            for _ in range(reps):
                eng.batch_device(one, n, hop, True, db=db[:1], index=idx[:1], stream=cur)
            torch.cuda.synchronize(dev)
            line["config"]["single_stream_columns_per_s"] = C * reps / (time.perf_counter() - t1)
        if world == 1 and args.workload == "batch64":
            # the per-bin parity dump (power, column, row for every bin: what the exact-index criterion forces
            # to exist in HBM, BASELINE.md "parity mode") timed on 16 of the streams, as a second roofline point
            import ctypes as C_
            lib = emspec.load()
            Sd, Ld = 16, 1 << 20
            Cd, K = emspec.num_columns(Ld, n, hop), n // 2 + 1
            sub = pcm[:Sd, :Ld].contiguous()
            pw_ = torch.empty((Sd, Cd, K), dtype=torch.float32, device=dev)
            cl_ = torch.empty((Sd, Cd, K), dtype=torch.int32, device=dev)
            rw_ = torch.empty((Sd, Cd, K), dtype=torch.int32, device=dev)

            def dump():
                rc = lib.emspec_parity_dump_device(eng._h, sub.data_ptr(), Sd, Ld, n, hop, 1, 0, Cd, pw_.data_ptr(),
                                                   cl_.data_ptr(), rw_.data_ptr(), C_.c_void_p(cur.cuda_stream))
                assert rc == 0
            dump()
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)
            for _ in range(5):
                dump()
            e1.record(cur)
            torch.cuda.synchronize(dev)
            dms = e0.elapsed_time(e1) / 5
            bpc = 4 * hop + 12 * K
            line["roofline_parity_dump"] = {
                "bound": "hbm", "achieved": Sd * Cd * bpc / (dms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": Sd * Cd * bpc / (dms * 1e-3) / 1e9 / HBM_PEAK_GBS, "bytes_per_column": bpc,
                "columns_per_s": Sd * Cd / (dms * 1e-3), "kernel": "frames_kernel<12> (emspec_parity_dump_device)",
                "note": "algorithmic bytes = 4*hop in + 12*(N/2+1) per-bin dump out; 16 streams x 2^20 samples"}
            del pw_, cl_, rw_
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(n, hop)
            line["cpu_baseline_js"] = js_baseline(n, hop)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
