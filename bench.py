#!/usr/bin/env python3
"""bench.py — reassigned spectrogram columns/s (4096-pt, hop 256, 48 kHz) on N MI355X.

A "step" is one pass of the hot path over one batch of synthetic audio that is
already resident in HBM: every rank turns its 64 streams x 2^22 samples
(BASELINE.json configs[2]; configs[3] at N=8) into 64 x 16,369 finished columns
(float32 dB + uint8 palette index per cell).  For N > 1 the streams shard
across ranks with no data-path exchange; the only collective is the gather of
the finished palette-index columns to rank 0 (north_star), done by libemspec
itself over RCCL (emspec_gather_columns: packed wire image, grouped send/recv)
per stream-chunk on a side stream, so it overlaps the next chunk's compute.

Prints ONE JSON line on rank 0 (contract: see the task statement / DESIGN.md §5, §6).
At N = 1 the line also carries `configs` (the other named BASELINE configs measured beside the
headline), `roofline_compute` (the bounds the kernel really sits under) and `cpu_baseline`.
"""
import argparse
import glob
import hashlib
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

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: peak FP32 vector
SIMDS = 256 * 4
# algorithmic flops per column (1.5 complex FFTs + per-bin + dB), the contract figure
FLOPS_PER_COLUMN = {1024: 0.10e6, 2048: 0.21e6, 4096: 0.45e6, 8192: 0.93e6, 16384: 1.9e6}


def executed_flops_per_column(n, rows=1024):
    """What the kernels really execute per column: ONE packed complex FFT (N/2 log2 N radix-2 butterflies of 10 flops:
    complex add, complex subtract, complex multiply as 2 mul + 2 fma) + the per-bin stage (N/2 bins x ~60 flops: stencils,
    three dot products, reciprocal, index arithmetic) + dB (rows x ~8) - not the 1.5 FFTs of the three-window formulation."""
    l2 = n.bit_length() - 1
    return 10.0 * (n // 2) * l2 + 60.0 * (n // 2) + 8.0 * rows


def sources_sha():
    """sha1 over the kernel / C-ABI sources (tools/sources_sha.py: the definition the library's Makefile compiles in):
    profile-derived numbers are only quoted when they were taken on these."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import sources_sha as _ss
    finally:
        sys.path.pop(0)
    return _ss.sources_sha(ROOT)


def library_sha():
    """The digits the LOADED libemspec.so was built from (emspec_build_info); None when it cannot say."""
    try:
        import emspec
        info = emspec.build_info()
        for tok in info.split():
            if tok.startswith("sources="):
                return tok[len("sources="):]
    except Exception:
        pass
    return None


def profile_for(workload, lib_sha="tree"):
    """Committed rocprofv3 summary of this workload (profiles/*_<workload>.json, written by tools/profile_json.py),
    newest first; returns (dict, fresh) where fresh says that the source tree AND the loaded library (lib_sha: what
    emspec_build_info reports; "tree" = do not check, for callers without a library) are the ones it was taken on - a
    stale prebuilt .so must not be quoted with a fresh tree's counters."""
    sha = sources_sha()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{workload}.json")))[::-1]:
        try:
            d = json.load(open(f))
            d["file"] = os.path.relpath(f, ROOT)
            return d, d.get("sources_sha") == sha and (lib_sha == "tree" or lib_sha == sha)
        except Exception:
            continue
    return None, False


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


def host_cores():
    """(cores this process may use, physical cores visible, logical CPUs visible): SMT siblings count once, and a
    cgroup CPU quota (the GPU box grants 16 CPUs of a 128-core host) caps the number of threads worth starting."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except Exception:
        cpus = list(range(os.cpu_count() or 1))
    seen = set()
    for c in cpus:
        p = f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list"
        seen.add(open(p).read().strip() if os.path.exists(p) else str(c))
    phys = max(1, len(seen))
    quota = phys
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(float(q) / float(per)))
    except Exception:
        pass
    return min(phys, quota), phys, len(cpus)


_HOST_CORES = host_cores()      # before anything loads an OpenMP runtime that might re-bind this thread


def cpu_baseline(n, hop, seconds_target=3.0):
    """Time the CPU port of the same pipeline (oracle/emspec_cpu_fast.c: Stockham radix-4 FFT, ring histogram per
    stream, vectorised dB; gcc -O3 -march=native, OpenMP proc_bind(close)) on a bounded sample of the same workload:
    one thread, then one thread per physical core.  A stand-in ("port"): the reference's CPU path is private."""
    import oracle as O
    from emspec import synth
    cores, phys, logical = _HOST_CORES
    cores = max(1, min(cores, O.max_threads()))
    cfg = O.make_cfg(n, hop, True)
    # one thread: calibrate on a short run, then ~seconds_target/3 of work
    probe = synth.streams(1, n + hop * 255)
    t0 = time.perf_counter(); O.fast_batch(cfg, probe, threads=1); dt = time.perf_counter() - t0
    rate1 = 256 / dt
    c1 = int(max(512, min(1 << 16, rate1 * seconds_target / 3)))
    base = synth.streams(1, n + hop * (c1 - 1))
    runs1 = []
    for _ in range(3):          # wall-clock over >= 3 runs, median
        t0 = time.perf_counter(); O.fast_batch(cfg, base, threads=1); runs1.append(time.perf_counter() - t0)
    dt1 = float(np.median(runs1))
    per_core = c1 / dt1
    # all cores: one stream per core, each the same length (rolled copies: same statistics, distinct data)
    ca = int(max(512, min(c1, per_core * seconds_target * 2 / 3)))
    La = n + hop * (ca - 1)
    pcm = np.stack([np.roll(base[0, :La], 977 * s) for s in range(cores)])
    runsa = []
    for _ in range(3):
        t0 = time.perf_counter(); O.fast_batch(cfg, pcm, threads=cores); runsa.append(time.perf_counter() - t0)
    dta = float(np.median(runsa))
    return {"value": cores * ca / dta, "runs": 3, "statistic": "median",
            "all_core_runs_s": runsa, "one_thread_runs_s": runs1, "unit": "columns/s", "cores": cores, "kind": "port",
            "per_core": per_core, "host_physical_cores": phys, "host_logical_cpus": logical,
            "sample": f"{cores} streams x {ca} columns on {cores} threads (one per core, OpenMP proc_bind(close); the box's CPU quota) (median of 3 runs: {dta:.1f} s); one thread: {c1} "
                      f"columns (median of 3: {dt1:.1f} s).  N={n}, hop={hop}, reassign on, float32 dB + palette index out; "
                      f"oracle/emspec_cpu_fast.c (same pipeline as the HIP kernels, Stockham radix-4 FFT, -O3 -march=native)"}


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


def host_buffer_configs(eng, pcm_dev, L, n, hop, R):
    """PCIe-inclusive rates of the host-buffer entry points on the headline shape (emspec_batch index out, emspec_batch_packed;
    pinned buffers from emspec_host_alloc), each with a PCIe roofline whose peak is the hipMemcpy rate measured here on the
    same buffers.  `value` of the bench line never includes PCIe; these are the figures a host-memory caller sees."""
    import ctypes as C_
    import emspec
    lib = emspec.load()
    S = int(pcm_dev.shape[0])
    Cn = emspec.num_columns(L, n, hop)
    pin = emspec.PinnedArray((S, L), np.float32)
    pix = emspec.PinnedArray((S, Cn, R), np.uint8)
    out = {}
    try:
        pin.array[...] = pcm_dev.cpu().numpy()
        hip = C_.CDLL("libamdhip64.so")
        scratch = torch.empty(pin.array.nbytes, dtype=torch.uint8, device=pcm_dev.device)

        def copy_rate(h2d):
            best = 0.0
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                if h2d:
                    rc = hip.hipMemcpy(C_.c_void_p(scratch.data_ptr()), C_.c_void_p(pin.array.ctypes.data), C_.c_size_t(pin.array.nbytes), 1)
                else:
                    rc = hip.hipMemcpy(C_.c_void_p(pix.array.ctypes.data), C_.c_void_p(scratch.data_ptr()), C_.c_size_t(pix.array.nbytes), 2)
                assert rc == 0
                best = max(best, (pin.array.nbytes if h2d else pix.array.nbytes) / (time.perf_counter() - t0) / 1e9)
            return best
        h2d, d2h = copy_rate(True), copy_rate(False)

        def duplex_rate():
            """both directions at once (two HIP streams, 1 GB each way): what PCIe gives a pipeline whose copies overlap"""
            other = torch.empty(pix.array.nbytes, dtype=torch.uint8, device=pcm_dev.device)
            scratch = torch.empty(pin.array.nbytes, dtype=torch.uint8, device=pcm_dev.device)
            s1, s2 = torch.cuda.Stream(device=pcm_dev.device), torch.cuda.Stream(device=pcm_dev.device)
            best = 0.0
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r1 = hip.hipMemcpyAsync(C_.c_void_p(scratch.data_ptr()), C_.c_void_p(pin.array.ctypes.data), C_.c_size_t(pin.array.nbytes), 1, C_.c_void_p(s1.cuda_stream))
                r2 = hip.hipMemcpyAsync(C_.c_void_p(pix.array.ctypes.data), C_.c_void_p(other.data_ptr()), C_.c_size_t(pix.array.nbytes), 2, C_.c_void_p(s2.cuda_stream))
                assert r1 == 0 and r2 == 0
                torch.cuda.synchronize()
                best = max(best, min(pin.array.nbytes, pix.array.nbytes) / (time.perf_counter() - t0) / 1e9)
            return best
        del scratch

        def timed(fn):
            fn()
            t = []
            for _ in range(3):
                t0 = time.perf_counter(); fn(); t.append(time.perf_counter() - t0)
            return float(np.median(t))
        o = emspec.Out(None, None, C_.c_void_p(pix.array.ctypes.data))
        dt = timed(lambda: eng._chk(lib.emspec_batch(eng._h, C_.c_void_p(pin.array.ctypes.data), S, L, n, hop, 1, C_.byref(o))))
        gbs = pin.array.nbytes / dt / 1e9
        idx_name = f"host buffers (pinned): {S} streams, FFT {n}, hop {hop}, reassignment ON, uint8 palette index out (emspec_batch)"
        out[idx_name] = {
            "columns_per_s": S * Cn / dt, "ms": dt * 1e3,
            "roofline": {"bound": "pcie", "achieved": gbs, "peak": min(h2d, d2h), "unit": "GB/s", "frac": gbs / min(h2d, d2h),
                         "note": "4*hop B in and R B out per column, both directions busy at once; peak = the slower of the hipMemcpy "
                                 "rates measured here on the same pinned buffers, one direction at a time; duplex_GBps = what each "
                                 "direction reaches when two plain hipMemcpyAsync run against each other (frac_of_duplex: against that)",
                         "h2d_GBps": h2d, "d2h_GBps": d2h}}
        # (the images of S streams always fit S * emspec_wire_bound; a dense input would not fit the index block's S*C*R bytes)
        pwire = emspec.PinnedArray((S * emspec.wire_bound(Cn, R),), np.uint8)
        wire = pwire.array
        offs = np.zeros(S + 1, np.int64)
        dt = timed(lambda: eng._chk(lib.emspec_batch_packed(eng._h, C_.c_void_p(pin.array.ctypes.data), S, L, n, hop, 1, C_.c_void_p(wire.ctypes.data),
                                                            C_.c_int64(wire.size), offs.ctypes.data_as(C_.c_void_p))))
        gbs = pin.array.nbytes / dt / 1e9
        out[f"host buffers (pinned): {S} streams, FFT {n}, hop {hop}, reassignment ON, packed wire images out (emspec_batch_packed)"] = {
            "columns_per_s": S * Cn / dt, "ms": dt * 1e3, "wire_bytes_per_column": float(offs[-1]) / (S * Cn),
            "roofline": {"bound": "pcie", "achieved": gbs, "peak": h2d, "unit": "GB/s", "frac": gbs / h2d,
                         "note": "4*hop B in per column (the images going out are ~0.18 of that); peak = the hipMemcpy H2D rate "
                                 "measured here on the same pinned buffer", "h2d_GBps": h2d, "d2h_GBps": d2h}}
        # the same call from ORDINARY (pageable) numpy arrays, reused across calls: the library overlaps the runtime's blocking copies
        # with a second host thread (round 6)
        pg_in, pg_idx = np.array(pin.array), np.empty((S, Cn, R), np.uint8)
        po = emspec.Out(None, None, C_.c_void_p(pg_idx.ctypes.data))
        dt = timed(lambda: eng._chk(lib.emspec_batch(eng._h, C_.c_void_p(pg_in.ctypes.data), S, L, n, hop, 1, C_.byref(po))))
        gbs = pg_in.nbytes / dt / 1e9
        out[f"host buffers (pageable): {S} streams, FFT {n}, hop {hop}, reassignment ON, uint8 palette index out (emspec_batch)"] = {
            "columns_per_s": S * Cn / dt, "ms": dt * 1e3,
            "equal_to_pinned_output": bool(np.array_equal(pg_idx, pix.array)) if eng.mode == emspec.MODE_EXACT else
                                      bool(np.max(np.abs(pg_idx.astype(np.int16) - pix.array.astype(np.int16))) <= 1),
            "roofline": {"bound": "pcie", "achieved": gbs, "peak": min(h2d, d2h), "unit": "GB/s", "frac": gbs / min(h2d, d2h),
                         "note": "as the pinned entry; ordinary numpy arrays with resident pages, the copies out on the library's second host thread"}}
        del pg_in, pg_idx
        # configs[1] through the host entry: ONE long stream (2^25 samples = 11.7 min of audio), pipelined by runs of its columns
        L1 = 1 << 25
        C1 = emspec.num_columns(L1, n, hop)
        p1, x1 = emspec.PinnedArray((1, L1), np.float32), emspec.PinnedArray((1, C1, R), np.uint8)
        try:
            reps = -(-L1 // L)
            p1.array[0, :] = np.tile(pin.array[0], reps)[:L1]
            o1 = emspec.Out(None, None, C_.c_void_p(x1.array.ctypes.data))
            dt = timed(lambda: eng._chk(lib.emspec_batch(eng._h, C_.c_void_p(p1.array.ctypes.data), 1, L1, n, hop, 1, C_.byref(o1))))
            gbs = p1.array.nbytes / dt / 1e9
            out[f"host buffers (pinned): configs[1] shape - 1 stream of 2^25 samples, FFT {n}, hop {hop}, reassignment ON, uint8 palette index out (emspec_batch)"] = {
                "columns_per_s": C1 / dt, "ms": dt * 1e3,
                "roofline": {"bound": "pcie", "achieved": gbs, "peak": min(h2d, d2h), "unit": "GB/s", "frac": gbs / min(h2d, d2h),
                             "note": "one stream cannot be chunked by streams: the pipeline's units are runs of >= 16,384 of its columns (+ D halo frames either side)"}}
        finally:
            p1.close()
            x1.close()
        # (last: the test makes two more HIP streams, and which copy engine a stream's transfers use follows from creation order)
        duplex = duplex_rate()
        rf = out[idx_name]["roofline"]
        rf["duplex_GBps"], rf["frac_of_duplex"] = duplex, rf["achieved"] / duplex
    finally:
        pin.close()
        pix.close()
        if "pwire" in locals():
            pwire.close()
    return out


def live_configs(dev_index, pcm_dev, n, hop, R, calls=1500):
    """The LIVE form of configs[2] (north_star: the per-frame computeSpectrogramColumn the renderer calls; README.md:36): S
    streams advance by one hop per call, ONE launch + ONE synchronisation (emspec_push_samples_multi: hop new samples per
    stream in, one finished dB column per stream out; emspec_columns: a whole frame per stream in), page-locked buffers read
    and written by the kernel in place.  What is timed is the call as the host thread sees it (wall clock around the C-ABI
    call with prebuilt ctypes arguments; filling the sample block is the audio callback's copy and not timed)."""
    import ctypes as C_
    import emspec
    lib = emspec.load()
    S = int(pcm_dev.shape[0])
    need = n + hop * (calls + 2)
    host = pcm_dev[:, :need].cpu().numpy()
    D_ = emspec.latency_columns(n, hop, True)
    out = {}
    for mode, mname in ((emspec.MODE_FAST, ""), (emspec.MODE_EXACT, "EXACT mode, ")):
        blk = emspec.PinnedArray((S, hop), np.float32)
        frm = emspec.PinnedArray((S, n), np.float32)
        odb = emspec.PinnedArray((S, 1, R), np.float32)
        cnt, first = np.zeros(S, np.int64), np.zeros(S, np.int64)
        try:
            with emspec.Engine(device=dev_index, mode=mode) as e:
                e.push_samples_multi(host[:, :n - hop].copy(), n, hop, True, want_db=False)      # prime: no frame complete yet
                args = (e._h, C_.c_void_p(blk.array.ctypes.data), C_.c_int32(S), C_.c_int64(hop), C_.c_int64(hop), C_.c_int32(n),
                        C_.c_int32(hop), C_.c_int32(1), C_.c_void_p(odb.array.ctypes.data), None, C_.c_int32(R), C_.c_int64(1),
                        C_.c_void_p(cnt.ctypes.data), C_.c_void_p(first.ctypes.data))
                ts = np.empty(calls)
                for i in range(calls):
                    blk.array[:] = host[:, n - hop + i * hop:n + i * hop]
                    t0 = time.perf_counter()
                    rc = lib.emspec_push_samples_multi(*args)
                    ts[i] = time.perf_counter() - t0
                    assert rc == 0 and cnt[0] == (1 if i >= D_ else 0), (rc, i, cnt[0])
                ts = ts[calls // 10:]
                med = float(np.median(ts))
                out[f"{mname}live: {S} streams, one hop per call (emspec_push_samples_multi, page-locked blocks), FFT {n}, hop {hop}, reassignment ON"] = {
                    "columns_per_s": S / med, "us_per_call": med * 1e6, "p90_us_per_call": float(np.percentile(ts, 90)) * 1e6,
                    "calls": int(ts.size), "streams": S,
                    "note": "one kernel launch + one stream synchronisation per call; the kernel reads the new samples from and writes "
                            "the finished dB columns to page-locked host memory (no copy engine); host time of the C-ABI call, wall clock"}
                e.reset()
                cols = np.zeros(S, np.int64)
                args = (e._h, C_.c_void_p(frm.array.ctypes.data), C_.c_int32(S), C_.c_int32(n), C_.c_int32(hop), C_.c_int32(1),
                        C_.c_void_p(odb.array.ctypes.data), None, C_.c_int32(R), C_.c_void_p(cols.ctypes.data))
                ts = np.empty(calls)
                for i in range(calls):
                    frm.array[:] = host[:, i * hop:i * hop + n]
                    t0 = time.perf_counter()
                    rc = lib.emspec_columns(*args)
                    ts[i] = time.perf_counter() - t0
                    assert rc == 0
                ts = ts[calls // 10:]
                med = float(np.median(ts))
                out[f"{mname}live: {S} streams, one frame per call (emspec_columns = {S} x computeSpectrogramColumn, page-locked blocks), FFT {n}, hop {hop}, reassignment ON"] = {
                    "columns_per_s": S / med, "us_per_call": med * 1e6, "p90_us_per_call": float(np.percentile(ts, 90)) * 1e6,
                    "calls": int(ts.size), "streams": S,
                    "note": f"{S * n * 4} B of frames per call read by the kernel over PCIe"}
        finally:
            for p_ in (blk, frm, odb):
                p_.close()
    return out


def node_host_configs(pcm_dev, L, n, hop, runs=5):
    """The same host-buffer and live entries driven from Node through the N-API addon (north_star: "host code stays in
    JavaScript/Node calling HIP through a thin C-ABI N-API addon"): em-spec_amd/js/bench_emspec.js on the SAME synthetic
    batch (handed over as a file in /dev/shm), page-locked buffers from allocPinned."""
    import shutil
    import subprocess
    node = shutil.which("node")
    js = os.path.join(ROOT, "em-spec_amd", "js")
    if node is None or not os.path.exists(os.path.join(js, "emspec.node")):
        return {"node host": {"error": "node or the built addon is not available on this box"}}
    S = int(pcm_dev.shape[0])
    path = f"/dev/shm/emspec_bench_pcm_{os.getpid()}.f32"
    try:
        pcm_dev.cpu().numpy().tofile(path)
        r = subprocess.run([node, os.path.join(js, "bench_emspec.js"), path, str(S), str(L), str(n), str(hop), str(runs), "1500"],
                           capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            return {"node host": {"error": (r.stderr or r.stdout)[-400:]}}
        d = json.loads(r.stdout.strip().splitlines()[-1])
    finally:
        if os.path.exists(path):
            os.remove(path)
    out = {}
    for key, mname in (("fast", ""), ("exact", "EXACT mode, ")):
        k = d[key]
        base = f"node host (pinned, {d['node']}): {mname}{S} streams, FFT {n}, hop {hop}, reassignment ON"
        out[f"{base}, uint8 palette index out (computeColumnsAsync)"] = {
            "columns_per_s": k["index_out"]["columns_per_s"], "ms": k["index_out"]["ms"], "runs_ms": k["index_out"]["runs"]}
        if "index_out_plain_arrays" in k:
            pk = k["index_out_plain_arrays"]
            out[f"node host (ordinary typed arrays, {d['node']}): {mname}{S} streams, FFT {n}, hop {hop}, reassignment ON, "
                f"uint8 palette index out (computeColumnsAsync)"] = {
                "columns_per_s": pk["columns_per_s"], "ms": pk["ms"], "runs_ms": pk["runs"],
                "sampled_cells_equal_to_pinned": pk["sampled_cells_equal_to_pinned"]}
        out[f"{base}, packed wire images out (computeColumnsPackedAsync)"] = {
            "columns_per_s": k["packed"]["columns_per_s"], "ms": k["packed"]["ms"], "runs_ms": k["packed"]["runs"],
            "wire_bytes_per_column": k["packed"]["wire_bytes_per_column"]}
        out[f"{base}, live: one hop per call (pushSamplesMulti)"] = {
            "columns_per_s": k["live_push"]["columns_per_s"], "us_per_call": k["live_push"]["us_per_call"],
            "p90_us_per_call": k["live_push"]["p90_us"], "calls": k["live_push"]["calls"]}
        out[f"{base}, live: one frame per call (computeSpectrogramColumns)"] = {
            "columns_per_s": k["live_frames"]["columns_per_s"], "us_per_call": k["live_frames"]["us_per_call"],
            "p90_us_per_call": k["live_frames"]["p90_us"], "calls": k["live_frames"]["calls"]}
    return out


class Watchdog:
    """A per-step deadline on a daemon thread: a rank that stops making progress (a peer died inside a collective, a kernel
    never returns) ends its process with exit code 1 - the launcher then stops the job - instead of hanging until the
    driver's own limit.  No re-exec of a process that touched the GPU: the thread only calls os._exit."""

    def __init__(self, rank):
        import threading
        self.rank, self.deadline, self.what = rank, None, ""
        self.lock = threading.Lock()
        t = threading.Thread(target=self._run, daemon=True)
        t.start()

    def arm(self, seconds, what):
        with self.lock:
            self.deadline, self.what = time.monotonic() + seconds, what

    def disarm(self):
        with self.lock:
            self.deadline = None

    def _run(self):
        while True:
            time.sleep(0.5)
            with self.lock:
                late = self.deadline is not None and time.monotonic() > self.deadline
                what = self.what
            if late:
                sys.stderr.write(f"[bench rank {self.rank}] watchdog: {what} exceeded its deadline - ending this rank (exit 1)\n")
                sys.stderr.flush()
                os._exit(1)


def time_launches(fn, stream, reps):
    """average HIP-event duration (ms) of reps back-to-back calls of fn on `stream` (after one untimed call)"""
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def roofline(cols, bytes_per_col, ms, note=None):
    ach = cols * bytes_per_col / (ms * 1e-3) / 1e9
    d = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
         "bytes_per_column": bytes_per_col, "kernel_ms": ms}
    if note:
        d["note"] = note
    return d


def main():
    # Exactly ONE line may reach stdout (the driver parses it), but RCCL prints a version banner there when a
    # communicator is created: keep the real stdout for the JSON line and send everything else to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    json_out = os.fdopen(json_fd, "w")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # (0.45 s of timed kernels at N = 1: long enough for a utilisation sampler to see)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="batch64", choices=["batch64", "single", "n16384", "n8192", "n1024", "paritydump"],
                    help="paritydump: N=1 only, a step = the per-bin dump of 16 streams x 2^20 samples (the second roofline point of the default line, here on its own so that tools/profile_workload.sh can profile it)")
    ap.add_argument("--streams", type=int, default=0, help="override streams per GPU")
    ap.add_argument("--log2-samples", type=int, default=22)
    ap.add_argument("--chunks", type=int, default=2, help="stream-chunks per step (gather overlap, N>1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="N=1: skip the side measurements of the other named configs")
    ap.add_argument("--force-chunks", type=int, default=0, help="N=1: launch per stream-chunk as the N>1 path does (no gather)")
    ap.add_argument("--reassign", type=int, default=1)
    ap.add_argument("--mode", default="fast", choices=["fast", "exact"],
                    help="exact = EMSPEC_MODE_EXACT (binary64 + 64-bit fixed-point histogram, DESIGN.md 3.7); the default line is "
                         "the fast mode and reports the exact mode's rate in `configs`")
    ap.add_argument("--gather", default="lib", choices=["lib", "torch", "loopback", "dist-loopback"],
                    help="lib = libemspec's RCCL gather (default); torch = torch.distributed.gather of raw columns; "
                         "loopback = N=1 rehearsal: the rank's own columns go through pack + RCCL self send/recv + expand; "
                         "dist-loopback = the same with torch.distributed's NCCL group alive as at N>1 (id broadcast, barriers "
                         "and reductions through it, two RCCL communicators in the process)")
    ap.add_argument("--root-streams", type=int, default=-1,
                    help="N>1: streams on rank 0, which also expands the gathered columns (default: try a few splits, keep the fastest)")
    ap.add_argument("--gather-expand", action="store_true",
                    help="N>1 / loopback: time the gather whose root EXPANDS every rank's image into plain [streams][columns][rows] arrays; "
                         "the default keeps the lossless packed images on the root (EMSPEC_GATHER_PACKED: directory + images, expanded on "
                         "demand with emspec_wire_unpack) and reports the expanding form beside it")
    ap.add_argument("--step-deadline-factor", type=float, default=20.0,
                    help="watchdog: a run may take max(--step-deadline-min, this x the slowest warm-up step) + 3 x steps x that step before the rank exits 1")
    ap.add_argument("--step-deadline-min", type=float, default=30.0, help="watchdog: the shortest per-step deadline, seconds")
    ap.add_argument("--hang-rank", type=int, default=-1, help="TEST HOOK: this rank stops (sleeps) at --hang-at-step of the timed run")
    ap.add_argument("--hang-at-step", type=int, default=1)
    ap.add_argument("--trial-budget-s", type=float, default=60.0,
                    help="N>1: wall-clock budget of the stream-split trials; when it is spent the remaining splits are skipped "
                         "(none measured: the modelled split, root 8 streams lighter per other rank, is used)")
    ap.add_argument("--dry-run-ranks", action="store_true",
                    help="TEST HOOK: every rank joins a gloo group, rank 0 prints a one-line JSON with the ranks' pids, nothing "
                         "touches a GPU (the CPU test of the self-spawning launcher); --hang-rank R makes rank R exit 3")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N>1 control flow on fewer GPUs than ranks (torch gather via host tensors)")
    args = ap.parse_args()
    args.gather_packed = not args.gather_expand

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # (a plain `bench.py --gpus N` never gets here: spawn_ranks() started the N ranks and relayed their line)
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    if args.dry_run_ranks:
        # TEST HOOK: the N > 1 start-up without a GPU - rendezvous, one reduction, rank 0's single line
        if rank == args.hang_rank and args.hang_at_step == 0:
            raise RuntimeError(f"injected failure of rank {rank} before the rendezvous")
        import torch.distributed as dist
        dist.init_process_group("gloo")
        seen = torch.zeros(world, dtype=torch.int64)
        seen[rank] = os.getpid()
        dist.all_reduce(seen)
        if rank == 0:
            json_out.write(json.dumps({"dry_run": True, "n_gpus": world, "pids": seen.tolist(), "steps": args.steps}) + "\n")
            json_out.flush()
        print(f"[bench rank {rank}] dry run: {world} ranks met")          # (stdout is stderr here: must not reach the line)
        dist.barrier()
        dist.destroy_process_group()
        if rank == args.hang_rank:
            sys.exit(3)
        return
    dist = None
    dist_on = world > 1 or args.gather == "dist-loopback"     # torch.distributed initialised (N>1, or the one-rank rehearsal of it)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29537")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    if dist_on:
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
    S = args.streams or (1 if args.workload == "single" else 64)     # streams per GPU (the job has world x S)
    L = 1 << args.log2_samples
    eng = emspec.Engine(device=dev_index, mode=emspec.MODE_EXACT if args.mode == "exact" else emspec.MODE_FAST)
    lib_sha = library_sha()       # what the LOADED library was built from; counters are quoted only when it is the tree's and the profile's
    R = eng.rows
    C = emspec.num_columns(L, n, hop)
    if args.workload == "paritydump":
        if world != 1:
            sys.exit("paritydump is a one-GPU measurement")
        import ctypes as C_
        lib = emspec.load()
        Sd, Ld = 16, 1 << 20
        Cd, K = emspec.num_columns(Ld, n, hop), n // 2 + 1
        sub = synth_device(Sd, Ld, 0, dev)
        pw_ = torch.empty((Sd, Cd, K), dtype=torch.float32, device=dev)
        cl_ = torch.empty((Sd, Cd, K), dtype=torch.int32, device=dev)
        rw_ = torch.empty((Sd, Cd, K), dtype=torch.int32, device=dev)
        cur = torch.cuda.current_stream(dev)

        def dump():
            assert lib.emspec_parity_dump_device(eng._h, sub.data_ptr(), Sd, Ld, n, hop, 1, 0, Cd, pw_.data_ptr(), cl_.data_ptr(),
                                                 rw_.data_ptr(), C_.c_void_p(cur.cuda_stream)) == 0
        for _ in range(max(args.warmup, 200)):     # a 0.65 ms kernel: >= 200 untimed launches let the clock settle on it (boxes differ by 20 % cold;
                                                   # 50 were not always enough under rocprofv3's serialised dispatches)
            dump()
        torch.cuda.synchronize(dev)
        ms = time_launches(dump, cur, args.steps)
        bpc = 4 * hop + 12 * K
        rf = roofline(Sd * Cd, bpc, ms, "algorithmic bytes = 4*hop in + 12*(N/2+1) per-bin dump out; 16 streams x 2^20 samples")
        pp, pf = profile_for("paritydump", lib_sha)
        rf["traffic"] = pp.get("hbm_bytes_per_launch") if (pp and pf) else None
        line = {"metric": "parity-dump columns/sec (4096-pt, hop 256, per-bin power + column + row)", "value": Sd * Cd / (ms * 1e-3),
                "unit": "columns/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "per-bin parity dump, 16 streams x 2^20 samples, FFT 4096, hop 256, reassignment ON",
                           "columns_per_step": Sd * Cd, "sources_sha": sources_sha(), "library": emspec.build_info()},
                "roofline": rf, "cpu_baseline": None}
        json_out.write(json.dumps(line) + "\n")
        json_out.flush()
        eng.close()
        return
    total_streams = world * S

    # ---- gather set-up.  lib: libemspec's own communicator (RCCL), id handed over through torch.distributed.
    gather_mode, gather_note = "none", None
    if dist_on:
        gather_mode = "torch" if (args.gather == "torch" or args.backend == "gloo") else "lib"
        if gather_mode == "lib":
            try:
                shard.comm_setup(eng, rank, world)
            except Exception as ex:           # keep the run alive on the old path and say so in the line
                gather_mode, gather_note = "torch", f"libemspec communicator failed ({ex}); fell back to torch.distributed.gather"
            ok = torch.tensor([1 if gather_mode == "lib" else 0], device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)      # every rank must be on the same path
            if int(ok.item()) == 0 and gather_mode == "lib":
                gather_mode, gather_note = "torch", "another rank could not create the libemspec communicator; torch.distributed.gather"
    elif args.gather == "loopback":
        eng.comm_init(emspec.comm_unique_id(), 0, 1)
        gather_mode = "lib"
    gathering = gather_mode != "none"
    cur = torch.cuda.current_stream(dev)
    comm_stream = torch.cuda.Stream(device=dev) if gathering else None
    gdev = dev if args.backend == "nccl" else torch.device("cpu")
    nbuf = 2 if gathering else 1       # N>1: two index buffers, so the gather of step k's last chunk overlaps step k+1's first kernels

    def barrier():
        if dist_on:
            # drain this rank's work first: libemspec's communicator and torch's are two RCCL communicators in one process,
            # and their kernels should not wait on peers at the same time
            torch.cuda.synchronize(dev)
            dist.barrier()
        torch.cuda.synchronize(dev)

    class Job:
        """This rank's shard of the job for one split of the streams (counts[r] streams on rank r): buffers, the
        chunked step and the pipelined gather."""

        def __init__(self, counts):
            self.counts = list(counts)
            self.uneven = len(set(counts)) > 1
            firsts = shard.first_streams(counts)
            Sl = self.S = counts[rank]
            self.pcm = synth_device(Sl, L, firsts[rank], dev)
            self.db = torch.empty((Sl, C, R), dtype=torch.float32, device=dev)
            self.idx_bufs = [torch.empty((Sl, C, R), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
            nch = max(1, min(args.chunks, min(counts))) if gathering else (max(1, min(args.force_chunks, Sl)) if args.force_chunks else 1)
            self.nch = nch
            chunk = lambda Sx: [(Sx * i // nch, Sx * (i + 1) // nch) for i in range(nch)]     # the same rule on every rank
            self.bounds = chunk(Sl)
            self.gathered = None
            self.packed = bool(args.gather_packed and gather_mode == "lib")
            self.per_rank = [chunk(c) for c in counts]
            if gathering and rank == 0:
                per_rank = self.per_rank
                if self.packed:               # [chunk] -> directory + the ranks' packed images
                    self.gathered = self.packed_buffers()
                elif gather_mode == "lib":    # [chunk] -> the ranks' blocks [streams of the chunk, C, R] one after the other
                    self.gathered = [torch.empty((sum(pr[ci][1] - pr[ci][0] for pr in per_rank), C, R), dtype=torch.uint8, device=dev)
                                     for ci in range(nch)]
                else:
                    self.gathered = [[torch.empty((pr[ci][1] - pr[ci][0], C, R), dtype=torch.uint8, device=gdev) for pr in per_rank]
                                     for ci in range(nch)]
            self.sent = [[None] * nch for _ in range(nbuf)]     # per (buffer, chunk): event after its last gather
            self.nstep = 0
            self.pending = []                                   # (buffer index, chunk index, ready event): computed, not yet gathered
            self.wire_bytes = [0, 0]                            # packed bytes this rank sent, columns they carried
            self.kev = []

        def packed_buffers(self):
            return [torch.empty((256 * (world + 1) + sum(emspec.wire_bound((pr[ci][1] - pr[ci][0]) * C, R) for pr in self.per_rank),),
                                dtype=torch.uint8, device=dev) for ci in range(self.nch)]

        def switch_to_packed(self):
            """keep the root's images packed from now on (EMSPEC_GATHER_PACKED): the root only receives"""
            self.flush()
            torch.cuda.synchronize(dev)
            self.packed = True
            if rank == 0:
                self.gathered = None
                self.gathered = self.packed_buffers()

        def do_gather(self, p, ci, ready):
            a, b = self.bounds[ci]
            ibuf = self.idx_bufs[p]
            with torch.cuda.stream(comm_stream):
                comm_stream.wait_event(ready)
                if gather_mode == "lib":
                    # pack + size exchange + send/recv + expand on comm_stream; the call synchronises comm_stream once, while
                    # the next chunk's kernels (already enqueued on the compute stream) keep the GPU busy
                    nb = eng.gather_columns(ibuf[a:b], root=0, out=self.gathered[ci] if rank == 0 else None, stream=comm_stream,
                                            loopback=(world == 1), packed=self.packed)
                    if rank != 0 or world == 1:     # (packed mode reports the root's own image size too: it is not sent)
                        self.wire_bytes[0] += nb
                        self.wire_bytes[1] += (b - a) * C if nb else 0
                else:
                    src = ibuf[a:b] if args.backend == "nccl" else ibuf[a:b].cpu()   # gloo rehearsal: host tensors
                    shard.gather_columns_into(src, self.gathered[ci] if rank == 0 else None, dst=0, uneven=self.uneven)
                self.sent[p][ci] = torch.cuda.Event()
                self.sent[p][ci].record(comm_stream)

        def step(self, timed=False):
            p = self.nstep % nbuf
            self.nstep += 1
            ibuf = self.idx_bufs[p]
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if timed else None
            for ci, (a, b) in enumerate(self.bounds):
                if self.sent[p][ci] is not None:
                    cur.wait_event(self.sent[p][ci])        # the gather that last read this chunk of ibuf
                if timed and ci == 0:
                    ev[0].record(cur)
                eng.batch_device(self.pcm[a:b], n, hop, bool(args.reassign), db=self.db[a:b], index=ibuf[a:b], stream=cur)
                if timed and ci == self.nch - 1:
                    ev[1].record(cur)
                if gathering:
                    ready = torch.cuda.Event()
                    ready.record(cur)
                    # gather the PREVIOUS chunk now that this one is enqueued behind it: the host may block in there
                    self.flush()
                    self.pending.append((p, ci, ready))
            if timed:
                self.kev.append(ev)

        def flush(self):
            while self.pending:
                self.do_gather(*self.pending.pop(0))

        def run(self, steps, timed=False, deadline_s=None):
            """barrier, `steps` steps (and the gathers they owe), barrier; returns the elapsed seconds, max over ranks.
            deadline_s: the watchdog ends this rank when the whole run - launches are asynchronous, so the host only meets the
            device again in the closing barrier - takes longer than deadline_s + 3 x steps x the slowest warm-up step."""
            self.wire_bytes[:] = [0, 0]
            if deadline_s:
                dog.arm(deadline_s + 3.0 * steps * max(warm_step_s[0], 0.01), f"a run of {steps} steps")
            barrier()
            t0 = time.perf_counter()
            for i in range(steps):
                if timed and rank == args.hang_rank and i == args.hang_at_step:     # test hook: a rank that stops making progress
                    time.sleep(3600)
                self.step(timed)
            self.flush()
            barrier()
            dog.disarm()
            el = time.perf_counter() - t0
            if dist_on:
                tmax = torch.tensor([el], dtype=torch.float64, device=gdev)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                el = float(tmax.item())
            return el

    dog = Watchdog(rank)
    warm_step_s = [0.0]               # the slowest warm-up step, seconds (set below; Job.run scales its deadline with it)
    if dist_on and gather_mode == "torch":
        # open the point-to-point connections the gather uses before anything is timed
        probe = torch.zeros(16, dtype=torch.uint8, device=gdev)
        shard.gather_columns_into(probe, [torch.empty_like(probe) for _ in range(world)] if rank == 0 else None, dst=0)

    # ---- how the streams are split over the ranks.  Rank 0 also receives and expands the other ranks' columns, so with
    # equal shards it is the slowest rank and every other GPU waits for it; a lighter shard for rank 0 evens that out.
    # How much lighter depends on what the expand and RCCL's receive kernels cost beside the column kernel on the
    # root, which only an N-GPU run shows - so a few splits are tried for three steps each and the fastest is kept.
    counts = [S] * world
    split_trials = None
    if world > 1 and gathering and args.root_streams >= 0:
        counts = shard.root_light_counts(world, total_streams, 0, max(args.root_streams, args.chunks))
    elif world > 1 and gathering and S >= 16:
        split_trials = []
        t_trials = time.perf_counter()
        warm = Job(counts)             # connections, code objects and the allocator's pools: paid before any split is timed
        warm.run(2, deadline_s=300.0)   # (connections open here: generous, but a peer that never joins must not hang the job)
        del warm
        # heaviest root first (equal shards), lightest last: if the budget runs out, what was measured includes the default
        for trial in shard.trial_splits(world, total_streams, args.chunks):
            # every rank must take the same decision: rank 0's clock decides, the flag is max-reduced
            over = torch.tensor([1.0 if (time.perf_counter() - t_trials) > args.trial_budget_s else 0.0], dtype=torch.float64, device=gdev)
            dist.all_reduce(over, op=dist.ReduceOp.MAX)
            if float(over.item()) > 0:
                split_trials.append({"streams_per_rank": None, "skipped": f"trial budget of {args.trial_budget_s:.0f} s spent"})
                break
            job = Job(trial)
            job.run(1, deadline_s=120.0)                   # the first gather of a run opens RCCL's connections
            el = job.run(3, deadline_s=120.0)
            split_trials.append({"streams_per_rank": trial, "columns_per_s": total_streams * C * 3 / el})
            del job
        measured = [t for t in split_trials if t.get("columns_per_s")]
        if measured:
            best = max(measured, key=lambda t: t["columns_per_s"])   # the same numbers on every rank (max-reduced times)
            counts = best["streams_per_rank"]
        else:                          # nothing measured inside the budget: the modelled split
            counts = shard.root_light_counts(world, total_streams, 0, max(args.chunks, total_streams - (world - 1) * (S + S // 8)))

    job = Job(counts)
    warm_s = 0.0
    dog.arm(600.0, "the warm-up")       # (the first gather opens RCCL's connections; a peer that never joins must not hang the job)
    for _ in range(max(args.warmup, 1 if gathering else 0)):   # the first gather opens RCCL's connections: never timed
        t0w = time.perf_counter()
        job.step()
        job.flush()
        torch.cuda.synchronize(dev)
        warm_s = max(warm_s, time.perf_counter() - t0w)
    job.flush()
    dog.disarm()
    warm_step_s[0] = warm_s
    step_deadline = max(args.step_deadline_min, args.step_deadline_factor * warm_s)
    elapsed = job.run(args.steps, timed=True, deadline_s=step_deadline)
    eng.device_status()           # raises if a kernel flagged a protocol error during the timed run (its columns would be invalid)
    wire_bytes = job.wire_bytes
    if dist_on:
        wb = torch.tensor(wire_bytes, dtype=torch.float64, device=gdev)
        dist.all_reduce(wb, op=dist.ReduceOp.SUM)
        wire_bytes = [float(wb[0].item()), float(wb[1].item())]
    # the same job with the images kept packed on the root (EMSPEC_GATHER_PACKED: nothing is lost, the root only receives and
    # expands on demand): three more steps after the timed run, reported beside the headline, never as the headline
    packed_cps = expand_cps = None
    if gathering and gather_mode == "lib" and not args.gather_packed:     # (also in the one-GPU loopback rehearsals)
        job.switch_to_packed()
        job.run(1, deadline_s=step_deadline)
        packed_cps = total_streams * C * 3 / job.run(3, deadline_s=step_deadline)
    elif gathering and gather_mode == "lib":
        # the default line times the packed form; the expanding form (the root turns every image back into plain arrays, as at
        # N = 1) runs three steps after it on the same split and is reported beside it
        args_packed_was = job.packed
        job.flush()
        torch.cuda.synchronize(dev)
        job.packed = False
        if rank == 0:
            job.gathered = None
            job.gathered = [torch.empty((sum(pr[ci][1] - pr[ci][0] for pr in job.per_rank), C, R), dtype=torch.uint8, device=dev)
                            for ci in range(job.nch)]
        job.run(1, deadline_s=step_deadline)
        expand_cps = total_streams * C * 3 / job.run(3, deadline_s=step_deadline)
        job.packed = args_packed_was
    S_nominal = S
    # every rank's own column-kernel time per step (HIP events on its launch stream), gathered for the line
    my_kms = float(np.mean([a.elapsed_time(b) for a, b in job.kev])) if job.kev else 0.0
    rank_kms = [my_kms]
    if dist_on:
        kt = torch.zeros(world, dtype=torch.float64, device=gdev)
        kt[rank] = my_kms
        dist.all_reduce(kt, op=dist.ReduceOp.SUM)
        rank_kms = [float(v) for v in kt.tolist()]
    # what the root's expand costs by itself: one other rank's chunk worth of columns, packed and expanded standalone,
    # times the (world - 1) images per gather and the chunks per step
    expand_ms = None
    if rank == 0 and gathering and gather_mode == "lib":
        a, b = job.bounds[0]
        src = job.idx_bufs[0][a:b]
        wire = torch.empty((emspec.wire_bound((b - a) * C, R),), dtype=torch.uint8, device=dev)
        nbw = eng.wire_pack(src, wire, stream=cur)
        back = torch.empty_like(src)
        one = time_launches(lambda: eng.wire_unpack(wire, nbw, back, stream=cur), cur, 5)
        expand_ms = one * max(1, world - 1) * job.nch
        del wire, back
    S, pcm, db, idx, nch, kev = job.S, job.pcm, job.db, job.idx_bufs[0], job.nch, job.kev   # rank 0's shard, for the roofline below

    if rank == 0:
        cols_per_step = total_streams * C
        value = cols_per_step * args.steps / elapsed
        # dominant-kernel roofline: algorithmic bytes per column for this output mode
        # (4*hop in, + 4*R dB out, + R palette-index out)
        bytes_per_col = 4 * hop + 4 * R + R
        kms = [a.elapsed_time(b) for a, b in kev]           # HIP events on the launch stream
        k_avg_ms = float(np.mean(kms))
        rf = roofline(S * C, bytes_per_col, k_avg_ms,
                      "algorithmic bytes (4*hop in + 4*R dB + R index out) x columns per launch / HIP-event duration of "
                      "the column kernel(s) on the launch stream; PMC traffic: profiles/")
        rf["algorithmic_bytes_per_launch"] = S * C * bytes_per_col
        prof, fresh = (None, False)
        if S == 64 and args.log2_samples == 22:
            prof, fresh = profile_for(args.workload if args.mode == "fast" else args.workload.replace("batch64", "exact64"), lib_sha)
        rf["traffic"] = prof["hbm_bytes_per_launch"] if (prof and fresh and "hbm_bytes_per_launch" in prof) else None
        rf["traffic_source"] = (f"{prof['file']} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; sources {prof['sources_sha']}"
                                f"{'' if fresh else ', STALE: kernels changed since, traffic withheld'})") if prof else None
        wl = (f"{S_nominal} concurrent 48 kHz streams per GPU x 2^{args.log2_samples} samples, FFT {n}, hop {hop}, reassignment "
              f"{'ON' if args.reassign else 'OFF'}, {R} log-frequency rows, outputs float32 dB + uint8 palette index")
        if gathering:
            wl += (f"; gather of the palette-index columns to rank 0 in {nch} overlapped chunks per step by "
                   + (("libemspec over RCCL (emspec_gather_columns, lossless packed wire image; the root "
                       + ("KEEPS the images packed: directory + images, expanded on demand by emspec_wire_unpack)" if args.gather_packed
                          else "expands every image into plain [streams][columns][rows] arrays)")) if gather_mode == "lib"
                      else "torch.distributed.gather (raw columns)"))
        line = {
            "metric": "reassigned spectrogram columns/sec (4096-pt, hop 256, 48 kHz)" if n == 4096 else
                      f"reassigned spectrogram columns/sec ({n}-pt, hop {hop}, 48 kHz)",
            "value": value, "unit": "columns/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64" if args.mode == "exact" else "f32", "data": "synthetic",
            "config": {"workload": wl + ("; EXACT mode (binary64, 64-bit fixed-point histogram)" if args.mode == "exact" else ""), "streams_per_gpu": S_nominal, "streams_per_rank": counts, "samples_per_stream": L, "columns_per_step": cols_per_step,
                       "parallelism": f"streams sharded {world} way(s)", "fused_kernel": eng.fused(n, hop, True),
                       "sources_sha": sources_sha(), "library": emspec.build_info(),
                       "library_matches_sources": lib_sha == sources_sha()},
            "roofline": rf,
        }
        if gathering:
            line["gather"] = {"path": gather_mode, "note": gather_note, "chunks_per_step": nch, "split_trials": split_trials,
                              "wire_bytes_per_column": (wire_bytes[0] / wire_bytes[1]) if wire_bytes[1] else None,
                              "raw_bytes_per_column": R,
                              "rccl_world": eng.comm_world if gather_mode == "lib" else None,      # what RCCL itself reports
                              "root_keeps_images_packed": bool(args.gather_packed and gather_mode == "lib"),
                              "packed_columns_per_s": packed_cps,     # the same split with EMSPEC_GATHER_PACKED (3 steps after the timed run)
                              "expanding_columns_per_s": expand_cps,  # ... or, when the packed form was timed, with the root expanding
                              "step_deadline_s": step_deadline,
                              "kernel_ms_per_rank": rank_kms,                 # column-kernel time per step on every rank (HIP events)
                              "streams_per_rank": counts,
                              "root_expand_ms_standalone": expand_ms}
        # the bounds this kernel really sits under (it is not HBM-bound: SURVEY.md §8(d) consistency warning)
        rc = {"flops_per_column": FLOPS_PER_COLUMN.get(n),
              "flop_note": "flops_per_column / flop_frac are ALGORITHMIC-EQUIVALENT (SURVEY.md §8(d): the three-window formulation, 1.5 FFTs); "
                           "executed_* count what the kernel really does (one packed FFT + spectral stencils)",
              "fp32_peak_tflops": FP32_PEAK_TFLOPS}
        if rc["flops_per_column"]:
            tf = rc["flops_per_column"] * (S * C) / (k_avg_ms * 1e-3) / 1e12
            rc["achieved_tflops"], rc["flop_frac"] = tf, tf / FP32_PEAK_TFLOPS
            rc["executed_flops_per_column"] = executed_flops_per_column(n, R)
            xtf = rc["executed_flops_per_column"] * (S * C) / (k_avg_ms * 1e-3) / 1e12
            rc["executed_tflops"], rc["executed_flop_frac"] = xtf, xtf / FP32_PEAK_TFLOPS
        if prof and fresh and prof.get("valu_insts_per_column") and prof.get("clock_ghz"):
            # cycle domain: a wave64 VALU instruction occupies its SIMD for 2 cycles (MI355X_MICROARCH.md), so the VALU
            # pipes of 1024 SIMDs offer clock/2 wave-instructions per second each
            rate = prof["valu_insts_per_column"] * (S * C) / (k_avg_ms * 1e-3)
            rc["valu_util"] = rate * 2.0 / (SIMDS * prof["clock_ghz"] * 1e9)
            rc["valu_insts_per_column"] = prof["valu_insts_per_column"]
            rc["clock_ghz"] = prof["clock_ghz"]
            rc["clock_note"] = prof.get("clock_note")
            rc["wait_any_share"] = prof.get("wait_any_share")     # SQ_WAIT_ANY / SQ_WAVE_CYCLES: s_waitcnt + barrier
            rc["source"] = f"{prof['file']} (SQ counters and clock of the same launch; sources {prof['sources_sha']})"
        elif prof:
            rc["valu_util"] = None
            rc["source"] = f"{prof['file']} is stale (kernels changed since): counter-derived fields withheld"
        line["roofline_compute"] = rc
        # the bound the kernel really sits under, in the same shape as `roofline`: VALU issue in the cycle domain (a wave64
        # vector instruction holds its SIMD's pipe 2 cycles - 4 for binary64 - MI355X_MICROARCH.md), with the LDS pipe beside it
        if prof and fresh and prof.get("valu_insts_per_column") and prof.get("clock_ghz"):
            # float32 kernels: 2 cycles per wave64 instruction.  EXACT mode: binary64 fma / add / mul / compare hold the pipe 4 cycles
            # (tools/ubench/f64_rate.hip: 4.06 measured), the rest (integer, float32, conversions, selects) 2; the ISA of
            # exact_fused4096_lr_kernel's frame loop is 41 % binary64 (568 of 1,369 vector instructions, `hipcc -S`; before the
            # thread-constant address table and the division-free dB stage: 37 % of 1,537; round 4's kernel: 48 % of 1,434 - its
            # dB stage was binary64), so the mix costs ~2.83 cycles; both extremes are given as well.
            f64_share = 0.415 if args.mode == "exact" else 0.0
            cyc = 2.0 + 2.0 * f64_share
            rate = prof["valu_insts_per_column"] * (S * C) / (k_avg_ms * 1e-3)
            peak = SIMDS * prof["clock_ghz"] * 1e9
            sq = prof.get("sq") or {}
            lds_busy = (sq.get("SQ_LDS_IDX_ACTIVE", 0.0) / (256.0 * prof["rocprof_median_ms"] * 1e-3 * prof["clock_ghz"] * 1e9)
                        if prof.get("rocprof_median_ms") else None)
            line["roofline_valu"] = {
                "bound": "valu-issue", "achieved": rate * cyc / 1e9, "peak": peak / 1e9, "unit": "G SIMD-cycles/s",
                "frac": rate * cyc / peak, "cycles_per_wave_instruction": cyc, "binary64_share_of_vector_instructions": f64_share,
                "frac_if_all_2_cycle": rate * 2.0 / peak, "frac_if_all_4_cycle": (rate * 4.0 / peak) if f64_share else None,
                "valu_wave_insts_per_column": prof["valu_insts_per_column"], "clock_ghz": prof["clock_ghz"],
                "lds_pipe_busy_frac": lds_busy, "lds_bank_conflict_share": prof.get("lds_bank_conflict_share"),
                "wait_any_share": prof.get("wait_any_share"),
                "note": "live wave-instruction rate (committed SQ_INSTS_VALU per column x this run's columns/s) x cycles per instruction / "
                        "(1024 SIMDs x the profiled clock); lds_pipe_busy_frac = SQ_LDS_IDX_ACTIVE / (256 CUs x kernel cycles) of the "
                        "profiled launch",
                "source": prof["file"]}
        else:
            line["roofline_valu"] = None

        if world == 1 and args.workload == "batch64" and not args.no_configs:
            # ---- the other named BASELINE configs, measured beside the headline on the same engine and input
            cfgs = {}

            def measure(name, Sx, Lx, nx, hx, re, reps):
                Cx = emspec.num_columns(Lx, nx, hx)
                px = pcm[:Sx, :Lx].contiguous()
                dbx = db.view(-1)[:Sx * Cx * R].view(Sx, Cx, R)
                ixx = idx.view(-1)[:Sx * Cx * R].view(Sx, Cx, R)
                ms = time_launches(lambda: eng.batch_device(px, nx, hx, re, db=dbx, index=ixx, stream=cur), cur, reps)
                bpc = 4 * hx + 5 * R
                cfgs[name] = {"columns_per_s": Sx * Cx / (ms * 1e-3), "columns_per_launch": Sx * Cx, "kernel_ms": ms,
                              "fused_kernel": eng.fused(nx, hx, re), "roofline": roofline(Sx * Cx, bpc, ms)}
                if FLOPS_PER_COLUMN.get(nx):
                    cfgs[name]["flop_frac"] = FLOPS_PER_COLUMN[nx] * Sx * Cx / (ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS
                return cfgs[name]

            measure("configs[0] shape: 1 stream, FFT 1024, hop 256, reassignment OFF", 1, 1 << 20, 1024, 256, False, 50)
            one = measure("configs[1]: 1 stream, FFT 4096, hop 256, reassignment ON", 1, L, 4096, 256, True, 50)
            c4 = measure("configs[4]: 64 streams, FFT 16384, hop 512, reassignment ON", 64, L, 16384, 512, True, 3)
            p4, f4 = profile_for("n16384", lib_sha)
            if p4:
                c4["roofline"]["traffic"] = p4.get("hbm_bytes_per_launch") if f4 else None
                c4["roofline"]["traffic_source"] = f"{p4['file']}{'' if f4 else ' (STALE: kernels changed since, withheld)'}"
            measure("64 streams, FFT 4096, hop 256, reassignment OFF", 64, L, 4096, 256, False, 3)
            measure("64 streams, FFT 1024, hop 256, reassignment ON", 64, 1 << 20, 1024, 256, True, 5)
            # EXACT mode (binary64 + 64-bit fixed point: indices equal to a float64 implementation, bytes reproducible) on
            # a second engine: configs[2] itself (64 streams; one fused kernel since round 4, no workspace) and configs[4] on
            # all 64 streams (N = 16384 runs the two-kernel records path, streams in chunks of the record workspace)
            xeng = emspec.Engine(device=dev_index, mode=emspec.MODE_EXACT)
            xname = "EXACT mode, configs[2]: 64 streams, FFT 4096, hop 256, reassignment ON"
            x4name = "EXACT mode, configs[4]: 64 streams, FFT 16384, hop 512, reassignment ON"
            x1name = "EXACT mode, 64 streams, FFT 1024, hop 256, reassignment ON"
            x2name = "EXACT mode, 64 streams, FFT 2048, hop 256, reassignment ON"
            for name, Sx, nx, hx, reps in ((xname, 64, 4096, 256, 3),
                                           (x4name, 64, 16384, 512, 2),
                                           (x1name, 64, 1024, 256, 5),      # one kernel since round 6 (four frames per team and half-iteration)
                                           (x2name, 64, 2048, 256, 5)):
                Cx = emspec.num_columns(L, nx, hx)
                px = pcm[:Sx].contiguous()
                if Sx * Cx * R <= db.numel():
                    dbx = db.view(-1)[:Sx * Cx * R].view(Sx, Cx, R)
                    ixx = idx.view(-1)[:Sx * Cx * R].view(Sx, Cx, R)
                else:      # (a smaller FFT yields a few columns more from the same samples than the headline's buffers hold)
                    dbx = torch.empty((Sx, Cx, R), dtype=torch.float32, device=dev)
                    ixx = torch.empty((Sx, Cx, R), dtype=torch.uint8, device=dev)
                ms = time_launches(lambda: xeng.batch_device(px, nx, hx, True, db=dbx, index=ixx, stream=cur), cur, reps)
                del dbx, ixx
                cfgs[name] = {"columns_per_s": Sx * Cx / (ms * 1e-3), "columns_per_launch": Sx * Cx, "kernel_ms": ms,
                              "dtype": "f64", "fused_kernel": xeng.fused(nx, hx, True), "roofline": roofline(Sx * Cx, 4 * hx + 5 * R, ms)}
            px_, fx_ = profile_for("exact64", lib_sha)
            if px_:
                cfgs[xname]["roofline"]["traffic"] = px_.get("hbm_bytes_per_launch") if fx_ else None
                cfgs[xname]["roofline"]["traffic_source"] = f"{px_['file']}{'' if fx_ else ' (STALE: kernels changed since, withheld)'}"
            px4, fx4 = profile_for("exact_n16384", lib_sha)
            if px4:
                cfgs[x4name]["roofline"]["traffic"] = px4.get("hbm_bytes_per_launch") if fx4 else None
                cfgs[x4name]["roofline"]["traffic_source"] = f"{px4['file']}{'' if fx4 else ' (STALE: kernels changed since, withheld)'}"
                cfgs[x4name]["kernel_split"] = px4.get("kernel_split") if fx4 else None
            px1, fx1 = profile_for("exact_n1024", lib_sha)
            if px1:
                cfgs[x1name]["roofline"]["traffic"] = px1.get("hbm_bytes_per_launch") if fx1 else None
                cfgs[x1name]["roofline"]["traffic_source"] = f"{px1['file']}{'' if fx1 else ' (STALE: kernels changed since, withheld)'}"
            xeng.device_status()      # a protocol error of the fused kernels' bounded waits would surface here
            xeng.close()
            if not args.no_cpu_baseline:
                # the binary64 bit model (oracle/emspec_exact.c, scalar C, one thread) on a bounded sample: the CPU figure beside
                # the EXACT mode's rate ("port": there is no reference CPU path to time)
                import oracle as O
                from emspec import synth as _synth
                xs = _synth.streams(1, 4096 + 256 * 1023)
                t0 = time.perf_counter(); O.batch_exact(O.make_cfg(4096, 256, True), xs, want=("db", "index"), threads=1); dtx = time.perf_counter() - t0
                cfgs[xname]["cpu_port_columns_per_s_one_thread"] = 1024 / dtx
            # ---- the host-buffer entries (what a Node host calls: every byte crosses PCIe): emspec_batch with the uint8 palette
            # index out, and emspec_batch_packed (the index columns as lossless wire images), from page-locked buffers, on the
            # three-stage pipeline (H2D | kernels | D2H on three HIP streams).  Their roofline is PCIe: the measured hipMemcpy
            # rate of the same buffers, one direction.
            try:
                cfgs.update(host_buffer_configs(eng, pcm, L, n, hop, R))
            except Exception as ex:            # never lose the line over the side measurement
                cfgs["host buffers"] = {"error": repr(ex)}
            # ---- configs[2] in its LIVE form: 64 streams, one hop (or one frame) per call, one launch
            try:
                cfgs.update(live_configs(dev_index, pcm, n, hop, R))
            except Exception as ex:
                cfgs["live"] = {"error": repr(ex)}
            # ---- the same entries driven from Node through the N-API addon
            try:
                cfgs.update(node_host_configs(pcm, L, n, hop))
            except Exception as ex:
                cfgs["node host"] = {"error": repr(ex)}
            line["configs"] = cfgs
            line["config"]["single_stream_columns_per_s"] = one["columns_per_s"]
            # `value` is the float32 (FAST) mode; the mode that meets north_star's exact-index criterion against a binary64
            # (JavaScript) evaluation is EXACT: the pair, side by side at the top of the line
            line["exact_mode"] = {"value": cfgs[xname]["columns_per_s"], "unit": "columns/s", "kernel_ms": cfgs[xname]["kernel_ms"],
                                  "dtype": "f64", "ratio_to_value": cfgs[xname]["columns_per_s"] / line["value"],
                                  "note": "the same workload (configs[2]) in EMSPEC_MODE_EXACT: binary64 arithmetic + 64-bit fixed-point "
                                          "histogram, bit-identical to the binary64 bit model; `value` above is the float32 mode"}

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
                rcode = lib.emspec_parity_dump_device(eng._h, sub.data_ptr(), Sd, Ld, n, hop, 1, 0, Cd, pw_.data_ptr(),
                                                      cl_.data_ptr(), rw_.data_ptr(), C_.c_void_p(cur.cuda_stream))
                assert rcode == 0
            for _ in range(50):        # a 0.7 ms kernel: let the clock settle on it before timing (boxes differ by 20 % cold)
                dump()
            dms = time_launches(dump, cur, 40)
            bpc = 4 * hop + 12 * K
            pd = roofline(Sd * Cd, bpc, dms, "algorithmic bytes = 4*hop in + 12*(N/2+1) per-bin dump out; 16 streams x 2^20 samples")
            pd["columns_per_s"] = Sd * Cd / (dms * 1e-3)
            pd["kernel"] = "parity dump (emspec_parity_dump_device)"
            pp, pf = profile_for("paritydump", lib_sha)
            pd["traffic"] = pp.get("hbm_bytes_per_launch") if (pp and pf) else None
            if pp:
                pd["traffic_source"] = f"{pp['file']}{'' if pf else ' (STALE, withheld)'}"
            line["roofline_parity_dump"] = pd
            del pw_, cl_, rw_
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(n, hop)
            line["cpu_baseline_js"] = js_baseline(n, hop)
        else:
            line["cpu_baseline"] = None
        json_out.write(json.dumps(line) + "\n")
        json_out.flush()
    eng.device_status()           # raises if a kernel flagged a protocol error during the run (the line above would be invalid)
    if dist_on:
        dog.arm(120.0, "the final barrier")
        dist.barrier()
        dist.destroy_process_group()
        dog.disarm()
    eng.close()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpus():
    """GPUs this process may use, counted WITHOUT a HIP call (the relay parent of an N > 1 job must not hold a GPU context
    while its ranks run): the KFD topology's nodes with SIMDs, cut down by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES /
    CUDA_VISIBLE_DEVICES.  Falls back to torch.cuda.device_count() where the topology is not readable."""
    import glob
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    count = 0
    for f in nodes:
        # a container lists every GPU of the host here while its device cgroup admits only some: a node whose properties
        # cannot be read, or whose render node cannot be opened, is not ours
        try:
            props = dict(ln.split(None, 1) for ln in open(f) if " " in ln.strip())
            if int(props.get("simd_count", "0")) <= 0:
                continue                                   # a CPU node
            minor = int(props.get("drm_render_minor", "-1"))
        except (OSError, ValueError):
            continue
        if minor >= 0 and not os.access(f"/dev/dri/renderD{minor}", os.R_OK | os.W_OK):
            continue
        count += 1
    if not nodes:
        return torch.cuda.device_count()
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            count = min(count, len([x for x in v.split(",") if x.strip() != ""]))
    return count


def spawn_ranks(argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset: the driver's own command form):
    start the N ranks as a CHILD process - `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port <free> bench.py <same arguments>` - relay rank 0's single JSON line to stdout, everything else
    to stderr, and return the child's exit code.  Returns None when there is nothing to spawn (N = 1, or already a rank of
    an external launcher).  This parent makes NO GPU call (argument parsing and a device count read from the KFD topology in
    sysfs: visible_gpus()) and never execs: the ranks are children (rccl.h:220 ncclCommInitRank runs in them).  A job that
    fails before rank 0 printed its line leaves ONE JSON error line on stdout (exit code, the last 2 KB of the ranks' stderr)."""
    import signal
    import subprocess
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument("--gpus", type=int, default=1)
    pre.add_argument("--backend", default="nccl")
    pre.add_argument("--dry-run-ranks", action="store_true")
    known, _ = pre.parse_known_args(argv)
    if known.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return None
    if known.backend == "nccl" and not known.dry_run_ranks:
        have = visible_gpus()
        if have < known.gpus:          # (sysfs may hide what the runtime can open: ask torch before refusing - counting devices does not initialise HIP on this image)
            have = max(have, torch.cuda.device_count())
        if have < known.gpus:          # RCCL refuses two ranks on one device: say so in a line the driver can parse
            print(json.dumps({"error": f"--gpus {known.gpus} with --backend nccl needs {known.gpus} visible GPUs, found {have}",
                              "n_gpus": known.gpus, "visible_gpus": have, "metric": None, "value": None}), flush=True)
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, _HOST_CORES[0] // known.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(known.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    sys.stderr.write("[bench] starting %d ranks: %s\n" % (known.gpus, " ".join(cmd)))
    sys.stderr.flush()
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, text=True, bufsize=1)
    # the ranks' stderr is relayed as it comes; its last 2 KB are kept for the error line of a failed job
    import collections
    import threading
    tail = collections.deque()
    tail_len = [0]
    rank_errors = []                   # the ranks' own one-line records ({"rank": r, "error": ...}), at most 16

    def relay_stderr():
        for ln in child.stderr:
            sys.stderr.write(ln)
            if ln.startswith('{"rank"') and len(rank_errors) < 16:
                rank_errors.append(ln.strip()[:600])
            tail.append(ln)
            tail_len[0] += len(ln)
            while tail_len[0] > 2048 and len(tail) > 1:
                tail_len[0] -= len(tail.popleft())
    relay = threading.Thread(target=relay_stderr, daemon=True)
    relay.start()

    def forward(signum, _frame):       # the driver's timeout must reach the ranks, not only this relay
        try:
            child.send_signal(signum)
        except Exception:
            pass
    for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sg, forward)
    lines = 0
    for ln in child.stdout:
        if ln.startswith("{") and lines == 0:
            sys.stdout.write(ln)
            sys.stdout.flush()
            lines += 1
        else:
            sys.stderr.write(ln)
    rc = child.wait()
    relay.join(timeout=10.0)
    if rc == 0 and lines != 1:
        sys.stderr.write(f"[bench] the ranks exited 0 but printed {lines} JSON lines\n")
        rc = 1
    rc = rc if rc >= 0 else 128 - rc
    if rc != 0 and lines == 0:
        # a failed N > 1 run still leaves ONE parseable line: what failed, on how many ranks, and the end of their output
        print(json.dumps({"error": f"the {known.gpus}-rank job failed (exit code {rc})", "n_gpus": known.gpus, "rc": rc,
                          "metric": None, "value": None, "rank_errors": rank_errors, "stderr_tail": "".join(tail)[-2048:]}), flush=True)
    return rc


if __name__ == "__main__":
    _rc = spawn_ranks(sys.argv[1:])
    if _rc is not None:
        sys.exit(_rc)
    try:
        main()
    except SystemExit:
        raise
    except BaseException as _ex:      # any rank's failure ends the job with a non-zero exit (the launcher then stops the others)
        import traceback
        sys.stderr.write(json.dumps({"rank": int(os.environ.get("RANK", "0")), "error": repr(_ex)[:500]}) + "\n")
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)
