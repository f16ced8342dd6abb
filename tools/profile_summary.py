#!/usr/bin/env python3
"""Copy the judged summaries of a gpurun_out/<round> profiling run into profiles/ (tracked).

usage: tools/profile_summary.py gpurun_out/r01 r01
Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), profiles/<tag>_bench_n1.json
and profiles/<tag>_hbm_traffic.json (FETCH_SIZE / WRITE_SIZE passes, corrected as
/opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE x2 on gfx950, KiB units)."""
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
os.makedirs(dst, exist_ok=True)
ks = sorted(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime, reverse=True)   # latest run first (gpurun merges into the directory)
if ks:
    shutil.copy(ks[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
bj = os.path.join(src, "bench_n1.json")
if os.path.exists(bj):
    shutil.copy(bj, os.path.join(dst, f"{tag}_bench_n1.json"))
out = {}
for d, name in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = sorted(glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv")), key=os.path.getmtime, reverse=True)
    if not f:
        continue
    vals = {}
    rows = [r for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == name and "emspec" in r["Kernel_Name"]]
    fused = [r for r in rows if "fused4096" in r["Kernel_Name"]]     # the headline kernel, when it ran
    rows = fused or rows
    big = max(int(r["Grid_Size"]) for r in rows)      # the timed batch launches (bench.py also times one-stream launches)
    for r in rows:
        if int(r["Grid_Size"]) == big:
            vals.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
    out[name] = {k: {"per_launch_KiB": v, "mean_KiB": sum(v) / len(v)} for k, v in vals.items()}
if out:
    kern = next(iter(out.get("FETCH_SIZE", out.get("WRITE_SIZE"))))
    fetch = out.get("FETCH_SIZE", {}).get(kern, {}).get("mean_KiB", 0.0)
    write = out.get("WRITE_SIZE", {}).get(kern, {}).get("mean_KiB", 0.0)
    out["summary"] = {
        "kernel": kern,
        "read_bytes_per_launch": 2.0 * fetch * 1024,      # gfx950: FETCH_SIZE reports 1/2 of a coalesced stream
        "write_bytes_per_launch": write * 1024,
        "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024,
        "correction": "FETCH_SIZE x2 (gfx950), counters in KiB; separate --pmc passes (MI355X_MICROARCH.md §HBM)",
        "workload": "bench.py default: 64 streams x 2^22 samples, N=4096, hop=256, dB + index outputs",
    }
    json.dump(out, open(os.path.join(dst, f"{tag}_hbm_traffic.json"), "w"), indent=1)
print("wrote", sorted(os.listdir(dst)))
