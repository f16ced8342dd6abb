#!/usr/bin/env python3
"""Summarise the two SQ --pmc passes of a profiling run (gpurun_out/<dir>/pmc_sq1, pmc_sq2) for the
batch launch of the fused kernel into profiles/<tag>_sq_counters.txt.  usage: tools/sq_summary.py <dir> <tag>"""
import csv, glob, os, sys
src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vals = {}
for dd in ("pmc_sq1", "pmc_sq2"):
    f = max(glob.glob(os.path.join(src, dd, "*", "*_counter_collection.csv")), key=os.path.getmtime)   # gpurun merges into the directory: take the latest run
    rows = [r for r in csv.DictReader(open(f)) if "fused4096" in r["Kernel_Name"]]
    big = max(int(r["Grid_Size"]) for r in rows)
    for r in rows:
        if int(r["Grid_Size"]) == big:
            vals[r["Counter_Name"]] = float(r["Counter_Value"])
            wg = int(r["Workgroup_Size"])
nw, wc = vals["SQ_WAVES"], vals["SQ_WAVE_CYCLES"]
iters = (1047616.0 / (nw / (wg / 64)) + 16) / 2   # columns per workgroup + 16 halo frames, two frames per iteration (bench.py: 1,047,616 columns per launch)
out = ["# SQ counters of the fused kernel for one bench.py batch launch (64 streams x 2^22 samples =",
       f"# {nw / (wg / 64):.0f} workgroups x {wg // 64} waves, {iters:.0f} two-frame iterations each), two --pmc passes:",
       "# rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES",
       "# rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS",
       "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles"]
out += [f"{k:28s} {vals[k]:.6g}" for k in sorted(vals)]
out += ["",
        f"share of wave time: WAIT_ANY {vals['SQ_WAIT_ANY'] / wc:.1%}  ACTIVE_INST_ANY {vals['SQ_ACTIVE_INST_ANY'] / wc:.1%}  WAIT_INST_ANY {vals['SQ_WAIT_INST_ANY'] / wc:.1%}",
        f"LDS bank-conflict cycles / LDS active cycles: {vals['SQ_LDS_BANK_CONFLICT'] / vals['SQ_LDS_IDX_ACTIVE']:.2%}",
        f"per wave per 2-frame iteration: {vals['SQ_INSTS_VALU'] / nw / iters:.0f} VALU, {vals['SQ_INSTS_SALU'] / nw / iters:.0f} SALU, "
        f"{vals['SQ_INSTS_LDS'] / nw / iters:.1f} LDS, {vals['SQ_INSTS_VMEM_RD'] / nw / iters:.1f} VMEM-read instructions; {4 * wc / nw / iters:.0f} cycles",
        f"VALU issue floor at 2 cycles/instruction, 4 waves per SIMD: {vals['SQ_INSTS_VALU'] / nw / iters * 4 * 2:.0f} cycles per iteration"]
try:   # VALU issue rate against the chip's measured rate (tools/ubench/valu_rate.hip)
    import json
    kms = json.load(open(os.path.join(root, "profiles", f"{tag}_bench_n1.json")))["roofline"]["kernel_ms"]
    rate = vals["SQ_INSTS_VALU"] / (kms * 1e-3) / 1024
    out.append(f"VALU wave-instructions per second and SIMD: {rate:.3g} over the {kms:.2f} ms launch (the chip saturates at 7.4e8 v_fma_f32 / "
               f"8.7e8 v_add_f32, profiles/r02_valu_rate.txt): {rate / 8.7e8:.0%}-{rate / 7.4e8:.0%} of the VALU issue rate")
except Exception as ex:
    out.append(f"(no bench line for the VALU-rate comparison: {ex})")
open(os.path.join(root, "profiles", f"{tag}_sq_counters.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out[-5:]))
