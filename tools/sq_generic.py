#!/usr/bin/env python3
"""Per-kernel SQ counter summary of the two --pmc passes of a generic-path run (pmc_sq1, pmc_sq2 under <dir>):
per-wave instruction counts and wait shares for every kernel.  usage: tools/sq_generic.py <dir>"""
import csv, glob, os, sys, collections
src = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for dd in ("pmc_sq1", "pmc_sq2"):
    f = glob.glob(os.path.join(src, dd, "*", "*_counter_collection.csv"))[0]
    for r in csv.DictReader(open(f)):
        vals[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in vals.items():
    m = {c: max(x) for c, x in v.items()}      # the largest launch of each kernel
    if "SQ_WAVES" not in m or m["SQ_WAVES"] < 1000:
        continue
    nw, wc = m["SQ_WAVES"], m["SQ_WAVE_CYCLES"]
    print(f"{k}\n  waves {nw:.0f}; per wave: {m['SQ_INSTS_VALU'] / nw:.0f} VALU {m['SQ_INSTS_SALU'] / nw:.0f} SALU "
          f"{m['SQ_INSTS_LDS'] / nw:.0f} LDS {m['SQ_INSTS_VMEM_RD'] / nw:.1f} VMEM-rd {m['SQ_INSTS_VMEM_WR'] / nw:.1f} VMEM-wr; "
          f"{4 * wc / nw:.0f} cycles")
    if "SQ_WAIT_ANY" in m:
        print(f"  WAIT_ANY {m['SQ_WAIT_ANY'] / wc:.1%} ACTIVE_INST_ANY {m['SQ_ACTIVE_INST_ANY'] / wc:.1%} WAIT_INST_ANY {m['SQ_WAIT_INST_ANY'] / wc:.1%} "
              f"WAIT_INST_LDS {m['SQ_WAIT_INST_LDS'] / wc:.1%}; LDS conflict {m['SQ_LDS_BANK_CONFLICT'] / max(m['SQ_LDS_IDX_ACTIVE'], 1):.1%}")
