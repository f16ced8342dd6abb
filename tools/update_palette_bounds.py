#!/usr/bin/env python3
"""Regenerate tests/golden/palette_bounds.json from one or more measurement runs of the GPU suite:
     EMSPEC_PALETTE_RATES=gpurun_out/palette_rates.jsonl python -m pytest tests -m gpu -q      (on the MI355X box)
     python tools/update_palette_bounds.py gpurun_out/palette_rates*.jsonl
Per test case: bound = max(2 x the largest measured rate, (largest measured count + 8) / cells) - the float32 sums are
order-dependent, so the count moves by a few cells from run to run; a case measured at zero gets room for eight cells."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = {}
for path in sys.argv[1:]:
    for line in open(path):
        r = json.loads(line)
        a = acc.setdefault(r["key"], {"bad": 0, "size": r["size"], "runs": 0})
        a["bad"] = max(a["bad"], r["bad"])
        a["runs"] += 1
out = {}
for k, a in sorted(acc.items()):
    rate = a["bad"] / a["size"]
    out[k] = {"measured_max": rate, "cells": a["size"], "runs": a["runs"], "bound": max(2.0 * rate, (a["bad"] + 8) / a["size"])}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "palette_bounds.json"), "w"), indent=1, sort_keys=True)
print(f"{len(out)} cases; largest measured rate {max(v['measured_max'] for v in out.values()):.3e}")
