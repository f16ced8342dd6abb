#!/usr/bin/env python3
"""Parity sweep over unusual row counts (64..4096, not powers of two) x four shapes, on an MI355X."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "em-spec_amd"), os.path.join(ROOT, "oracle")]
import numpy as np, emspec, oracle as O
from emspec import synth
bad=0
for rows in (64, 68, 100, 252, 1000, 1020, 1024, 1028, 3000, 4096):
    for (n,hop) in ((4096,256),(1024,256),(16384,512),(2048,2048)):
        frames=20
        pcm=synth.streams(2, n+hop*(frames-1))
        try:
            with emspec.Engine(rows=rows) as e:
                out=e.batch(pcm,n,hop,True,want=("db","index","rgba"))
        except emspec.EmspecError as ex:
            print("rows",rows,n,hop,"->",ex); continue
        odb,orgba,oidx=O.batch_f32(O.make_cfg(n,hop,True,rows=rows),pcm)
        err=float(np.max(np.abs(out["db"]-odb))); d=int(np.abs(out["index"].astype(int)-oidx.astype(int)).max())
        ok = err<8.7e-4 and d<=1 and np.array_equal(out["rgba"], O.default_lut()[out["index"]])
        if not ok: bad+=1; print("FAIL rows",rows,n,hop,err,d)
print("rows fuzz done, failures:",bad)
