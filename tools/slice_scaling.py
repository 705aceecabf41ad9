#!/usr/bin/env python3
"""K4 time of every world-th task of each launch class (the alignment-sliced multi-GPU route's per-rank work) for world = 1..64:
is there a fixed cost per fill?  T(w) = a + b / w."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from phamclust_amd import hip
from phamclust_amd.synth import synth_packed
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
pk = synth_packed(n, 5000)
ctx = hip.Context(0); ctx.upload(pk)
stream = torch.cuda.current_stream().cuda_stream
plan = ctx.plan_dev("peq", stream)
res = torch.empty(max(int(plan["n_distinct_alignments"]), 1), dtype=torch.int64, device="cuda")
ws, ts = [], []
for w in (1, 2, 4, 8, 16, 32, 64):
    ctx.align_slice_dev(0, w, res.data_ptr(), stream)
    ms = [ctx.align_slice_dev(r % w, w, res.data_ptr(), stream)["ms_align"] for r in range(3)]
    ws.append(w); ts.append(float(np.median(ms)))
    print(f"world {w:>2d}: {ts[-1]:8.2f} ms   x world = {ts[-1] * w:8.1f}", flush=True)
A = np.vstack([np.ones(len(ws)), 1.0 / np.array(ws)]).T
a, b = np.linalg.lstsq(A, np.array(ts), rcond=None)[0]
print(f"fit T(w) = {a:.2f} + {b:.1f} / w ms")
