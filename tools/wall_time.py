#!/usr/bin/env python3
"""Whole-matrix wall time as SURVEY 8(d) defines it: upload + kernels + D2H of the condensed vector, from packed host
arrays to a host ndarray (TSV parsing and SymMatrix construction excluded)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phamclust_amd import build, hip
from phamclust_amd.synth import synth_packed
build.build_all()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
metric = sys.argv[2] if len(sys.argv) > 2 else "peq"
pk = synth_packed(n, 5000)
ctx = hip.Context(0)
ctx.upload(pk); ctx.fill(metric)                    # warm: code objects, work buffers
for _ in range(3):
    t0 = time.perf_counter(); ctx.upload(pk); t1 = time.perf_counter(); out = ctx.fill(metric); t2 = time.perf_counter()
    print(f"synth({n},5000) -m {metric}: upload {1e3 * (t1 - t0):.1f} ms + fill to host {1e3 * (t2 - t1):.1f} ms = {1e3 * (t2 - t0):.1f} ms "
          f"-> {pk.n_pairs / (t2 - t0):.3e} pairs/s", flush=True)
