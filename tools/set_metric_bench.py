#!/usr/bin/env python3
"""N-sweep of the set metrics (gcs / jc / pocp / af) on one GPU: SURVEY 8(d) asks for it because at the BASELINE size
(N = 2,000: 16 MB out) a fill is a single ~50 us kernel and launch latency, not HBM, sets the roofline fraction.

Per (metric, N): device time of the fill (HIP events inside the library), pairs/s, algorithmic HBM bytes
(bitmap once + 16 B per genome + 8 B per pair) against 8 TB/s, and the wall time of the three host-facing calls:
pc_fill (pageable numpy buffer), pc_fill_borrow (context-owned pinned buffer) and pc_fill_dev (result stays in HBM).
Writes one JSON document (default profiles/<tag>_set_metric_sweep.json); run under rocprofv3 --kernel-trace --stats to
get the per-kernel durations that must agree with `device_ms`.

    python tools/set_metric_bench.py [--sizes 2000,5000,20000] [--metrics jc,gcs,pocp,af] [--out FILE] [--check 20000]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="2000,5000,20000")
ap.add_argument("--metrics", default="jc,gcs,pocp,af")
ap.add_argument("--phams", type=int, default=5000)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--check", type=int, default=20000, help="random pairs checked against the oracle per (metric, N)")
ap.add_argument("--out", default=None)
a = ap.parse_args()

import torch
from phamclust_amd import build, hip
from phamclust_amd.synth import synth_packed

build.build_all()
ctx = hip.Context(0)
rows = []
for n in [int(x) for x in a.sizes.split(",")]:
    pk = synth_packed(n, a.phams)
    t0 = time.perf_counter(); ctx.upload(pk); upload_ms = (time.perf_counter() - t0) * 1e3
    # what the set metrics need: part 1 of the upload only (metrics.py:26-157 never read a residue)
    up_sets, wall_cold = [], {}
    for _ in range(4):
        t0 = time.perf_counter(); ctx.upload(pk, residues=False); up_sets.append((time.perf_counter() - t0) * 1e3)
    for metric in a.metrics.split(","):
        w = []
        for _ in range(4):                     # SURVEY 8(d)'s wall time of one matrix: upload + kernels + D2H to pinned host
            t0 = time.perf_counter(); ctx.upload(pk, residues=False); ctx.fill(metric, True, borrow=True); w.append((time.perf_counter() - t0) * 1e3)
        wall_cold[metric] = float(np.median(w[1:]))
    upload_sets_ms = float(np.median(up_sets[1:]))
    n_pairs = pk.n_pairs
    algo_bytes = pk.n_genomes * pk.words_per_row * 8 + 16 * pk.n_genomes + 8 * n_pairs
    dev_out = torch.empty(max(n_pairs, 1), dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for metric in a.metrics.split(","):
        ctx.fill(metric, True, borrow=True)                                   # warm-up: code objects, pinned buffer, LUT
        dev_ms, wall_dev, wall_borrow, wall_page = [], [], [], []
        for _ in range(a.steps):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            st = ctx.fill_dev(metric, True, dev_out.data_ptr(), stream)
            torch.cuda.synchronize(); wall_dev.append((time.perf_counter() - t0) * 1e3)
            dev_ms.append(st["ms_total"])
            t0 = time.perf_counter(); got = ctx.fill(metric, True, borrow=True); wall_borrow.append((time.perf_counter() - t0) * 1e3)
        kept = got.copy() if a.check else None                               # the loan ends with the next fill (r04: the view says so)
        for _ in range(2):
            t0 = time.perf_counter(); page = ctx.fill(metric, True); wall_page.append((time.perf_counter() - t0) * 1e3)
        ok = None
        if a.check:
            from oracle import oracle as O
            rng = np.random.default_rng(n)
            s_idx, t_idx = rng.integers(0, n, a.check), rng.integers(0, n, a.check)
            lo, hi = np.minimum(s_idx, t_idx), np.maximum(s_idx, t_idx)
            keep = lo < hi; lo, hi = lo[keep], hi[keep]
            cond = lo * n - lo * (lo + 1) // 2 + (hi - lo - 1)
            ok = bool(np.array_equal(page[cond], O.pairs(pk, metric, lo, hi, True))) and bool(np.array_equal(page, kept))
        d = float(np.median(dev_ms))
        row = {"metric": metric, "n_genomes": n, "n_phams": pk.n_phams, "genome_pairs": n_pairs, "device_ms": d,
               "pairs_per_s_device": n_pairs / d * 1e3, "algorithmic_bytes": algo_bytes, "achieved_GBps": algo_bytes / d / 1e6,
               "hbm_frac_of_8TBps": algo_bytes / d / 1e6 / 8000.0,
               "wall_ms": {"pc_fill_dev": float(np.median(wall_dev)), "pc_fill_borrow_pinned": float(np.median(wall_borrow)),
                           "pc_fill_pageable": float(np.median(wall_page))},
               "pairs_per_s_end_to_end_pinned": n_pairs / float(np.median(wall_borrow)) * 1e3, "upload_ms": upload_ms,
               "upload_sets_only_ms": upload_sets_ms, "wall_upload_sets_plus_fill_to_pinned_host_ms": wall_cold[metric],
               "pairs_per_s_wall_incl_upload": n_pairs / wall_cold[metric] * 1e3,
               "oracle_sample_equal": ok}
        rows.append(row)
        print(json.dumps(row), flush=True)
doc = {"tool": "tools/set_metric_bench.py", "workload": f"synth(N,{a.phams})", "steps": a.steps,
       "note": "device_ms: HIP events around the fill inside the library; wall_ms: host clock around the call, result delivered "
               "as named.  An 8-byte value per pair over PCIe Gen5 x16 (~55 GB/s) bounds the host-delivered rate at ~7e9 pairs/s.",
       "rows": rows}
if a.out:
    with open(a.out, "w") as fh:
        json.dump(doc, fh, indent=1)
        fh.write("\n")
