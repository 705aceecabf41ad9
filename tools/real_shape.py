#!/usr/bin/env python3
"""All six metrics on the real-collection-shaped workload synth_real(N): plan sizes, chunks, how many of the
alignments are distinct, TCUPS, which set-metric kernel family the selector picked and whether forcing another is faster; every
fill sample-checked against the oracle.   python tools/real_shape.py [-n 5000] [--out profiles/r04/real_shape.json]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from phamclust_amd import build, hip
from phamclust_amd.synth import synth_real

ap = argparse.ArgumentParser()
ap.add_argument("-n", type=int, default=5000)
ap.add_argument("--check", type=int, default=3000)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--out", default="")
ap.add_argument("--also", default="1000,2000", help="smaller collections of the same shape: peq only (where a few long alignments would be the fill's critical path), with the strip-mined launches one row per wave beside the default")
a = ap.parse_args()
build.build_all()
t0 = time.time(); pk = synth_real(a.n); t_gen = time.time() - t0
n = a.n
lens = np.diff(pk.seq_off)
ctx = hip.Context(0)
t0 = time.time(); ctx.upload(pk); t_up = time.time() - t0
stream = torch.cuda.current_stream().cuda_stream
out_dev = torch.empty(pk.n_pairs, dtype=torch.float64, device="cuda")
rng = np.random.default_rng(5)
s_idx = rng.integers(0, n - 1, a.check); t_idx = rng.integers(0, n, a.check)
lo, hi = np.minimum(s_idx, t_idx), np.maximum(s_idx, t_idx)
keep = lo < hi; lo, hi = lo[keep], hi[keep]
cond = torch.as_tensor(lo * n - lo * (lo + 1) // 2 + (hi - lo - 1), device="cuda")
from oracle import oracle as O
rec = {"workload": f"synth_real({n})", "genomes": n, "phams": pk.n_phams, "genes": int(pk.n_genes), "residues": int(pk.residues.size),
       "longest_gene": int(lens.max()), "genes_over_4096": int((lens > 4096).sum()), "generate_s": t_gen, "upload_s": t_up, "metrics": {}}
for metric in ("gcs", "jc", "pocp", "af", "aai", "peq"):
    best = None
    for _ in range(a.steps):
        st = ctx.fill_dev(metric, True, out_dev.data_ptr(), stream); torch.cuda.synchronize()
        if best is None or st["ms_total"] < best["ms_total"]:
            best = st
    got = out_dev[cond].cpu().numpy()
    want = O.pairs(pk, metric, lo, hi, as_distance=True)
    row = {"ms": best["ms_total"], "pairs_per_s": pk.n_pairs / best["ms_total"] * 1e3, "oracle_pairs": int(lo.size),
           "bit_exact": bool(np.array_equal(got, want)), "max_abs_diff": float(np.abs(got - want).max())}
    if metric in ("aai", "peq"):
        row.update(n_alignments=best["n_alignments"], n_distinct_alignments=best["n_distinct_alignments"],
                   distinct_ratio=best["n_distinct_alignments"] / max(best["n_alignments"], 1), n_cells=best["n_cells"],
                   n_distinct_cells=best["n_distinct_cells"], n_chunks=best["n_chunks"], n_tasks=best["n_tasks"], n_launches=best["n_align_launches"],
                   plan_bytes_at_56_per_alignment=56 * best["n_alignments"], ms_plan=best["ms_plan"], ms_align=best["ms_align"], ms_reduce=best["ms_reduce"],
                   tcups_distinct=best["n_distinct_cells"] / max(best["ms_align"], 1e-9) / 1e9,
                   tcups_as_the_reference_would_run=best["n_cells"] / max(best["ms_align"], 1e-9) / 1e9)
    else:
        row["selector_picked"] = ctx.last_set_kernel()
        forced = {}
        for k in ("popc", "sparse", "sparse64", "sparsecol", "walker"):
            os.environ["PC_SET_KERNEL"] = k
            ms = min(ctx.fill_dev(metric, True, out_dev.data_ptr(), stream)["ms_total"] for _ in range(a.steps))
            torch.cuda.synchronize()
            if ctx.last_set_kernel() == k:                      # (a family that does not exist for the metric is not forced)
                forced[k] = ms
                assert np.array_equal(out_dev[cond].cpu().numpy(), want), (metric, k)
        os.environ.pop("PC_SET_KERNEL", None)
        row["forced_ms"] = forced
        row["fastest"] = min(forced, key=forced.get)
    rec["metrics"][metric] = row
    print(metric, json.dumps(row), flush=True)
rec["peq_at_other_sizes"] = {}
for n2 in [int(x) for x in a.also.split(",") if x]:
    pk2 = synth_real(n2)
    ctx.upload(pk2)
    out2 = torch.empty(pk2.n_pairs, dtype=torch.float64, device="cuda")
    row = {}
    for label, pipe in (("default", None), ("PC_PIPE=0", "0")):
        if pipe is None: os.environ.pop("PC_PIPE", None)
        else: os.environ["PC_PIPE"] = pipe
        best = min((ctx.fill_dev("peq", True, out2.data_ptr(), stream) for _ in range(a.steps)), key=lambda st: st["ms_total"])
        torch.cuda.synchronize()
        row[label] = {"ms": best["ms_total"], "ms_align": best["ms_align"], "n_distinct_cells": best["n_distinct_cells"],
                      "tcups_distinct": best["n_distinct_cells"] / max(best["ms_align"], 1e-9) / 1e9, "n_launches": best["n_align_launches"]}
    os.environ.pop("PC_PIPE", None)
    r2 = np.random.default_rng(6)
    s2 = r2.integers(0, n2 - 1, 600); t2 = r2.integers(0, n2, 600)
    lo2, hi2 = np.minimum(s2, t2), np.maximum(s2, t2); k2 = lo2 < hi2; lo2, hi2 = lo2[k2], hi2[k2]
    got2 = out2[torch.as_tensor(lo2 * n2 - lo2 * (lo2 + 1) // 2 + (hi2 - lo2 - 1), device="cuda")].cpu().numpy()
    row["bit_exact"] = bool(np.array_equal(got2, O.pairs(pk2, "peq", lo2, hi2, as_distance=True))); row["oracle_pairs"] = int(lo2.size)
    rec["peq_at_other_sizes"][str(n2)] = row
    print("peq", n2, json.dumps(row), flush=True)
if a.out:
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(rec, open(a.out, "w"), indent=1)
