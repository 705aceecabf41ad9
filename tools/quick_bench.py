#!/usr/bin/env python3
"""Ad-hoc single-GPU timing of one metric on synth(N, P) (development aid; bench.py is the contract)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from phamclust_amd import build, hip
from phamclust_amd.synth import synth_packed

ap = argparse.ArgumentParser()
ap.add_argument("-n", type=int, default=2000)
ap.add_argument("-p", type=int, default=5000)
ap.add_argument("-m", default="peq")
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--check", type=int, default=0, help="check this many random pairs against the oracle")
a = ap.parse_args()
build.build_all()
t0 = time.time(); pk = synth_packed(a.n, a.p); t1 = time.time()
print(f"synth({a.n},{a.p}): {t1 - t0:.2f}s genes={pk.n_genes} residues={pk.residues.size}", flush=True)
ctx = hip.Context(0)
t0 = time.time(); ctx.upload(pk); print(f"upload {time.time() - t0:.2f}s", flush=True)
for i in range(a.steps):
    t0 = time.time()
    out, st = ctx.fill(a.m, True, want_stats=True)
    dt = time.time() - t0
    gc = st["n_distinct_cells"] / max(st["ms_align"], 1e-9) / 1e6 if st["n_distinct_cells"] else 0.0
    print(f"step {i}: wall {dt * 1e3:.1f} ms dev {st['ms_total']:.2f} ms plan {st['ms_plan']:.2f} align {st['ms_align']:.2f} "
          f"reduce {st['ms_reduce']:.2f} | pairs/s {pk.n_pairs / dt:.3e} | aln {st['n_alignments']} cells {st['n_cells']:.3e} "
          f"distinct aln {st['n_distinct_alignments']} cells {st['n_distinct_cells']:.3e} tasks {st['n_tasks']} launches {st['n_align_launches']} GCUPS {gc:.1f}", flush=True)
if a.check:
    from oracle import oracle as O
    rng = np.random.default_rng(1)
    n = a.n
    s_idx = rng.integers(0, n - 1, a.check); t_idx = rng.integers(0, n, a.check)
    lo, hi = np.minimum(s_idx, t_idx), np.maximum(s_idx, t_idx)
    keep = lo < hi; lo, hi = lo[keep], hi[keep]
    cond = lo * n - lo * (lo + 1) // 2 + (hi - lo - 1)
    want = O.pairs(pk, a.m, lo, hi, True)
    print("oracle check: random pairs", lo.size, "equal:", bool(np.array_equal(out[cond], want)),
          "mismatches", int((out[cond] != want).sum()), flush=True)
