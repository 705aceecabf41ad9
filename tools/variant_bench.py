#!/usr/bin/env python3
"""Kernel-only GCUPS of every systolic variant on uniform-length synthetic genes (tuning aid).

For each length L: 96 column genes of L residues, each aligned against `rows` row genes of
similar length (one workgroup task per 256 rows), forced through each variant that fits."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from phamclust_amd import build, hip
from phamclust_amd.genome import Genome
from phamclust_amd.pack import pack_genomes

ap = argparse.ArgumentParser()
ap.add_argument("--lens", default="60,100,150,207,260,330,420,520,640,800,1000,1200")
ap.add_argument("--rows", type=int, default=208)
ap.add_argument("--cols", type=int, default=1536)
ap.add_argument("--variants", default="8,10,12,13,14,15,16,17,18,19,20,22,24")
a = ap.parse_args()
build.build_all()
rng = np.random.default_rng(1)
aa = np.array(list("ACDEFGHIKLMNPQRSTVWY"))
ctx = hip.Context(0)
print("L     " + " ".join(f"W={w:<5d}" for w in map(int, a.variants.split(","))), flush=True)
for L in map(int, a.lens.split(",")):
    g = Genome("cols"); h = Genome("rows")
    for i in range(a.cols):
        g.add(f"c{i:03d}", "".join(aa[rng.integers(0, 20, L)]))
    for i in range(a.rows):
        h.add(f"r{i:03d}", "".join(aa[rng.integers(0, 20, max(1, L + int(rng.integers(-L // 20 - 1, L // 20 + 2))))]))
    pk = pack_genomes([g, h])
    ctx.upload(pk)
    rows = np.repeat(np.arange(a.cols, a.cols + a.rows, dtype=np.int32), a.cols)
    cols = np.tile(np.arange(a.cols, dtype=np.int32), a.rows)
    cells = float(np.sum(np.diff(pk.seq_off)[rows].astype(np.float64) * np.diff(pk.seq_off)[cols]))
    out = []
    for w in map(int, a.variants.split(",")):
        if w > 0 and L > 64 * w:                      # (w = -1: the general kernel, any length)
            out.append("   -   "); continue
        ctx.align_pairs(rows, cols, variant=w)
        ctx.align_pairs(rows, cols, variant=w)
        out.append(f"{cells / ctx.last_align_ms() / 1e6:7.0f}")
    print(f"{L:<5d} " + " ".join(out), flush=True)
