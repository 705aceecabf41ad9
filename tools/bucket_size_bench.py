#!/usr/bin/env python3
"""K4 rate against BUCKET SIZE (rows aligned against one column gene), default chooser, uniform lengths: what a collection of
small phams, a small matrix or a rank's shard pays per bucket.   python tools/bucket_size_bench.py [--lens 100,207,420,800]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from phamclust_amd import build, hip
from phamclust_amd.genome import Genome
from phamclust_amd.pack import pack_genomes

ap = argparse.ArgumentParser()
ap.add_argument("--lens", default="100,207,420,800")
ap.add_argument("--rows", default="1,2,3,4,6,8,12,16,24,32,48,64,128,208")
ap.add_argument("--alignments", type=int, default=400000)
a = ap.parse_args()
build.build_all()
rng = np.random.default_rng(1)
aa = np.array(list("ACDEFGHIKLMNPQRSTVWY"))
ctx = hip.Context(0)
print("L     W  | GCUPS at rows per bucket: " + " ".join(f"{r:>6s}" for r in a.rows.split(",")), flush=True)
for L in map(int, a.lens.split(",")):
    out = []
    for R in map(int, a.rows.split(",")):
        ncols = max(64, min(20000, a.alignments * 207 * 207 // (L * L) // R))
        g = Genome("cols"); h = Genome("rows")
        for i in range(ncols):
            g.add(f"c{i:05d}", "".join(aa[rng.integers(0, 20, L)]))
        for i in range(R):
            h.add(f"r{i:03d}", "".join(aa[rng.integers(0, 20, max(1, L + int(rng.integers(-L // 20 - 1, L // 20 + 2))))]))
        pk = pack_genomes([g, h])
        ctx.upload(pk)
        rows = np.repeat(np.arange(ncols, ncols + R, dtype=np.int32), ncols)
        cols = np.tile(np.arange(ncols, dtype=np.int32), R)
        cells = float(np.sum(np.diff(pk.seq_off)[rows].astype(np.float64) * np.diff(pk.seq_off)[cols]))
        ctx.align_pairs(rows, cols)
        ctx.align_pairs(rows, cols)
        out.append(f"{cells / ctx.last_align_ms() / 1e6:6.0f}")
    print(f"{L:<5d} {hip.Context.variant_width(L):<2d} |                           " + " ".join(out), flush=True)
