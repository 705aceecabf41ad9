#!/usr/bin/env python3
"""K4 rate on LONG column genes (strip-mined passes, k_nw_strip) against the general kernel: homolog pairs of L residues, enough of
them to fill the chip.   python tools/long_gene_bench.py [--lens 5000,9000,20000] [--pairs 2048]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from phamclust_amd import build, hip
from phamclust_amd.genome import Genome
from phamclust_amd.pack import pack_genomes

ap = argparse.ArgumentParser()
ap.add_argument("--lens", default="4500,5000,9000,20000")
ap.add_argument("--pairs", type=int, default=2048)
ap.add_argument("--variants", default="0,32,48,64,-1")
ap.add_argument("--check", type=int, default=6, help="pairs checked against the oracle per length")
a = ap.parse_args()
build.build_all()
rng = np.random.default_rng(3)
aa = np.array(list("ACDEFGHIKLMNPQRSTVWY"))
ctx = hip.Context(0)
print("L      pairs | GCUPS by variant (0 = chooser, -1 = general kernel): " + " ".join(f"{v:>7s}" for v in a.variants.split(",")), flush=True)
for L in map(int, a.lens.split(",")):
    n = max(64, min(a.pairs, int(a.pairs * (9000 / L) ** 2) if L > 9000 else a.pairs))
    g, h = Genome("cols"), Genome("rows")
    for i in range(n):
        base = aa[rng.integers(0, 20, L)]
        mut = base.copy(); flip = rng.random(L) < 0.15; mut[flip] = aa[rng.integers(0, 20, int(flip.sum()))]
        g.add(f"c{i:05d}", "".join(base)); h.add(f"r{i:05d}", "".join(mut[: L - int(rng.integers(0, 40))]))
    pk = pack_genomes([g, h])
    ctx.upload(pk)
    rows = np.arange(n, 2 * n, dtype=np.int32); cols = np.arange(n, dtype=np.int32)
    lens = np.diff(pk.seq_off)
    cells = float(np.sum(lens[rows].astype(np.float64) * lens[cols]))
    out = []
    want = None
    for v in map(int, a.variants.split(",")):
        if v == -1 and n * L * L > 3e11:
            out.append("      -"); continue
        ident, diag = ctx.align_pairs(rows, cols, variant=v)
        ident, diag = ctx.align_pairs(rows, cols, variant=v)
        out.append(f"{cells / ctx.last_align_ms() / 1e6:7.0f}")
        if want is None:
            want = (ident, diag)
            if a.check:
                from oracle import oracle as O
                k = rng.choice(n, size=min(a.check, n), replace=False).astype(np.int32)
                _, wi, wd = O.nw_batch(pk.residues, pk.seq_off, rows[k], cols[k])
                assert np.array_equal(ident[k], wi) and np.array_equal(diag[k], wd), "differs from the oracle"
        else:
            assert np.array_equal(ident, want[0]) and np.array_equal(diag, want[1]), f"variant {v} differs from the chooser's"
    print(f"{L:<6d} {n:>5d} |                                                       " + " ".join(out), flush=True)
