#!/usr/bin/env python3
"""What the sort-based plan buys on redundant data.  Real phage collections hold many byte-identical proteins
(closely related genomes); the default synthetic set has next to none.  Here a fraction rho of the genes of
synth(N, 5000) is overwritten with the sequence of the first gene of the same (cluster of 40 genomes, pham) group,
and the -m peq fill is timed and checked against the oracle on random pairs."""
import argparse
import dataclasses
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from phamclust_amd import build, hip
from phamclust_amd.synth import synth_packed

ap = argparse.ArgumentParser()
ap.add_argument("-n", type=int, default=5000)
ap.add_argument("--rho", default="0,0.3,0.6,0.9")
ap.add_argument("--check", type=int, default=3000)
ap.add_argument("--world", type=int, default=0, help="also run the shard fills of a WORLD-rank job back to back (cost-balanced deal): "
                "what merging duplicates per rank only costs against the unsharded fill")
a = ap.parse_args()
build.build_all()
base = synth_packed(a.n, 5000)
G = base.n_genes
genome_of = np.repeat(np.arange(base.n_genomes, dtype=np.int64), np.diff(base.gene_off))
group = (genome_of // 40) * (base.n_phams + 1) + base.gene_pham
_, first_of_group = np.unique(group, return_index=True)
canon = first_of_group[np.searchsorted(np.unique(group), group)]            # gene whose sequence the copies take
lens0 = np.diff(base.seq_off)
ctx = hip.Context(0)
for rho in map(float, a.rho.split(",")):
    rng = np.random.default_rng(7)
    src = np.where(rng.random(G) < rho, canon, np.arange(G))
    lens = lens0[src]
    seq_off = np.zeros(G + 1, dtype=np.int64); np.cumsum(lens, out=seq_off[1:])
    idx = np.repeat(base.seq_off[:-1][src] - seq_off[:-1], lens) + np.arange(seq_off[-1])
    residues = np.ascontiguousarray(base.residues[idx])
    tlen = np.zeros(base.n_genomes, dtype=np.int64); np.add.at(tlen, genome_of, lens)
    pk = dataclasses.replace(base, seq_off=seq_off, residues=residues, tlen=tlen).validate()
    ctx.upload(pk)
    ctx.fill("peq")
    out, st = ctx.fill("peq", True, want_stats=True)
    line = (f"rho {rho:.1f}: {st['ms_total']:.1f} ms/fill = {pk.n_pairs / st['ms_total'] * 1e3:.3e} pairs/s | alignments {st['n_alignments']} "
            f"distinct {st['n_distinct_alignments']} ({st['n_distinct_alignments'] / max(st['n_alignments'], 1):.3f}) | cells {st['n_cells']:.3e} "
            f"computed {st['n_distinct_cells']:.3e} | plan {st['ms_plan']:.1f} align {st['ms_align']:.1f} reduce {st['ms_reduce']:.1f} ms")
    if a.check:
        from oracle import oracle as O
        n = pk.n_genomes
        r2 = np.random.default_rng(1)
        s_idx, t_idx = r2.integers(0, n - 1, a.check), r2.integers(0, n, a.check)
        lo, hi = np.minimum(s_idx, t_idx), np.maximum(s_idx, t_idx)
        keep = lo < hi; lo, hi = lo[keep], hi[keep]
        want = O.pairs(pk, "peq", lo, hi, as_distance=True)
        line += f" | {lo.size} random pairs vs oracle: {'equal' if np.array_equal(out[lo * n - lo * (lo + 1) // 2 + (hi - lo - 1)], want) else 'MISMATCH'}"
    print(line, flush=True)
    if a.world > 1:
        import torch
        stream = torch.cuda.current_stream().cuda_stream
        ms, distinct, cells = [], 0, 0.0
        for r in range(a.world):
            ctx.set_shard(r, a.world, balanced=True)
            buf = torch.empty(ctx.shard_stride(), dtype=torch.float64, device="cuda")
            ctx.fill_shard_dev("peq", True, buf.data_ptr(), stream)
            s_ = ctx.fill_shard_dev("peq", True, buf.data_ptr(), stream); torch.cuda.synchronize()
            ms.append(s_["ms_total"]); distinct += s_["n_distinct_alignments"]; cells += s_["n_distinct_cells"]
        ctx.set_shard(0, 1)
        # the alignment-sliced route: every rank plans the whole fill, aligns its slice; the root reduces after one collective
        plan = ctx.plan_dev("peq", stream); plan = ctx.plan_dev("peq", stream)
        n = plan["n_distinct_alignments"]
        res = torch.zeros(max(n, 1), dtype=torch.int64, device="cuda"); acc = torch.zeros_like(res)
        sl = []
        for r in range(a.world):
            ctx.align_slice_dev(r, a.world, res.data_ptr(), stream)
            s_ = ctx.align_slice_dev(r, a.world, res.data_ptr(), stream); torch.cuda.synchronize()
            sl.append(s_["ms_align"]); acc += res
        outm = torch.empty(pk.n_pairs, dtype=torch.float64, device="cuda")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ctx.reduce_dev("peq", True, acc.data_ptr(), outm.data_ptr(), stream); e1.record(); torch.cuda.synchronize()
        same = bool(np.array_equal(outm.cpu().numpy(), out))
        print(f"         {a.world} ranks, alignments sliced: plan {plan['ms_plan']:.1f} ms (every rank) + slowest slice {max(sl):.1f} ms (mean {sum(sl) / len(sl):.1f}) "
              f"+ reduce of {n * 8 / 1e6:.0f} MB of results + matrix on the root {e0.elapsed_time(e1):.1f} ms; == unsharded fill: {same}", flush=True)
        print(f"         {a.world} ranks: slowest {max(ms):.1f} ms, sum {sum(ms):.1f} ms ({sum(ms) / st['ms_total']:.2f} x the unsharded fill); alignments computed over all ranks "
              f"{distinct} = {distinct / max(st['n_distinct_alignments'], 1):.2f} x the unsharded count", flush=True)
