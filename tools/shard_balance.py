#!/usr/bin/env python3
"""Rehearse the 8-rank shard on ONE GPU: every rank's shard-local fill run back to back, per-rank work
(pairs, alignments, DP cells, device ms), then the hand-gathered shards assembled and compared with the
unsharded fill.  Shows how well the boustrophedon deal balances the ranks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from phamclust_amd import hip
from phamclust_amd.synth import synth_packed

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
metric = sys.argv[3] if len(sys.argv) > 3 else "peq"
balanced = len(sys.argv) > 4 and sys.argv[4] == "balanced"      # pc_set_shard_balanced instead of the boustrophedon deal
pk = synth_packed(n, 5000)
ctx = hip.Context(0); ctx.upload(pk)
stream = torch.cuda.current_stream().cuda_stream
full = torch.empty(pk.n_pairs, dtype=torch.float64, device="cuda")
st = ctx.fill_dev(metric, True, full.data_ptr(), stream); torch.cuda.synchronize()
print(f"unsharded: {st['ms_total']:.1f} ms, {st['n_cells']:.3e} cells", flush=True)
parts, rows = [], []
for r in range(world):
    ctx.set_shard(r, world, balanced=balanced)
    buf = torch.empty(ctx.shard_stride(), dtype=torch.float64, device="cuda")
    ctx.fill_shard_dev(metric, True, buf.data_ptr(), stream)            # warm
    s = ctx.fill_shard_dev(metric, True, buf.data_ptr(), stream); torch.cuda.synchronize()
    parts.append(buf); rows.append(s)
    print(f"rank {r}: pairs {s['n_pairs']} aln {s['n_alignments']} cells {s['n_cells']:.4e} ms {s['ms_total']:.1f} (plan {s['ms_plan']:.1f} align {s['ms_align']:.1f})", flush=True)
out = torch.empty(pk.n_pairs, dtype=torch.float64, device="cuda")
gathered = torch.cat(parts)
ctx.assemble_dev(gathered.data_ptr(), world, out.data_ptr(), stream); torch.cuda.synchronize()
cells = np.array([s["n_cells"] for s in rows], float); ms = np.array([s["ms_total"] for s in rows])
print(f"assembled == unsharded: {bool(torch.equal(out, full))}")
print(f"cells max/mean {cells.max() / cells.mean():.4f}; ms max/mean {ms.max() / ms.mean():.4f}; sum of rank ms {ms.sum():.1f} vs unsharded {st['ms_total']:.1f}; "
      f"ideal {world}-GPU step ~ {ms.max():.1f} ms -> {pk.n_pairs / ms.max() * 1e3:.3e} pairs/s")
