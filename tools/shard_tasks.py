import os, sys
sys.path.insert(0, "/root/repo")
import torch
from phamclust_amd import hip
from phamclust_amd.synth import synth_packed
pk = synth_packed(5000, 5000)
ctx = hip.Context(0); ctx.upload(pk)
stream = torch.cuda.current_stream().cuda_stream
full = torch.empty(pk.n_pairs, dtype=torch.float64, device="cuda")
st = ctx.fill_dev("peq", True, full.data_ptr(), stream); torch.cuda.synchronize()
print("unsharded", {k: st[k] for k in ("n_chunks", "n_tasks", "n_alignments", "n_distinct_alignments", "n_cells", "n_distinct_cells", "ms_align", "n_align_launches")})
for world, bal in ((1, False), (2, False), (8, False), (8, True)):
    ctx.set_shard(0, world, balanced=bal)
    buf = torch.empty(ctx.shard_stride(), dtype=torch.float64, device="cuda")
    ctx.fill_shard_dev("peq", True, buf.data_ptr(), stream)
    s = ctx.fill_shard_dev("peq", True, buf.data_ptr(), stream); torch.cuda.synchronize()
    print("world", world, "balanced", bal, {k: s[k] for k in ("n_chunks", "n_tasks", "n_alignments", "n_distinct_alignments", "n_cells", "n_distinct_cells", "ms_align", "n_align_launches")},
          "GCUPS", s["n_distinct_cells"] / s["ms_align"] / 1e6)
