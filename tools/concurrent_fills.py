#!/usr/bin/env python3
"""Is a small fill short of work or short of schedule?  Two contexts on ONE GPU fill the same matrix at the same time (two host
threads, two streams): if both finish in less than twice the time of one, a single fill leaves the chip partly idle (launch
tails, ramps) and scheduling could recover it; if it takes twice as long, the time is instruction issue and nothing is idle.
    python tools/concurrent_fills.py [n_genomes] [metric]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phamclust_amd import hip
from phamclust_amd.synth import synth_packed

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
metric = sys.argv[2] if len(sys.argv) > 2 else "peq"
pk = synth_packed(n, 5000)
ctxs = [hip.Context(0), hip.Context(0)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
outs = [torch.empty(pk.n_pairs, dtype=torch.float64, device="cuda") for _ in ctxs]
for c in ctxs:
    c.upload(pk)
    c.fill(metric)


def run(i, reps, times):
    for _ in range(reps):
        t0 = time.perf_counter()
        ctxs[i].fill_dev(metric, True, outs[i].data_ptr(), streams[i].cuda_stream)
        streams[i].synchronize()
        times.append((time.perf_counter() - t0) * 1e3)


solo = []
run(0, 4, solo)
both = [[], []]
torch.cuda.synchronize()
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(i, 4, both[i])) for i in range(2)]
for t in th: t.start()
for t in th: t.join()
wall = (time.perf_counter() - t0) * 1e3
print(f"synth({n},5000) {metric}: one fill alone {min(solo):.1f} ms (host clock); two contexts x 4 fills at once: {wall:.1f} ms wall = "
      f"{wall / 8:.1f} ms per fill ({wall / 8 / min(solo):.3f} x); results equal: {bool(torch.equal(outs[0], outs[1]))}")
