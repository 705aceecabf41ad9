#!/usr/bin/env python3
"""A/B of the alignment-launch stream count in ONE process (devices differ by several %, so only
interleaved rounds on one device are comparable)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phamclust_amd import hip
from phamclust_amd.synth import synth_packed
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
pk = synth_packed(n, 5000)
ctxs = {}
for k in (1, 2, 4, 8):
    os.environ["PC_ALIGN_STREAMS"] = str(k)
    ctxs[k] = hip.Context(0); ctxs[k].upload(pk)
for k, c in ctxs.items():
    c.fill("peq")
res = {k: [] for k in ctxs}
for rnd in range(4):
    for k, c in ctxs.items():
        _, st = c.fill("peq", want_stats=True)
        res[k].append(st["ms_align"])
for k, v in res.items():
    print(f"streams={k}: align ms {['%.1f' % x for x in v]} median {sorted(v)[len(v)//2]:.1f}", flush=True)
