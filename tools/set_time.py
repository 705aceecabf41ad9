#!/usr/bin/env python3
"""Device time of set-metric fills, one line per (variant, metric, N): min and median of `--steps` fills (HIP events inside the
library), which kernel family ran, and a sampled oracle check.  `--variants a,b` runs each csrc/libphamclust_hip_<name>.so
(tools/build_variant.py) in a child process of its own (a process binds one library), "base" = the release library.

    python tools/set_time.py --sizes 2000,20000 --metrics jc,pocp,af --variants base,noepi [--env PC_SET_KERNEL=sparse64]
"""
import argparse
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="2000,20000")
ap.add_argument("--metrics", default="jc,pocp,af")
ap.add_argument("--phams", type=int, default=5000)
ap.add_argument("--steps", type=int, default=9)
ap.add_argument("--check", type=int, default=4000)
ap.add_argument("--variants", default=None)
ap.add_argument("--env", action="append", default=[])
a = ap.parse_args()

if a.variants is not None:
    for v in a.variants.split(","):
        env = dict(os.environ)
        env.pop("PHAMCLUST_NATIVE_VARIANT", None)
        if v != "base":
            env["PHAMCLUST_NATIVE_VARIANT"] = v
        for kv in a.env:
            k, _, val = kv.partition("=")
            env[k] = val
        argv = [sys.executable, os.path.abspath(__file__), "--sizes", a.sizes, "--metrics", a.metrics, "--phams", str(a.phams), "--steps", str(a.steps), "--check", str(a.check)]
        rc = subprocess.call(argv, env=env)
        if rc:
            print(json.dumps({"variant": v, "failed": rc}), flush=True)
    sys.exit(0)

import numpy as np
import torch
from phamclust_amd import hip
from phamclust_amd.synth import synth_packed

variant = os.environ.get("PHAMCLUST_NATIVE_VARIANT", "base")
ctx = hip.Context(0)
stream = torch.cuda.current_stream().cuda_stream
for n in [int(x) for x in a.sizes.split(",")]:
    pk = synth_packed(n, a.phams)
    ctx.upload(pk, residues=False)
    out = torch.empty(max(pk.n_pairs, 1), dtype=torch.float64, device="cuda")
    for metric in a.metrics.split(","):
        ms = []
        for _ in range(a.steps + 1):
            torch.cuda.synchronize()
            ms.append(ctx.fill_dev(metric, True, out.data_ptr(), stream)["ms_total"])
        torch.cuda.synchronize()
        ms = ms[1:]
        ok = None
        if a.check:
            from oracle import oracle as O
            rng = np.random.default_rng(n)
            s_idx, t_idx = rng.integers(0, n, a.check), rng.integers(0, n, a.check)
            lo, hi = np.minimum(s_idx, t_idx), np.maximum(s_idx, t_idx)
            keep = lo < hi; lo, hi = lo[keep], hi[keep]
            cond = lo * n - lo * (lo + 1) // 2 + (hi - lo - 1)
            got = out[torch.as_tensor(cond, device="cuda")].cpu().numpy()
            ok = bool(np.array_equal(got, O.pairs(pk, metric, lo, hi, True)))
        algo = pk.n_genomes * pk.words_per_row * 8 + 16 * pk.n_genomes + 8 * pk.n_pairs
        print(json.dumps({"variant": variant, "metric": metric, "n": n, "kernel": ctx.last_set_kernel(), "ms_min": round(min(ms), 4),
                          "ms_median": round(float(np.median(ms)), 4), "TBps_algorithmic": round(algo / float(np.median(ms)) / 1e9, 3),
                          "oracle_sample_equal": ok, "env": {k: os.environ[k] for k in os.environ if k.startswith("PC_S")}}), flush=True)
