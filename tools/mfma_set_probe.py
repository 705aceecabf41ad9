#!/usr/bin/env python3
"""Probe: what would the matrix pipe buy the set metrics?

shared = B . B^T over the genomes x phams presence matrix is a dense contraction once B's bits are spread into bytes, and
pocp / af are C . B^T (+ its transpose) with C = gene counts (i8) or summed lengths (base-128 i8 planes).  At ~2 % density a
dense contraction does ~50 x the arithmetic of the bitset popcount, but MFMA is ~30 x faster per operation.  This script
measures the best case without writing a kernel: the LIBRARY int8 GEMM (torch._int_mm -> hipBLASLt, i8 x i8 -> i32, exact)
on the same synth(N, 5000) presence matrix, against the product's popcount kernel (pc_fill_dev jc / gcs).  The GEMM time is
a lower bound for an MFMA formulation: it leaves an N x N int32 matrix in HBM -- both triangles -- that still needs the fp64
epilogue (division, 1 - x, round(., 6)) and the condensed store, which k_set_popc does in the same pass.

    python tools/mfma_set_probe.py [--sizes 2000,5000,20000] [--out profiles/r03/experiments/mfma_set_probe.json]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="2000,5000,20000")
ap.add_argument("--phams", type=int, default=5000)
ap.add_argument("--out", default=None)
a = ap.parse_args()

from phamclust_amd import build, hip
from phamclust_amd.synth import synth_packed

build.build_all()
ctx = hip.Context(0)
dev = torch.device("cuda", 0)
rows = []


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        t.append(e0.elapsed_time(e1))
    return float(np.median(t))


for n in [int(x) for x in a.sizes.split(",")]:
    pk = synth_packed(n, a.phams)
    ctx.upload(pk, residues=False)
    W = pk.words_per_row
    bits = np.unpackbits(pk.bitmap.reshape(n, W).view(np.uint8), axis=1, bitorder="little")      # [N, 64 W] bytes of 0 / 1
    B = torch.from_numpy(bits.astype(np.int8)).to(dev).contiguous()                               # K = 64 W (multiple of 64)
    out = torch.empty(max(pk.n_pairs, 1), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    jc_ms = float(np.median([ctx.fill_dev("jc", True, out.data_ptr(), stream)["ms_total"] for _ in range(6)][1:]))
    row = {"n_genomes": n, "n_phams": pk.n_phams, "k_padded": int(B.shape[1]), "popcount_jc_device_ms": jc_ms,
           "dense_int8_ops": 2 * n * n * int(B.shape[1])}
    Bt = B.t().contiguous()
    try:
        S = torch._int_mm(B, Bt)                                                                  # [N, N] int32, exact
        gemm_ms = timed(lambda: torch._int_mm(B, Bt))
        # exactness against the popcount kernel's integer: shared = round(jc * (nph_s + nph_t) / (1 + jc)) is awkward; use the bitmap on the host for a sample
        rng = np.random.default_rng(n)
        s_idx, t_idx = rng.integers(0, n, 4000), rng.integers(0, n, 4000)
        bm = pk.bitmap.reshape(n, W)
        want = np.array([int(np.unpackbits((bm[s] & bm[t]).view(np.uint8)).sum()) for s, t in zip(s_idx, t_idx)])
        got = S[torch.as_tensor(s_idx, device=dev), torch.as_tensor(t_idx, device=dev)].cpu().numpy()
        row.update({"library": "torch._int_mm (hipBLASLt i8 x i8 -> i32)", "gemm_full_square_ms": gemm_ms,
                    "gemm_Tops": row["dense_int8_ops"] / gemm_ms / 1e9, "gemm_sample_exact": bool(np.array_equal(got, want)),
                    "int32_result_bytes": int(S.numel()) * 4})
        del S
    except Exception as exc:                                                                     # noqa: BLE001
        row["int_mm_error"] = repr(exc)[:300]
    try:                                                                                          # bf16 inputs, fp32 accumulate: exact for 0/1 entries, counts < 2^24
        Bh = B.to(torch.bfloat16)
        Bht = Bh.t().contiguous()
        row["gemm_bf16_full_square_ms"] = timed(lambda: torch.matmul(Bh, Bht))
        row["gemm_bf16_Tflops"] = row["dense_int8_ops"] / row["gemm_bf16_full_square_ms"] / 1e9
        del Bh, Bht
    except Exception as exc:                                                                     # noqa: BLE001
        row["bf16_error"] = repr(exc)[:300]
    best = min(x for x in (row.get("gemm_full_square_ms"), row.get("gemm_bf16_full_square_ms")) if x)
    # what an MFMA route still has to do after the GEMM: read N^2 int32, write N(N-1)/2 fp64 (HBM at ~4 TB/s achieved)
    row["epilogue_traffic_floor_ms"] = (n * n * 4 + pk.n_pairs * 8) / 4e12 * 1e3
    row["mfma_route_lower_bound_ms"] = best + row["epilogue_traffic_floor_ms"]
    row["popcount_over_mfma_lower_bound"] = jc_ms / row["mfma_route_lower_bound_ms"]
    rows.append(row)
    print(json.dumps(row), flush=True)
    del B, Bt, out
    torch.cuda.empty_cache()
doc = {"tool": "tools/mfma_set_probe.py", "rows": rows,
       "decision_rule": "VERDICT r02: keep the popcount kernel unless the matrix pipe shows >= 2 x on jc at N = 20,000 (device time)"}
if a.out:
    with open(a.out, "w") as fh:
        json.dump(doc, fh, indent=1)
        fh.write("\n")
