#!/usr/bin/env python3
"""Calibration (build container only): the LIVE reference's matrix_de_novo vs this repo's oracle on identical input,
so that the oracle-based cpu_baseline measured on the GPU box can be related to the true reference.
Set metrics only: aai/peq need parasail, which is absent here."""
import json
import os
import sys
import time
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True
from oracle import oracle as O
from phamclust_amd.synth import synth_packed
from phamclust_amd.pack import unpack_genomes

stub = types.ModuleType("parasail"); stub.blosum62 = None; stub.nw_trace_diag_16 = None
sys.modules["parasail"] = stub                      # set metrics never touch it
sys.path.insert(0, "/root/reference/src")
from phamclust.cli import METRICS
from phamclust.genome import Genome as RefGenome
from phamclust.matrix import matrix_de_novo

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
packed = synth_packed(N, 5000)
ours = unpack_genomes(packed)
ref = []
for g in ours:
    r = RefGenome(g.name)
    for pham, translations in g.phams.items():
        for t in translations:
            r.add(pham, t)
    ref.append(r)
pairs = N * (N - 1) // 2
out = {"n_genomes": N, "pairs": pairs, "host": f"{os.cpu_count()} vCPU build container", "rows": []}
for metric in ("gcs", "jc", "pocp", "af"):
    t0 = time.perf_counter(); m = matrix_de_novo(ref, METRICS[metric], 1); t_ref = time.perf_counter() - t0
    t0 = time.perf_counter(); v1 = O.fill(packed, metric, True, nthreads=1); t_o1 = time.perf_counter() - t0
    t0 = time.perf_counter(); v8 = O.fill(packed, metric, True, nthreads=os.cpu_count()); t_o8 = time.perf_counter() - t0
    import numpy as np
    same = bool(np.array_equal(np.array([w for s, t, w in m if s != t]), v1))
    out["rows"].append({"metric": metric, "reference_t1_pairs_per_s": pairs / t_ref, "oracle_t1_pairs_per_s": pairs / t_o1,
                        f"oracle_t{os.cpu_count()}_pairs_per_s": pairs / t_o8, "oracle_over_reference_t1": t_ref / t_o1,
                        "values_identical": same})
    print(out["rows"][-1], flush=True)
json.dump(out, open(os.path.join(REPO, "profiles", "calibration_reference_vs_oracle.json"), "w"), indent=1)
