#!/usr/bin/env python3
"""Wall time of the whole pipeline (TSV in, clusters out) on a synthetic collection, stage by stage as the
pipeline logs them; shows what is left on the host once the matrix fill runs on the GPU."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phamclust_amd import build
from phamclust_amd.synth import synth_packed, write_tsv_packed

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
metric = sys.argv[2] if len(sys.argv) > 2 else "peq"
build.build_all()
with tempfile.TemporaryDirectory() as tmp:
    tsv = os.path.join(tmp, "in.tsv")
    t0 = time.perf_counter(); write_tsv_packed(synth_packed(n, 5000), tsv); t1 = time.perf_counter()
    print(f"wrote {os.path.getsize(tsv) / 1e6:.0f} MB TSV in {t1 - t0:.1f} s", flush=True)
    out = os.path.join(tmp, "out"); os.makedirs(out)
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, "-m", "phamclust_amd", tsv, out, "-m", metric, "-d"], capture_output=True, text=True)
    t1 = time.perf_counter()
    import re
    print("\n".join(l for l in p.stdout.splitlines() if not re.match(r"^cluster \d+: ", l))[-6000:]); print(p.stderr[-3000:])
    print(f"pipeline exit {p.returncode}: {t1 - t0:.1f} s wall for {n} genomes, -m {metric}", flush=True)
    n_clusters = len([d for d in os.listdir(out) if d.startswith("cluster_")])
    print("cluster directories:", n_clusters)
