"""Per-phase times of pc_upload (PC_UPLOAD_TIMING=1 makes the library print them) at two sizes; third upload of each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PC_UPLOAD_TIMING"] = "1"
from phamclust_amd import hip
from phamclust_amd.synth import synth_packed
for n in (2000, 5000):
    pk = synth_packed(n, 5000)
    ctx = hip.Context(0)
    for i in range(6):
        print(f"--- N={n} upload {i}", file=sys.stderr, flush=True)
        t0 = time.perf_counter(); ctx.upload(pk); print(f"N={n} upload total ms {(time.perf_counter() - t0) * 1e3:.2f}", file=sys.stderr, flush=True)
    t0 = time.perf_counter(); ctx.upload(pk, residues=False); print(f"N={n} upload (sets only) total ms {(time.perf_counter() - t0) * 1e3:.2f}", file=sys.stderr, flush=True)
    ctx.close()
