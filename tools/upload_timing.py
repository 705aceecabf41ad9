import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ["PC_UPLOAD_TIMING"] = "1"
from phamclust_amd import hip
from phamclust_amd.synth import synth_packed
for n in (5000,):
    pk = synth_packed(n, 5000)
    ctx = hip.Context(0)
    for i in range(3):
        t0 = time.perf_counter(); ctx.upload(pk); print("upload total ms", (time.perf_counter() - t0) * 1e3, flush=True)
