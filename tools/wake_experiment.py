"""Where do the ~20 ms go that a fill sometimes takes longer (host clock) right after an upload?  Upload, then one fill with stats
(device ms from HIP events) against the host clock around it; with and without idle time in between."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phamclust_amd import hip
from phamclust_amd.synth import synth_packed
pk = synth_packed(5000, 5000)
ctx = hip.Context(0)
out = torch.empty(pk.n_pairs, dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
ctx.upload(pk); ctx.fill_dev("peq", True, out.data_ptr(), st)
for idle in (0.0, 0.0, 0.0, 0.05, 0.05, 0.2, 0.0, 0.0):
    t0 = time.perf_counter(); ctx.upload(pk); t1 = time.perf_counter()
    if idle: time.sleep(idle)
    t2 = time.perf_counter(); s = ctx.fill_dev("peq", True, out.data_ptr(), st); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"idle {idle:.2f} s: upload {1e3*(t1-t0):6.1f} ms, fill host clock {1e3*(t3-t2):6.1f} ms, device {s['ms_total']:6.1f} ms (plan {s['ms_plan']:.1f} align {s['ms_align']:.1f} reduce {s['ms_reduce']:.1f})", flush=True)
