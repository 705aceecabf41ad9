#!/usr/bin/env python3
"""Two peq fills of synth_real(N) and nothing else (run it under `rocprofv3 --kernel-trace`), or, with --summarise DIR, the K4
launches of the LAST fill in that trace: kernel, workgroups, waves per workgroup, start and duration -- where the time of the
real-collection-shaped fill goes (its long genes run on k_nw_strip).
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/x -o t -- python3 tools/real_trace.py -n 5000
  python3 tools/real_trace.py --summarise gpurun_out/x"""
import argparse, csv, glob, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("-n", type=int, default=5000)
ap.add_argument("--metric", default="peq")
ap.add_argument("--summarise", default="")
ap.add_argument("--synth", type=int, default=0, help="phams of the benchmark's synth(n, phams) instead of synth_real(n)")
a = ap.parse_args()
if a.summarise:
    path = sorted(glob.glob(os.path.join(a.summarise, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = [r for r in csv.DictReader(open(path))]
    k4 = [r for r in rows if re.search(r"k_nw_(systolic|strip|general)", r["Kernel_Name"])]
    k4.sort(key=lambda r: int(r["Start_Timestamp"]))
    # fills are separated by gaps of more than 5 ms between K4 launches
    fills, cur = [], []
    for r in k4:
        if cur and int(r["Start_Timestamp"]) - max(int(x["End_Timestamp"]) for x in cur) > 5e6: fills.append(cur); cur = []
        cur.append(r)
    fills.append(cur)
    last = fills[-1]
    t0 = min(int(r["Start_Timestamp"]) for r in last); t1 = max(int(r["End_Timestamp"]) for r in last)
    print(f"{path}: {len(fills)} fills; last: {len(last)} K4 launches over {(t1 - t0) / 1e6:.2f} ms")
    print(f"{'kernel':44s} {'wgs':>7s} {'waves':>5s} {'lds':>6s} {'start ms':>9s} {'ms':>8s}")
    for r in last:
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        wg = int(r["Workgroup_Size_X"] if "Workgroup_Size_X" in r else r["Workgroup_Size"]); grid = int(r["Grid_Size_X"] if "Grid_Size_X" in r else r["Grid_Size"])
        print(f"{name:44s} {grid // wg:7d} {wg // 64:5d} {int(r.get('LDS_Block_Size', r.get('LDS_Block_Size_v', 0)) or 0):6d} "
              f"{(int(r['Start_Timestamp']) - t0) / 1e6:9.2f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:8.2f}")
    sys.exit(0)
import torch
from phamclust_amd import build, hip
from phamclust_amd.synth import synth_real, synth_packed
build.build_all()
pk = synth_packed(a.n, a.synth) if a.synth else synth_real(a.n)
ctx = hip.Context(0); ctx.upload(pk)
stream = torch.cuda.current_stream().cuda_stream
out_dev = torch.empty(pk.n_pairs, dtype=torch.float64, device="cuda")
for _ in range(2):
    st = ctx.fill_dev(a.metric, True, out_dev.data_ptr(), stream); torch.cuda.synchronize()
    print({k: st[k] for k in ("ms_total", "ms_plan", "ms_align", "ms_reduce", "n_tasks", "n_align_launches", "n_distinct_alignments", "n_distinct_cells")}, flush=True)
