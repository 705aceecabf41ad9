#!/usr/bin/env python3
"""The K4 (k_nw_*) launches of the last-but-one fill in a rocprofv3 kernel trace on a timeline: start / end / duration (ms), workgroups x waves,
kernel -- launches longer than 1.5 ms and everything in the fill's last 3 ms -- and the launches in flight over the fill (40 bins).
    python tools/k4_timeline.py <dir>/t_kernel_trace.csv"""
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
nw=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0].replace("void ",""),int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"]),int(r["Workgroup_Size_X"])//64) for r in rows if "k_nw" in r["Kernel_Name"])
fills=[];cur=None
for x in nw:
    if cur is None or x[0]>max(y[1] for y in cur)+500_000:
        if cur: fills.append(cur)
        cur=[]
    cur.append(x)
fills.append(cur)
print(len(fills),"fills")
for g in fills[-2:-1]:
    t0=g[0][0]; t1=max(x[1] for x in g)
    print("fill:",len(g),"launches", (t1-t0)/1e6,"ms")
    for s,e,n,wg,wv in g:
        if (e-s)/1e6>1.5 or (t1-e)/1e6<3: print(f"{(s-t0)/1e6:8.2f} {(e-t0)/1e6:8.2f} {(e-s)/1e6:8.2f}  {wg:7d} wgs x {wv} waves  {n}")
    nb=40
    print("in flight:", " ".join(str(sum(1 for x in g if x[0] <= t0+(t1-t0)*(b+0.5)/nb < x[1])) for b in range(nb)))
