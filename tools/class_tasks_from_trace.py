#!/usr/bin/env python3
"""Tasks per K4 launch (variant, cell, waves) of every fill in a rocprofv3 kernel trace, fills side by side (fills are
separated by gaps): which launch classes a sharded fill inflates."""
import csv, re, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"k_nw_systolic<(\d+), (\d+)(, (true|false))?>", r["Kernel_Name"])
    if m:
        wg = int(r["Workgroup_Size_X"])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(m.group(1)), m.group(4) or "-", wg // 64, int(r["Grid_Size_X"]) // wg, int(r.get("LDS_Block_Size", 0) or 0)))
rows.sort()
groups, cur, end = [], [], 0
for r in rows:
    if cur and r[0] > end + 200_000: groups.append(cur); cur = []
    cur.append(r); end = max(end, r[1])
groups.append(cur)
print(len(groups), "fills:", [len(g) for g in groups], "tasks:", [sum(r[5] for r in g) for g in groups])
sel = [int(x) for x in sys.argv[2:]] or list(range(len(groups)))
tabs = []
for gi in sel:
    t = collections.OrderedDict()
    for r in sorted(groups[gi], key=lambda r: (r[2], r[3], r[4], r[6])):
        t[(r[2], r[3], r[4], r[6])] = t.get((r[2], r[3], r[4], r[6]), 0) + r[5]
    tabs.append(t)
keys = sorted(set(k for t in tabs for k in t))
print("W cell waves lds | tasks per selected fill")
for k in keys:
    print(f"{k[0]:>3d} {k[1]:<6s} {k[2]:>2d} {k[3]:>6d} | " + " ".join(f"{t.get(k, 0):>8d}" for t in tabs))
