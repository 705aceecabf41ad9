#!/usr/bin/env python3
"""Timeline of the K4 launches of the last fill in a rocprofv3 kernel trace (launches on their eight streams, not serialised):
start/end offsets per launch, the launches in flight over time, and what ends last.
    python tools/fill_timeline.py gpurun_out/TAG/t_kernel_trace.csv [fills_in_trace]"""
import csv, re, sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"k_nw_systolic<(\d+), (\d+)(, (true|false))?>", r["Kernel_Name"])
    if m:
        wg = int(r["Workgroup_Size_X"])
        rows.append(dict(s=int(r["Start_Timestamp"]), e=int(r["End_Timestamp"]), W=int(m.group(1)), tasks=int(r["Grid_Size_X"]) // wg,
                         waves=wg // 64, cell={"true": "profile", "false": "compare", None: "-"}[m.group(4)], lds=int(r.get("LDS_Block_Size", 0) or 0)))
rows.sort(key=lambda r: r["s"])
fills = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if not fills:                                   # fills are separated by gaps without any K4 launch in flight
    groups, cur, end = [], [], 0
    for r in rows:
        if cur and r["s"] > end + 200_000: groups.append(cur); cur = []
        cur.append(r); end = max(end, r["e"])
    groups.append(cur)
else:
    per = len(rows) // fills; groups = [rows[i * per:(i + 1) * per] for i in range(fills)]
g = groups[-1]
t0 = g[0]["s"]; t1 = max(r["e"] for r in g)
print(f"{len(groups)} fills in the trace; last: {len(g)} launches, {(t1 - t0) / 1e6:.2f} ms from first start to last end; sum of durations {sum(r['e'] - r['s'] for r in g) / 1e6:.1f} ms")
print("start_ms  end_ms   dur_ms  W  tasks waves cell     lds_B")
for r in g:
    print(f"{(r['s'] - t0) / 1e6:8.2f} {(r['e'] - t0) / 1e6:8.2f} {(r['e'] - r['s']) / 1e6:7.2f} {r['W']:>3d} {r['tasks']:>6d} {r['waves']:>3d}   {r['cell']:<8s} {r['lds']:>6d}")
nb = 40
print("launches in flight (40 bins over the fill):", " ".join(str(sum(1 for r in g if r["s"] <= t0 + (t1 - t0) * (b + 0.5) / nb < r["e"])) for b in range(nb)))
print("last to end:")
for r in sorted(g, key=lambda r: -r["e"])[:8]:
    print(f"  ends {(t1 - r['e']) / 1e6:6.2f} ms before the fill's end: W {r['W']} tasks {r['tasks']} waves {r['waves']} {r['cell']} started {(r['s'] - t0) / 1e6:.2f} ran {(r['e'] - r['s']) / 1e6:.2f} ms")
