#!/usr/bin/env python3
"""Per-launch-class durations of the K4 launches of one fill from rocprofv3 kernel traces taken with the launches
serialised (tools/trace_serial.sh), one build or two side by side:
    python tools/class_times.py gpurun_out/new_serial/t_kernel_trace.csv [gpurun_out/old_serial/t_kernel_trace.csv]
Launches are matched by variant W and order of launch (the class order is the same in both builds)."""
import csv, re, sys


def load(path):
    rows = []
    for r in csv.DictReader(open(path)):
        m = re.search(r"k_nw_systolic<(\d+), (\d+)(, (true|false))?>", r["Kernel_Name"])
        if m:
            wg = int(r["Workgroup_Size_X"])
            rows.append((int(r["Start_Timestamp"]), int(m.group(1)), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6,
                         int(r["Grid_Size_X"]) // wg, wg // 64, {"true": "profile", "false": "compare", None: "-"}[m.group(4)],
                         int(r.get("LDS_Block_Size", 0) or 0)))
    rows.sort()
    n = len(rows)
    per_fill = n // 2 if n % 2 == 0 else n                   # trace_serial.sh runs one warm-up fill and one timed fill ... plus bench's wall fills
    for k in (4, 3, 2, 1):
        if n % k == 0: per_fill = n // k; break
    return rows[-per_fill:]


new = load(sys.argv[1])
old = load(sys.argv[2]) if len(sys.argv) > 2 else None
used = set()
print("W   tasks waves cell      lds_B   ms" + ("   | ms_old ratio" if old else ""))
tot_n = tot_o = 0.0
for r in sorted(new, key=lambda r: (r[1], r[0])):
    line = f"{r[1]:<3d} {r[3]:>6d} {r[4]:>4d} {r[5]:<8s} {r[6]:>6d} {r[2]:7.2f}"
    tot_n += r[2]
    if old:
        c = [(i, o) for i, o in enumerate(old) if o[1] == r[1] and i not in used]
        if c:
            i, o = c[0]; used.add(i); tot_o += o[2]
            line += f"   | {o[2]:7.2f} {r[2] / o[2]:.3f}"
    print(line)
print(f"sum {tot_n:.1f} ms" + (f"   | {tot_o:.1f} ms  {tot_n / tot_o:.3f}" if old else ""))
