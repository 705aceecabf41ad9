#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace / stats / PMC passes) into one small JSON for profiles/.

usage: summarize_profile.py --trace DIR [--fetch DIR] [--write DIR] --fills N --out FILE
"""
import argparse
import collections
import csv
import glob
import json


def one(d, pat):
    f = glob.glob(f"{d}/**/*{pat}", recursive=True)
    return f[0] if f else None


ap = argparse.ArgumentParser()
ap.add_argument("--trace", required=True)
ap.add_argument("--fetch")
ap.add_argument("--write")
ap.add_argument("--fills", type=int, required=True, help="matrix fills in the traced run (warmup + steps)")
ap.add_argument("--out", required=True)
ap.add_argument("--note", default="")
a = ap.parse_args()

rows = list(csv.DictReader(open(one(a.trace, "_kernel_trace.csv"))))
per = collections.defaultdict(lambda: {"calls": 0, "total_ms": 0.0})
nw = []
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    per[name]["calls"] += 1
    per[name]["total_ms"] += ms
    if "k_nw" in name:
        nw.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for v in per.values():
    v["avg_ms"] = v["total_ms"] / v["calls"]
# alignment launches of one fill overlap on several streams: group launches into fills by gaps > 1 ms between them
nw.sort()
spans, cur_s, cur_e = [], None, None
for s, e in nw:
    if cur_s is None or s > cur_e + 1_000_000:
        if cur_s is not None:
            spans.append((cur_e - cur_s) / 1e6)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
if cur_s is not None:
    spans.append((cur_e - cur_s) / 1e6)
out = {"note": a.note, "fills": a.fills, "kernels": dict(sorted(per.items(), key=lambda kv: -kv[1]["total_ms"])),
       "k_nw_systolic": {"launches_per_fill": len(nw) / max(a.fills, 1), "sum_of_durations_ms_per_fill": sum(e - s for s, e in nw) / 1e6 / max(a.fills, 1),
                         "span_ms_per_fill": spans, "comment": "launches of different column-gene classes overlap on several streams; span = first start to last end "
                                                               "of one fill's launches = what bench.py times with HIP events (roofline.ms_kernels_per_fill)"}}


def pmc(d, counter):
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(one(d, "_counter_collection.csv"))):
        if r["Counter_Name"] == counter:
            agg["k_nw_systolic" if "k_nw" in r["Kernel_Name"] else r["Kernel_Name"].split("(")[0].replace("void ", "")] += float(r["Counter_Value"])
    return agg


if a.fetch and a.write:
    f, w = pmc(a.fetch, "FETCH_SIZE"), pmc(a.write, "WRITE_SIZE")
    out["hbm_traffic_one_fill"] = {
        "FETCH_SIZE_KB": dict(f), "WRITE_SIZE_KB": dict(w),
        "k_nw_systolic_bytes": {"fetch_raw": f["k_nw_systolic"] * 1024, "fetch_corrected_x2": 2 * f["k_nw_systolic"] * 1024,
                                "write": w["k_nw_systolic"] * 1024, "total_corrected": (2 * f["k_nw_systolic"] + w["k_nw_systolic"]) * 1024},
        "comment": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of one fill; gfx950 FETCH_SIZE counts 64 B per 128-B request, so it is doubled "
                   "as MI355X_MICROARCH.md prescribes; narrow (byte/dword) gathers are not a calibrated pattern, treat as +-2x"}
json.dump(out, open(a.out, "w"), indent=1)
print(json.dumps(out["k_nw_systolic"], indent=1))
if "hbm_traffic_one_fill" in out:
    print(json.dumps(out["hbm_traffic_one_fill"]["k_nw_systolic_bytes"], indent=1))
