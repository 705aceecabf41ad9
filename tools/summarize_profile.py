#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace / stats / PMC passes) into one small JSON for profiles/.

usage: summarize_profile.py --trace DIR [--fetch DIR] [--write DIR] --fills N --out FILE
       summarize_profile.py k4span TRACE_DIR OUT.txt [BENCH.json]

k4span: the K4 (alignment) launches of every fill in a rocprofv3 --kernel-trace of `bench.py` -- launches, first start -> last end
(the SPAN: what bench.py's HIP events around the alignment stage see, `roofline.ms_kernels_per_fill`), and the SUM of the launches'
own durations (they overlap on several streams, so the sum is 2-3 x the span: it is what --stats' per-kernel totals add up to, and
NOT the time the roofline divides by).  With the bench line: DP cells / span = TCUPS, against the 4.915-TCUPS VALU ceiling.  The
text file goes to profiles/<round>/final/bench_peq5000_k4_span.txt, so that `frac` can be redone from tracked files alone.
"""
import argparse
import collections
import csv
import glob
import json


def one(d, pat):
    f = glob.glob(f"{d}/**/*{pat}", recursive=True)
    return f[0] if f else None


def k4_fills(trace_dir):
    """[(first start ns, last end ns, launches, summed ns, {kernel: summed ns})] per fill: K4 launches grouped by gaps > 1 ms."""
    rows = list(csv.DictReader(open(one(trace_dir, "_kernel_trace.csv"))))
    nw = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in rows if "k_nw" in r["Kernel_Name"])
    fills, cur = [], None
    for s, e, name in nw:
        if cur is None or s > cur[1] + 1_000_000:
            if cur is not None:
                fills.append(cur)
            cur = [s, e, 0, 0, collections.defaultdict(int)]
        cur[1] = max(cur[1], e); cur[2] += 1; cur[3] += e - s; cur[4][name] += e - s
    if cur is not None:
        fills.append(cur)
    return fills


def k4span(trace_dir, out_path, bench_path=None):
    import sys
    fills = k4_fills(trace_dir)
    line = None
    if bench_path:
        line = json.loads([l for l in open(bench_path).read().splitlines() if l.startswith("{")][-1])
    lines = [f"# K4 (k_nw_*) launches per fill in the rocprofv3 kernel trace under {trace_dir} (tools/summarize_profile.py k4span)",
             "# span = first K4 start -> last K4 end of a fill: the time `roofline` divides by; sum = the launches' own durations added up (they overlap on",
             "# several streams): what the per-kernel totals of *_kernel_stats.csv add up to", f"{'fill':>4s} {'launches':>8s} {'span_ms':>10s} {'sum_ms':>10s} {'sum/span':>8s}"]
    for k, (s, e, n, tot, per) in enumerate(fills):
        lines.append(f"{k:4d} {n:8d} {(e - s) / 1e6:10.3f} {tot / 1e6:10.3f} {tot / max(e - s, 1):8.2f}")
    spans = [(e - s) / 1e6 for s, e, *_ in fills]
    if line and line.get("kernel_source_hash"):
        lines.insert(1, f"# device code of the traced library (bench.py kernel_source_hash): {line['kernel_source_hash']}")
    if spans:
        steady = spans[1:] if len(spans) > 1 else spans          # the first fill of a process pays code-object loads
        mean = sum(steady) / len(steady)
        lines.append(f"span, mean over fills 1..{len(spans) - 1}: {mean:.3f} ms (min {min(steady):.3f}, max {max(steady):.3f}); first fill {spans[0]:.3f} ms")
        if line and "roofline" in line and "dp_cells_computed" in line["roofline"]:
            cells = line["roofline"]["dp_cells_computed"]
            peak = line["roofline"]["peak_gcups"]
            lines.append(f"DP cells computed per fill (bench line): {cells} -> {cells / mean / 1e9:.4f} TCUPS over the traced span = {cells / mean / 1e6 / peak:.4f} of the "
                         f"{peak / 1e3:.3f}-TCUPS VALU ceiling; the bench line's own HIP-event figure (no profiler attached): ms_kernels_per_fill "
                         f"{line['roofline']['ms_kernels_per_fill']:.3f} ms, frac {line['roofline']['frac']:.4f}")
        per = collections.defaultdict(int)
        for *_, p in fills[1:] if len(fills) > 1 else fills:
            for name, v in p.items():
                per[name] += v
        nf = max(len(fills) - 1, 1)
        lines.append("# summed duration per kernel and fill (ms), fills 1..:")
        for name, v in sorted(per.items(), key=lambda kv: -kv[1])[:12]:
            lines.append(f"{v / 1e6 / nf:10.3f}  {name}")
    text = "\n".join(lines) + "\n"
    open(out_path, "w").write(text)
    sys.stdout.write(text)


import sys as _sys
if len(_sys.argv) > 1 and _sys.argv[1] == "k4span":
    k4span(_sys.argv[2], _sys.argv[3], _sys.argv[4] if len(_sys.argv) > 4 else None)
    raise SystemExit(0)


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
