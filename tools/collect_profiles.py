#!/usr/bin/env python3
"""Turn what tools/profile_round.sh left under gpurun_out/<tag>_* into the tracked records under profiles/:

  <tag>_bench_peq5000.json                 the bench line
  <tag>_bench_peq5000_kernel_stats.csv     rocprofv3 --kernel-trace --stats of `bench.py --steps 3 --warmup 1`
  <tag>_set_metrics_kernel_stats.csv       the same for the set-metric sweep (k_set_popc, k_walk)
  <tag>_set_metric_sweep.json              N-sweep rows (tools/set_metric_bench.py)
  <tag>_counters.json                      PMC sums per kernel family and per fill: HBM traffic (FETCH_SIZE x2 on gfx950,
                                           WRITE_SIZE, separate passes), VALU instructions per DP cell, clock
  traffic.json                             entry stamped with the hash of the kernel sources it was measured on
usage: collect_profiles.py <tag> [<out dir under profiles/, e.g. r04/final>]
With an out dir the records drop the tag prefix: profiles/r04/final/bench_peq5000.json, ... (one directory per round, r04 on).
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
tag = sys.argv[1]
G = os.path.join(REPO, "gpurun_out")
P = os.path.join(REPO, "profiles")
OUT_DIR = sys.argv[2] if len(sys.argv) > 2 else ""
os.makedirs(os.path.join(P, OUT_DIR), exist_ok=True)


def rec(name):
    """Tracked path of record `name`: profiles/<tag>_<name>, or profiles/<out dir>/<name>."""
    return os.path.join(P, OUT_DIR, name) if OUT_DIR else os.path.join(P, f"{tag}_{name}")


def rec_rel(name):
    return os.path.relpath(rec(name), REPO)


def find(sub, suffix):
    hits = glob.glob(os.path.join(G, f"{tag}_{sub}", "**", f"*{suffix}"), recursive=True)
    return hits[0] if hits else None


def family(name):
    name = name.replace("void ", "")
    if name.startswith("k_nw_systolic"):
        return "k_nw_systolic"
    return name.split("(")[0]


def pmc(sub):
    path = find(sub, "counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    if not path:
        return agg, calls
    for r in csv.DictReader(open(path)):
        k = family(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
    return agg, {k: len(v) for k, v in calls.items()}


def git_head():
    try:
        return subprocess.check_output(["git", "-C", REPO, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        return None


import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(REPO, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
src_hash = bench.kernel_source_hash()

line = None
bp = os.path.join(G, f"{tag}_bench_peq5000.json")
if os.path.exists(bp):
    line = json.loads([l for l in open(bp).read().splitlines() if l.startswith("{")][-1])
    shutil.copy(bp, rec("bench_peq5000.json"))
for sub, dst in (("trace_peq", "bench_peq5000_kernel_stats.csv"), ("trace_set", "set_metrics_kernel_stats.csv")):
    path = find(sub, "kernel_stats.csv")
    if path:
        shutil.copy(path, rec(dst))
sp = os.path.join(G, f"{tag}_set_metric_sweep.json")
if os.path.exists(sp):
    shutil.copy(sp, rec("set_metric_sweep.json"))

out = {"tag": tag, "git_head_when_collected": git_head(), "kernel_source_hash": src_hash,
       "how": "tools/profile_round.sh: rocprofv3 --pmc in separate passes (FETCH_SIZE | WRITE_SIZE | SQ_*/GRBM), each around "
              "`bench.py --steps 1 --warmup 0` (3 fills: the step + the two wall-time fills) or one set-metric sweep",
       "gfx950_corrections": "FETCH_SIZE counts 64 B per 128-B request: doubled (MI355X_MICROARCH.md, HBM); FETCH_SIZE/WRITE_SIZE are in KB"}
fetch, fc = pmc("pmc_fetch")
write, wc = pmc("pmc_write")
valu, vc = pmc("pmc_valu")
fills = fc.get("k_walk<5>", 0) or 1
if fetch and write:
    per = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch[k].get("FETCH_SIZE", 0.0) * 1024, write[k].get("WRITE_SIZE", 0.0) * 1024
        per[k] = {"fetch_raw_bytes": f / fills, "fetch_corrected_bytes": 2 * f / fills, "write_bytes": w / fills,
                  "traffic_bytes": (2 * f + w) / fills, "launches_per_fill": fc.get(k, 0) / fills}
    out["peq5000_hbm_traffic_per_fill"] = {"fills_in_run": fills, "kernels": per}
    nw = per.get("k_nw_systolic")
    if nw and line:
        cells = line["roofline"]["dp_cells_computed"]
        algo = line["roofline"]["hbm"]["algorithmic_bytes_per_fill"]
        out["peq5000_hbm_traffic_per_fill"]["k_nw_systolic_vs_algorithmic"] = {"algorithmic_bytes": algo, "traffic_over_algorithmic": nw["traffic_bytes"] / algo,
                                                                              "bytes_per_dp_cell": nw["traffic_bytes"] / cells}
        tj = os.path.join(P, "traffic.json")
        doc = json.load(open(tj))
        doc["entries"] = [e for e in doc["entries"] if not (e["workload"] == "synth(5000,5000) -m peq" and e["n_gpus"] == 1)]
        doc["entries"].append({"workload": "synth(5000,5000) -m peq", "n_gpus": 1, "traffic_bytes_per_fill": int(nw["traffic_bytes"]),
                               "fetch_raw_bytes": int(nw["fetch_raw_bytes"]), "write_bytes": int(nw["write_bytes"]),
                               "source": rec_rel("counters.json"), "kernel_source_hash": src_hash, "git_head_when_collected": git_head()})
        json.dump(doc, open(tj, "w"), indent=1)
if valu:
    v = valu.get("k_nw_systolic", {})
    vf = vc.get("k_walk<5>", 0) or 1
    out["peq5000_valu_per_fill"] = {k: val / vf for k, val in v.items()}
    if line and v.get("SQ_INSTS_VALU"):
        cells = line["roofline"]["dp_cells_computed"]
        out["peq5000_valu_per_fill"]["valu_lane_instructions_per_dp_cell"] = v["SQ_INSTS_VALU"] / vf * 64 / cells
        out["peq5000_valu_per_fill"]["note"] = ("SQ_INSTS_VALU counts wave-level instructions; x64 lanes / DP cells computed; the hand-scheduled cell is 10 or 11 "
                                               "(per launch class), the rest is idle lanes, step prologues, refills")
sf, sfc = pmc("pmc_set_fetch")
sw, swc = pmc("pmc_set_write")
if sf and sw:
    per = {}
    for k in sorted(set(sf) | set(sw)):
        n = max(sfc.get(k, 1), 1)
        f, w = sf[k].get("FETCH_SIZE", 0.0) * 1024, sw[k].get("WRITE_SIZE", 0.0) * 1024
        per[k] = {"launches": n, "fetch_corrected_bytes_per_launch": 2 * f / n, "write_bytes_per_launch": w / max(swc.get(k, 1), 1),
                  "traffic_bytes_per_launch": 2 * f / n + w / max(swc.get(k, 1), 1)}
    out["set_metrics_n20000_hbm_traffic"] = {"kernels": per, "algorithmic_bytes_per_fill": 20000 * 79 * 8 + 16 * 20000 + 8 * (20000 * 19999 // 2)}
json.dump(out, open(rec("counters.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
