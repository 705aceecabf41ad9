#!/usr/bin/env python3
"""bench.py -- the headline measurement: genome-pairs/sec of the N x N matrix fill.

    python bench.py --gpus N --steps K --warmup W          any N: for N > 1 outside a launcher this process starts
                                                           `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a
                                                           CHILD before it has made a GPU call, and relays rank 0's line and the
                                                           child's exit status (it never re-executes itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...       the same ranks, launched by the caller
    python bench.py --gpus N --route process               ONE process drives the N GPUs (pc_multi_*: a host thread per device inside the
                                                           library, the shard exchange as peer copies) -- the same shard and assembly,
                                                           so the peer-copy exchange can be compared with the RCCL gather on one node

One "step" = one full matrix fill (BASELINE.json metric: genome-pairs/sec, peq) of the
workload synth(5000, 5000) -m peq -- the configuration the target is quoted on -- with the
packed genomes already resident in HBM.  With N ranks the SAME matrix is filled by N GPUs
(static pair shard + one RCCL gather + device-side assembly on rank 0), so scaling is
"strong".  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline      the dominant kernels (the K4 alignment launches of one fill, timed with HIP
                events on the stream they run on, inside the library).  The recurrence is bound
                by VALU issue: every instruction class of the DP cell (v_max_f64 included) retires
                one wave64 instruction per 4 clocks per SIMD (measured: profiles/valu_issue_rate.json),
                so peak = 256 CU x 4 SIMD x 16 lanes x 2.4 GHz lane-ops/s and achieved = 10 x DP cells
                computed / kernel time: 10 instructions is the cheaper of the kernel's two cells
                (launch classes whose LDS cannot hold its 16-bit profile at four waves per SIMD run
                the 11-instruction cell: counted as a loss, not as a lower ceiling).  `hbm` inside it
                is the figure the north star asks for: algorithmic bytes = sum(la+lb) residues
                read + 16 B per alignment (bucket entry in, result out) against 8 TB/s -- small
                by construction.  `traffic` = HBM bytes of those launches from rocprofv3 PMC
                passes (profiles/traffic.json), reported only when they were taken on the same
                kernel sources as the library that is running.
  cpu_baseline  the oracle (C restatement of the reference path, OpenMP over rows) timed on
                this host on a bounded sample of the same workload (leading matrix rows), on
                every core the process may use, plus a one-thread sample.
  wall_ms_incl_upload_d2h   SURVEY 8(d)'s wall time of one matrix: upload of the packed genomes
                + kernels + D2H of the condensed vector (never `value`); `value_wall` = pairs / that.
  stage_ms      N = 1: plan / align / reduce of one fill (HIP events inside the library).  N > 1: per-rank
                stages as max over ranks, the exchange (gather or reduce) and the root-only assembly as rank 0
                sees them, `shards` (pairs / cells min-max over ranks), `config.dist_mode`.
Exit status 1 (and "valid": false) when the sampled oracle check is not bit-exact.
"""

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genomes", type=int, default=5000)
    ap.add_argument("--phams", type=int, default=5000)
    ap.add_argument("--metric", default="peq", choices=["gcs", "jc", "pocp", "af", "aai", "peq"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="lower bound of CPU-baseline work; 0 disables")
    ap.add_argument("--verify-pairs", type=int, default=20000, help="check this many random pairs against the oracle")
    ap.add_argument("--route", default="rank", choices=["rank", "process"],
                    help="N > 1: 'rank' = one process per GPU, one RCCL gather (the measured contract); 'process' = this process drives "
                         "all N GPUs through pc_multi_* (peer copies)")
    return ap.parse_args()


def spawn_ranks(a):
    """`python bench.py --gpus N` outside a launcher: run the N ranks as a CHILD `torch.distributed.run`, relay rank 0's JSON line
    (with `launched_by` added) and the child's exit status.  Nothing here may touch the GPU -- this process never imports torch,
    never loads libphamclust_hip and never builds: the ranks own the devices (a process that has initialised HIP must not exec,
    and has no business holding a context next to the ranks' on device 0)."""
    import subprocess
    from phamclust_amd.distributed import free_port              # numpy only
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["PC_BENCH_PARENT_T0"] = repr(time.time())
    touched = sorted(m for m in ("torch", "phamclust_amd.hip") if m in sys.modules)
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for text in child.stdout:                                     # rank 0's one JSON line; anything else a rank printed goes to stderr
        try:
            obj = json.loads(text)
            if isinstance(obj, dict) and "metric" in obj:
                line = obj
                continue
        except ValueError:
            pass
        sys.stderr.write(text)
    rc = child.wait()
    if line is not None:
        line["launched_by"] = {"how": "bench.py parent -> child `python -m torch.distributed.run` (no GPU call, no torch import in the parent)",
                               "parent_modules_touching_gpu": touched, "child_exit_status": rc}
        print(json.dumps(line), flush=True)
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks exited 0 without a result line\n")
        rc = 1
    return rc


def cpu_quota():
    """CPUs this process's cgroup may use at once (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited: a GPU box
    with 256 hardware threads may grant its one-GPU tenant 16 of them, and 256 OpenMP threads on 16 CPUs run slower
    than 16."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        return None if quota == "max" else max(1, int(int(quota) / int(period)))
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:
            quota = int(fh.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
            period = int(fh.read())
        return None if quota <= 0 else max(1, quota // period)
    except (OSError, ValueError):
        return None


def cpu_baseline(packed, metric, min_seconds):
    """Oracle (kind 'port') on leading rows until >= min_seconds of work (<= ~2.5x that)."""
    from oracle import oracle as O
    affinity = len(os.sched_getaffinity(0))
    quota = cpu_quota()
    threads = max(1, min(affinity, quota) if quota else affinity)          # every core this process may actually use
    n = packed.n_genomes
    rows_done, pairs, aln, cells, elapsed = 0, 0, 0, 0, 0.0
    chunk = 1
    while rows_done < n - 1 and elapsed < min_seconds:
        hi = min(n - 1, rows_done + chunk)
        t0 = time.perf_counter()
        _, a, c = O.fill_rows(packed, metric, rows_done, hi, as_distance=True, nthreads=threads)
        dt = time.perf_counter() - t0
        elapsed += dt
        pairs += sum(n - 1 - s for s in range(rows_done, hi))
        aln += a
        cells += c
        per_row = max(dt / (hi - rows_done), 1e-9)
        rows_done = hi
        chunk = max(1, min(int((min_seconds - elapsed) / per_row * 1.1) + 1, 4 * chunk))
    # one thread as well (a short sample), so the per-core rate of the port is on record
    one_rows = max(1, rows_done // max(threads, 1) // 4)
    t0 = time.perf_counter()
    _, a1, c1 = O.fill_rows(packed, metric, 0, one_rows, as_distance=True, nthreads=1)
    dt1 = time.perf_counter() - t0
    pairs1 = sum(n - 1 - s for s in range(one_rows))
    single = {"value": pairs1 / dt1, "cores": 1, "gcups": c1 / dt1 / 1e9, "seconds": dt1, "rows": one_rows}
    out = {"value": pairs / elapsed, "unit": "genome-pairs/s", "cores": threads, "host_cpu_count": os.cpu_count(), "affinity_cpus": affinity,
           "cgroup_cpu_quota": quota, "kind": "port",
           "single_thread": single,
           "sample": f"oracle/pc_oracle.c (OpenMP, {threads} threads) on matrix rows [0,{rows_done}) of the same workload: "
                     f"{pairs} pairs, {aln} alignments, {cells} DP cells in {elapsed:.2f} s",
           "gcups": cells / elapsed / 1e9, "seconds": elapsed,
           "vs_reference_note": "a compiled C port, i.e. a GENEROUS stand-in for the reference's Python + joblib path; the reference "
                                "itself cannot travel to this box"}
    try:                                                  # how the port relates to the real reference (build container, one thread)
        with open(os.path.join(REPO, "profiles", "calibration_reference_vs_oracle.json")) as fh:
            cal = json.load(fh)
        ratios = {row["metric"]: round(row["oracle_over_reference_t1"], 1) for row in cal["rows"]}
        out["calibration_vs_reference"] = {
            "source": "profiles/calibration_reference_vs_oracle.json (live reference matrix_de_novo -t 1 vs this port -t 1, synth(400,5000), identical values)",
            "port_over_reference_one_thread": ratios,
            "aai_peq": "not calibratable: parasail is absent from the reference tree and this image; the reference's README quotes "
                       "~475-500 peq pairs/s on an Apple M1"}
    except (OSError, ValueError, KeyError):
        pass
    return out


def kernel_source_hash():
    """sha256 of the DEVICE code the library carries (the .hip_fatbin section of libphamclust_hip.so): PMC records are only
    quoted for the kernels they were taken on.  (Named for what it was in r02's first records: a hash over the kernel
    sources, which went stale with every comment.)"""
    import hashlib
    import struct
    path = os.path.join(REPO, "phamclust_amd", "csrc", "libphamclust_hip.so")
    with open(path, "rb") as fh:
        blob = fh.read()
    try:
        shoff, = struct.unpack_from("<Q", blob, 0x28)
        shentsize, shnum, shstrndx = struct.unpack_from("<HHH", blob, 0x3A)
        def section(i):
            name, _type, _flags, _addr, off, size = struct.unpack_from("<IIQQQQ", blob, shoff + i * shentsize)
            return name, off, size
        _, stroff, strsize = section(shstrndx)
        names = blob[stroff:stroff + strsize]
        for i in range(shnum):
            name, off, size = section(i)
            if names[name:names.index(b"\0", name)] == b".hip_fatbin":
                return hashlib.sha256(blob[off:off + size]).hexdigest()[:16]
    except (struct.error, ValueError):
        pass
    return hashlib.sha256(blob).hexdigest()[:16]


def verify_sample(packed, metric, n_verify, fetch):
    """Sampled check of the assembled matrix against the oracle (random pairs over the whole triangle); ``fetch(condensed
    indices) -> values`` reads them from wherever the matrix is."""
    from oracle import oracle as O
    import numpy as np
    from phamclust_amd.metrics import parity_note
    rng = np.random.default_rng(12345)
    n = packed.n_genomes
    s_idx = rng.integers(0, n - 1, n_verify)
    t_idx = rng.integers(0, n, n_verify)
    lo, hi = np.minimum(s_idx, t_idx), np.maximum(s_idx, t_idx)
    keep = lo < hi
    lo, hi = lo[keep], hi[keep]
    cond = lo * n - lo * (lo + 1) // 2 + (hi - lo - 1)
    got = np.asarray(fetch(cond))
    want = O.pairs(packed, metric, lo, hi, as_distance=True)
    return {"pairs": int(lo.size), "how": "random pairs of the full matrix vs oracle/pc_oracle.c",
            "max_abs_diff": float(np.max(np.abs(got - want))), "bit_exact": bool(np.array_equal(got, want)),
            "parity": parity_note(metric)}


def child_probe(a, rank, world):
    """PC_BENCH_CHILD_PROBE=1 (tests/test_distributed_cpu.py): what a rank does BEFORE its first GPU call -- join the group the
    parent's launcher set up -- and nothing after: rank 0 prints a line that says how many ranks met.  Runs where there is no GPU."""
    import torch.distributed as dist
    dist.init_process_group(os.environ.get("PC_BENCH_BACKEND", "gloo"))
    box = [None] * world
    dist.all_gather_object(box, {"rank": rank, "pid": os.getpid(), "argv": sys.argv[1:]})
    if rank == 0:
        print(json.dumps({"metric": "genome-pairs/sec", "value": None, "probe": True, "n_gpus": a.gpus, "ranks_seen": dist.get_world_size(),
                          "ranks": box, "parent_t0": os.environ.get("PC_BENCH_PARENT_T0")}), flush=True)
    dist.destroy_process_group()
    if os.environ.get("PC_BENCH_CHILD_PROBE_FAIL") == str(rank):
        sys.exit(7)


def main_one_process(a):
    """--route process: this process drives the N GPUs (hip.MultiContext = pc_multi_*).  One step = upload-free fill of the whole
    matrix DELIVERED to page-locked host memory (the call's contract: shard fills on every device, peer copies into the root's
    gather buffer, assembly, D2H) -- so `value` here contains the D2H the rank route's `value` leaves out; `value_resident_estimate`
    takes it off again.  PC_BENCH_DEVICE_IDS=0,0 names the devices (an id may repeat: rehearsal on a one-GPU box)."""
    from phamclust_amd import build, hip
    from phamclust_amd.synth import synth_packed
    build.build_all()
    ids = [int(x) for x in os.environ["PC_BENCH_DEVICE_IDS"].split(",")] if os.environ.get("PC_BENCH_DEVICE_IDS") else list(range(a.gpus))
    if len(ids) != a.gpus:
        sys.exit(f"PC_BENCH_DEVICE_IDS names {len(ids)} devices, --gpus {a.gpus}")
    packed = synth_packed(a.genomes, a.phams)
    n_pairs = packed.n_pairs
    needs_residues = a.metric in ("aai", "peq")
    t0 = time.perf_counter()
    mc = hip.MultiContext(ids)
    t_create = time.perf_counter() - t0
    t0 = time.perf_counter()
    mc.upload(packed, residues=needs_residues)
    upload_s = time.perf_counter() - t0
    out, st = None, None
    for _ in range(a.warmup):
        out, st = mc.fill(a.metric, True, want_stats=True, borrow=True)
    stats = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out, st = mc.fill(a.metric, True, want_stats=True, borrow=True)        # synchronous: returns with the matrix on the host
        stats.append(st)
    elapsed = time.perf_counter() - t0
    world = len(ids)
    per = [s["per_device"] for s in stats]
    mean = lambda f: sum(f(s) for s in stats) / len(stats)                       # noqa: E731
    ms_align_max = mean(lambda s: max(p["ms_align"] for p in s["per_device"]))
    ms_align_min = mean(lambda s: min(p["ms_align"] for p in s["per_device"]))
    last = per[-1]
    tot = lambda key: sum(p[key] for p in last)                                # noqa: E731
    distinct_same_gpu = len(set(ids)) < len(ids)
    line = {
        "metric": "genome-pairs/sec", "value": n_pairs * a.steps / elapsed, "unit": "genome-pairs/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "int32", "data": "synthetic", "route": "process", "ranks_seen": 1, "device_ids": ids,
        "config": {"workload": f"synth({a.genomes},{a.phams}) -m {a.metric}: full N x N distance-matrix fill from ONE process over {world} device(s); "
                               f"`value` = fills with the genomes resident in HBM, the matrix DELIVERED to page-locked host memory (D2H included)",
                   "n_genomes": a.genomes, "n_phams": packed.n_phams, "metric_selector": a.metric, "genome_pairs": n_pairs,
                   "n_genes": packed.n_genes, "n_residues": int(packed.residues.size),
                   "parallelism": f"static cost-balanced pair shard over {world} device(s) driven by one process (pc_multi_*) + peer copies into the "
                                  f"root's gather buffer + device assembly" + (" [REHEARSAL: contexts share a GPU]" if distinct_same_gpu else "")},
        "stage_ms": {"align_max_over_devices": ms_align_max, "align_min_over_devices": ms_align_min,
                     "plan_max_over_devices": mean(lambda s: max(p["ms_plan"] for p in s["per_device"])),
                     "device_total_max_over_devices": mean(lambda s: s["ms_total"]),
                     "exchange_slowest_copy": mean(lambda s: s["ms_exchange"]), "assemble_root": mean(lambda s: s["ms_assemble"]),
                     "exchange": "every device copies its shard to the root's gather buffer (hipMemcpyPeerAsync / same-device copy)",
                     "exchange_bytes": int(8 * n_pairs), "clock": "HIP events on each device's stream, inside the library"},
        "peer_access": stats[-1]["peer_access"],
        "shards": {"pairs_min_max": [min(p["n_pairs"] for p in last), max(p["n_pairs"] for p in last)],
                   "cells_min_max": [min(p["n_cells"] for p in last), max(p["n_cells"] for p in last)]},
        "create_s": t_create, "upload_s": upload_s,
    }
    line["value_wall_incl_init"] = n_pairs / (t_create + upload_s + elapsed / a.steps)
    if needs_residues:
        line["roofline"], line["kernel_source_hash"] = aligned_roofline(
            a, world, tot("n_alignments"), tot("n_cells"), tot("n_residue_bytes"), tot("n_distinct_alignments"), tot("n_distinct_cells"),
            ms_align_max, [ms_align_min, ms_align_max])
    else:
        nb = packed.n_genomes * packed.words_per_row * 8 + 16 * packed.n_genomes + 8 * n_pairs
        t = mean(lambda s: s["ms_total"]) / 1e3
        line["roofline"] = {"bound": "hbm", "achieved": nb / t / 1e9 if t > 0 else 0.0, "peak": 8000.0 * world, "unit": "GB/s",
                            "frac": nb / t / 1e9 / (8000.0 * world) if t > 0 else 0.0, "traffic": None, "algorithmic_bytes_per_fill": nb}
    if a.verify_pairs > 0 and n_pairs > 0:
        import numpy as np
        line["verified"] = verify_sample(packed, a.metric, a.verify_pairs, lambda cond: np.asarray(out)[cond])
    line["valid"] = bool(line.get("verified", {}).get("bit_exact", True))
    print(json.dumps(line), flush=True)
    mc.close()
    if not line["valid"]:
        sys.exit(1)


def aligned_roofline(a, world, n_aln, n_cells, n_rbytes, n_daln, n_dcells, ms_align, align_span):
    """The `roofline` object of an aai / peq fill: the K4 launches of one fill; multi-GPU: work of all devices / slowest device's time."""
    # this rank-set's K4 launches of one fill; multi-GPU: work of all ranks / slowest rank's time
    algo_bytes = n_rbytes + 16 * n_aln
    per_gpu_time = ms_align / 1e3
    hbm_achieved = algo_bytes / world / per_gpu_time / 1e9 if per_gpu_time > 0 else 0.0
    gcups = n_dcells / world / per_gpu_time / 1e9 if per_gpu_time > 0 else 0.0
    # The bound that binds: VALU issue, priced per instruction class.  A SIMD retires a wave64 VALU instruction in 4 clocks
    # (16 lanes/clk) for v_max_f64, SDWA / DPP / VOP3 forms, compares and v_addc, and in 2 (32 lanes/clk) for the plain VOP2
    # add / sub / and / or once several waves share it (tests/hw/valu_rate.hip -> profiles/valu_issue_rate.json: 4.15 and
    # 2.15 in unmixed streams).  The cell is 10 instructions: four v_max_f64 acting as lexicographic (score, tie-break tag, path
    # statistics) maxima and two SDWA adds from the profile (4 clocks each), four tag / open-penalty fix-ups -- v_or, v_or,
    # v_and, v_sub (2 each): 32 issue clocks per 64 cells, 4,915 GCUPS.  r01-r03 priced all ten at 4 clocks (3,932 GCUPS): a
    # probe of identical waves running a mixed stream in lockstep measures 4.0 per instruction whatever the mix
    # (tests/hw/valu_mix.hip).  The kernel itself refutes that ceiling: its counters show 11.49 executed VALU instructions per
    # cell (the 11-instruction compare cell of long column genes, step prologues, idle lanes), and 11.49 x 64-lane
    # instructions at this line's rate are 3.8 clocks per instruction per SIMD -- below 4 -- so its waves, which are not in
    # lockstep, do get the cheap class cheaper; and turning two VOP3 re-tags into VOP2 ones (r04) took 6 % off the fill at the
    # same instruction count.  `frac_at_4_clk_per_instruction` is the r03 figure's successor, for continuity only.
    instr_per_cell = 10
    issue_clk_per_cell = 6 * 4 + 4 * 2
    peak_gcups = 256 * 4 * 64 * 2.4e9 / issue_clk_per_cell / 1e9
    roof = {
        "bound": "valu", "achieved": gcups / 1e3, "peak": peak_gcups / 1e3, "unit": "TCUPS (DP cells/s; VALU issue clocks per cell priced per instruction class)",
        "frac": gcups / peak_gcups, "traffic": None,
        "kernel": "k_nw_systolic_tier<TIER,RULE,cell> / k_nw_systolic<W,RULE> (all launches of one fill, per GPU)",
        "instr_per_cell": instr_per_cell, "issue_clk_per_cell": issue_clk_per_cell, "peak_gcups": peak_gcups, "achieved_gcups": gcups,
        "frac_at_4_clk_per_instruction": gcups / (256 * 4 * 16 * 2.4e9 / instr_per_cell / 1e9),
        "issue_rate_source": "profiles/valu_issue_rate.json (unmixed: 4.15 / 2.15 clk), profiles/r05/final/counters.json (11.49 executed VALU instructions per cell), "
                             "profiles/r04/experiments/retag_one_op_ab.txt, valu_mix_probe.txt",
        "ms_kernels_per_fill": ms_align, "n_alignments": n_aln, "dp_cells": n_cells,
        # what the kernels computed: identical (row sequence, column sequence) pairs are aligned once per rank
        "n_distinct_alignments": n_daln, "dp_cells_computed": n_dcells, "gcups_per_gpu": gcups,
        "hbm": {"bound": "hbm", "achieved": hbm_achieved, "peak": 8000.0, "unit": "GB/s", "frac": hbm_achieved / 8000.0,
                "algorithmic_bytes_per_fill": algo_bytes,
                "note": "sum(la+lb) residues + 16 B per alignment over the kernels' time: far below HBM speed by construction "
                        "(the recurrence is integer-VALU bound, no MFMA: there is no dense contraction)"},
    }
    if align_span:
        roof["ms_kernels_min_max_over_ranks"] = align_span
    src_hash = kernel_source_hash()
    # the same quantity from a rocprofv3 --kernel-trace of this command (first K4 start -> last K4 end per fill; tools/summarize_profile.py
    # k4span): a tracked file, so that `frac` can be redone from profiles/ alone.  Quoted with the device code it was taken on.
    import glob
    import re
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*", "final", "bench_peq5000_k4_span.txt")), reverse=True):
        if world != 1 or (a.genomes, a.phams, a.metric) != (5000, 5000, "peq"):
            break
        text = open(path).read()
        m = re.search(r"span, mean over fills [0-9.]+: ([0-9.]+) ms", text)
        h = re.search(r"kernel_source_hash\): (\w+)", text)
        if m:
            span = float(m.group(1))
            roof["rocprof_k4_span"] = {"source": os.path.relpath(path, REPO), "span_ms_per_fill": span, "frac_from_span": n_dcells / span / 1e6 / peak_gcups,
                                       "taken_on_kernel_source_hash": h.group(1) if h else None, "same_device_code": bool(h and h.group(1) == src_hash)}
        break
    try:                                              # PMC traffic measured offline for this exact workload AND these sources
        with open(os.path.join(REPO, "profiles", "traffic.json")) as fh:
            for e in json.load(fh)["entries"]:
                if e["workload"] == f"synth({a.genomes},{a.phams}) -m {a.metric}" and e["n_gpus"] == world:
                    if e.get("kernel_source_hash") == src_hash:
                        roof["traffic"] = e["traffic_bytes_per_fill"]
                        roof["traffic_source"] = e["source"]
                    else:
                        roof["traffic_stale"] = {"bytes_per_fill": e["traffic_bytes_per_fill"], "source": e["source"],
                                                             "taken_on_kernel_source_hash": e.get("kernel_source_hash")}
    except (OSError, KeyError, ValueError):
        pass
    return roof, src_hash


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and a.gpus > 1 and a.route == "rank":
        sys.exit(spawn_ranks(a))                               # before anything below: no torch, no library, no GPU in this process
    if a.route == "process":
        if world != 1:
            sys.exit("--route process is ONE process driving all GPUs: do not start it under a launcher")
        return main_one_process(a)
    if world != a.gpus:
        a.gpus = world
    if os.environ.get("PC_BENCH_CHILD_PROBE") == "1":         # tests/: the spawn plumbing alone, on a machine without a GPU
        return child_probe(a, rank, world)

    # PC_BENCH_FORCE_DIST=1: the whole N > 1 flow -- process group, sharded fill, exchange, assembly, the all_reduce bookkeeping of
    # this file -- with ONE rank, so that it runs through RCCL on a one-GPU box (two ranks cannot share a GPU under RCCL)
    multi = world > 1 or os.environ.get("PC_BENCH_FORCE_DIST") == "1"
    import torch
    import torch.distributed as dist
    from phamclust_amd import build, hip
    from phamclust_amd.distributed import dist_mode, fill_distributed, uses_alignment_slices
    from phamclust_amd.metrics import parity_note
    from phamclust_amd.synth import synth_packed

    # PC_BENCH_BACKEND=gloo rehearses the N>1 flow on a box with fewer GPUs than ranks (ranks share devices, the
    # gather is staged through host memory); the measured configuration is always nccl (= RCCL), one GPU per rank
    backend = os.environ.get("PC_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if multi and world == 1:                               # the N > 1 flow with ONE rank (see `multi` above)
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
        os.environ["PHAMCLUST_DIST_FORCE_EXCHANGE"] = "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if int(os.environ.get("LOCAL_RANK", "0")) == 0:
        build.build_all()                 # no-op when the in-tree libraries are current; one rank only
    if multi:
        dist.barrier()
    # What a rank costs before it can fill anything (never part of `value`): this process's own start -- interpreter, torch
    # import, process-group init -- up to the first barrier all ranks pass.  Max over ranks below.  (The launcher's own
    # start-up is on top: tools/launch_cost.py measures the whole command.)
    from phamclust_amd.startup import process_start_time
    init_s = time.time() - process_start_time()

    packed = synth_packed(a.genomes, a.phams)
    ctx = hip.Context(local_rank)
    needs_residues = a.metric in ("aai", "peq")
    ctx.upload(packed, residues=needs_residues)            # gcs / jc / pocp / af never read a residue: part 1 of the upload only
    n_pairs = packed.n_pairs
    sliced = multi and uses_alignment_slices(a.metric)                               # PHAMCLUST_DIST_MODE=alignments

    def step():
        if not multi:
            out = torch.empty(max(n_pairs, 1), dtype=torch.float64, device="cuda")
            st = ctx.fill_dev(a.metric, True, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
            return out[:n_pairs], st
        return fill_distributed(ctx, a.metric, True)

    def fence():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    out, st = None, None
    for _ in range(a.warmup):
        out, st = step()
    fence()
    t0 = time.perf_counter()
    stats = []
    for _ in range(a.steps):
        out, st = step()
        stats.append(st)
    fence()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        agg = torch.tensor([[s["n_alignments"], s["n_cells"], s["n_residue_bytes"], s["n_tasks"], s["ms_align"] * 1e3,
                             s["ms_total"] * 1e3, s["n_distinct_alignments"], s["n_distinct_cells"],
                             s["ms_plan"] * 1e3, s["ms_reduce"] * 1e3, s.get("ms_exchange", 0.0) * 1e3, s.get("shard_pairs", 0),
                             s["n_chunks"]] for s in stats],
                           dtype=torch.float64, device="cuda")
        agg_sum = agg.clone(); dist.all_reduce(agg_sum, op=dist.ReduceOp.SUM)
        agg_max = agg.clone(); dist.all_reduce(agg_max, op=dist.ReduceOp.MAX)
        agg_min = agg.clone(); dist.all_reduce(agg_min, op=dist.ReduceOp.MIN)
        # the fixed cost of being N ranks (every rank takes part: max over ranks): process start -> group ready, and one upload
        t = torch.tensor([init_s], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        init_s = float(t.item())
        t0 = time.perf_counter()
        ctx.upload(packed, residues=needs_residues)
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        upload_s = float(t.item())
    if rank != 0:
        if multi:
            dist.destroy_process_group()
        return

    ms_step = elapsed / a.steps * 1e3
    value = n_pairs * a.steps / elapsed
    if not multi:
        n_aln = stats[-1]["n_alignments"]; n_cells = stats[-1]["n_cells"]; n_rbytes = stats[-1]["n_residue_bytes"]
        n_daln = stats[-1]["n_distinct_alignments"]; n_dcells = stats[-1]["n_distinct_cells"]
        ms_align = sum(s["ms_align"] for s in stats) / len(stats)
        ms_dev = sum(s["ms_total"] for s in stats) / len(stats)
        align_span = None
    else:
        # "pairs" mode: every rank reports its shard, the job is the sum; "alignments" mode: every rank plans the whole
        # fill and reports the whole job's counts
        whole = agg_max if sliced else agg_sum
        n_aln = int(whole[-1, 0]); n_cells = int(whole[-1, 1]); n_rbytes = int(whole[-1, 2])
        n_daln = int(whole[-1, 6]); n_dcells = int(whole[-1, 7])
        ms_align = float(agg_max[:, 4].mean()) / 1e3          # slowest rank's alignment time per step
        ms_dev = float(agg_max[:, 5].mean()) / 1e3
        align_span = [float(agg_min[:, 4].mean()) / 1e3, float(agg_max[:, 4].mean()) / 1e3]
        # where a multi-GPU step's time goes: per-rank stages as max over ranks (mean over steps), the exchange as rank 0
        # sees it (it ends when the slowest rank's data has arrived: it contains the wait for that rank), and the root-only
        # stages (assembly of the gathered shards, or the matrix build of the alignment-sliced route)
        root_ms = lambda key: sum(s.get(key, 0.0) for s in stats) / len(stats)
        stage_multi = {"plan_max_over_ranks": float(agg_max[:, 8].mean()) / 1e3, "align_max_over_ranks": float(agg_max[:, 4].mean()) / 1e3,
                       "align_min_over_ranks": float(agg_min[:, 4].mean()) / 1e3,
                       "reduce_max_over_ranks": float(agg_max[:, 9].mean()) / 1e3,
                       "device_total_max_over_ranks": float(agg_max[:, 5].mean()) / 1e3,
                       "exchange_rank0": root_ms("ms_exchange"), "exchange_max_over_ranks": float(agg_max[:, 10].mean()) / 1e3,
                       "exchange": "one RCCL reduce of the alignment results" if sliced else "one RCCL gather of the matrix shards",
                       "exchange_bytes": int(stats[-1].get("exchange_bytes", 0)),
                       "assemble_rank0": root_ms("ms_root_reduce" if sliced else "ms_assemble"),
                       "clock": "HIP events on the stream the work runs on" if backend == "nccl" else "host clock (rehearsal transport stages through host memory)"}
        shard_span = {"pairs_min_max": [int(agg_min[-1, 11]), int(agg_max[-1, 11])], "cells_min_max": [int(agg_min[-1, 1]), int(agg_max[-1, 1])],
                      "distinct_cells_min_max": [int(agg_min[-1, 7]), int(agg_max[-1, 7])], "chunks_max": int(agg_max[-1, 12])}

    line = {
        "metric": "genome-pairs/sec", "value": value, "unit": "genome-pairs/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "int32", "data": "synthetic", "route": "rank",
        "ranks_seen": dist.get_world_size() if multi else 1, "devices_visible": torch.cuda.device_count(),
        "config": {"workload": f"synth({a.genomes},{a.phams}) -m {a.metric}: full N x N distance-matrix fill; `value` = fills with the genomes resident in "
                               f"HBM and the matrix left in HBM (upload and D2H excluded: see value_wall)",
                   "dist_mode": (dist_mode() if a.metric in ("aai", "peq") else "pairs") if multi else None,
                   "n_genomes": a.genomes, "n_phams": packed.n_phams, "metric_selector": a.metric, "genome_pairs": n_pairs,
                   "n_genes": packed.n_genes, "n_residues": int(packed.residues.size),
                   "parallelism": (f"alignments sliced over {world} GPUs (every rank plans the whole fill) + 1 RCCL reduce of their results + matrix on rank 0"
                                   if sliced else f"static pair shard over {world} GPU(s)" + (" + 1 RCCL gather + device assembly" if multi else ""))
                                  + ("" if backend == "nccl" or world == 1 else f" [REHEARSAL: backend {backend}, ranks share GPUs]")
                                  + (" [REHEARSAL: ONE rank, exchange forced (PC_BENCH_FORCE_DIST)]" if multi and world == 1 else "")},
    }
    if a.metric in ("aai", "peq"):
        line["roofline"], line["kernel_source_hash"] = aligned_roofline(a, world, n_aln, n_cells, n_rbytes, n_daln, n_dcells, ms_align, align_span)
    else:
        nb = packed.n_genomes * packed.words_per_row * 8 + 16 * packed.n_genomes + 8 * n_pairs
        t = ms_dev / 1e3
        line["roofline"] = {"bound": "hbm", "achieved": nb / t / 1e9 if t > 0 else 0.0, "peak": 8000.0, "unit": "GB/s",
                            "frac": nb / t / 1e9 / 8000.0 if t > 0 else 0.0, "traffic": None,
                            "kernel": "k_set_popc / k_walk", "algorithmic_bytes_per_fill": nb, "ms_kernels_per_fill": ms_dev}
    line["device_ms_per_fill"] = ms_dev
    # `value` is the resident-fill rate because the bench contract says so (inputs in HBM when the timed region starts; a PCIe-inclusive
    # rate "is never `value`"); SURVEY 8(d)'s wall time of one matrix -- upload + kernels + D2H -- is `value_wall` on the same line
    line["value_resident"] = value
    line["value_definition"] = ("value = value_resident: fills with inputs resident in HBM and the matrix left in HBM (the bench contract's definition); "
                                "value_wall = SURVEY 8(d): upload + kernels + D2H of one matrix, host clock" + ("" if world == 1 else "; N > 1: value_wall_incl_init"))
    if not multi:
        line["stage_ms"] = {k: sum(s[k] for s in stats) / len(stats) for k in ("ms_plan", "ms_align", "ms_reduce")}
        line["n_chunks"] = stats[-1]["n_chunks"]
    else:
        line["stage_ms"] = stage_multi
        line["shards"] = shard_span
        # the fixed cost of being N ranks, and the rate a user of one matrix sees with it: process start -> group ready (max over
        # ranks), + one upload, + one fill with its exchange (the timed step)
        line["init_s"] = init_s
        line["upload_s"] = upload_s
        line["value_wall_incl_init"] = n_pairs / (line["init_s"] + line["upload_s"] + elapsed / a.steps) if n_pairs else 0.0
        line["init_note"] = ("init_s: process start -> torch imported, process group initialised, first barrier passed (max over ranks); "
                             "value_wall_incl_init = pairs / (init_s + upload_s + one step): what ONE matrix costs a job that has to start its ranks first")

    if a.verify_pairs > 0 and n_pairs > 0:
        line["verified"] = verify_sample(packed, a.metric, a.verify_pairs,
                                         lambda cond: out[torch.as_tensor(cond, device=out.device)].cpu().numpy())
    if world == 1:
        # SURVEY 8(d)'s wall time of one matrix: upload + kernels + D2H of the condensed vector (host clock, second of two)
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.upload(packed, residues=needs_residues)
            t1 = time.perf_counter()
            host = ctx.fill(a.metric, True, borrow=True)
            t2 = time.perf_counter()
        line["wall_ms_incl_upload_d2h"] = (t2 - t0) * 1e3
        line["wall_breakdown_ms"] = {"upload": (t1 - t0) * 1e3, "kernels_plus_d2h_to_pinned_host": (t2 - t1) * 1e3}
        line["pairs_per_s_incl_upload_d2h"] = n_pairs / (t2 - t0) if n_pairs else 0.0
        # SURVEY 8(d)'s figure for one matrix, host clock: upload + kernels + D2H of the condensed vector
        line["value_wall"] = line["pairs_per_s_incl_upload_d2h"]
        assert host.shape[0] == n_pairs
    if world == 1 and a.cpu_seconds > 0:
        line["cpu_baseline"] = cpu_baseline(packed, a.metric, a.cpu_seconds)
    valid = bool(line.get("verified", {}).get("bit_exact", True))
    line["valid"] = valid
    print(json.dumps(line), flush=True)
    if multi:
        dist.destroy_process_group()
    if not valid:
        sys.exit(1)


if __name__ == "__main__":
    main()
