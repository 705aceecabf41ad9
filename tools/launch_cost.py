#!/usr/bin/env python3
"""What `phamclust --gpus N` costs before the first pair is filled: wall time of
`phamclust synthN.tsv out -m peq` on one rank and as 2 / 4 ranks under the launcher (ranks share the one GPU of this box, the
gather rides gloo: everything but the RCCL transport itself is the product's N>1 route), split into the stages of the run's
`timing:` log line.  PHAMCLUST_FORCE_GPUS=1 keeps the CLI from falling back to one GPU, which is what it would do here.
    python tools/launch_cost.py [-n 5000] [--ranks 1,2,4] [--out profiles/r04/launch_cost.txt]"""
import argparse, json, os, shutil, subprocess, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from phamclust_amd import build
from phamclust_amd.synth import synth_packed, write_tsv_packed

ap = argparse.ArgumentParser()
ap.add_argument("-n", type=int, default=5000)
ap.add_argument("--ranks", default="1,2,4")
ap.add_argument("--metric", default="peq")
ap.add_argument("--out", default="")
a = ap.parse_args()
build.build_all()
work = tempfile.mkdtemp(prefix="launch_cost_")
tsv = os.path.join(work, f"synth{a.n}.tsv")
write_tsv_packed(synth_packed(a.n, 5000), tsv)
lines = [f"# tools/launch_cost.py -n {a.n} --ranks {a.ranks} -m {a.metric}: wall seconds of the CLI, by stage (rank 0's `timing:` line; second of two runs each)",
         "# interpreter_start = command start -> rank 0's interpreter running (under --gpus N: the parent's own start, its load for the estimate,",
         "#   torch.distributed.run); imports = the package and its dependencies; process_group_and_context = torch import + init_process_group + HIP context"]
rows = {}
runs = [(1, "process")] + [(r, route) for r in [int(x) for x in a.ranks.split(",")] if r > 1 for route in ("process", "launcher")]
for ranks, route in runs:
    best = None
    for rep in range(2):
        out = os.path.join(work, f"out_{ranks}_{route}_{rep}")
        # "process": THIS process drives `ranks` contexts (all on device 0 here) through pc_multi_*; "launcher": `ranks` processes under
        # torch.distributed.run sharing the GPU, the gather over gloo
        env = dict(os.environ, PYTHONPATH=REPO, PHAMCLUST_DIST_BACKEND="gloo", PHAMCLUST_FORCE_GPUS="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
                   PHAMCLUST_MULTI=route, PHAMCLUST_GPU_IDS=",".join(["0"] * ranks))
        t0 = time.time()
        proc = subprocess.run([sys.executable, "-m", "phamclust_amd", tsv, out, "-m", a.metric, "--gpus", str(ranks)], env=env, capture_output=True, text=True)
        wall = time.time() - t0
        if proc.returncode != 0:
            print(proc.stdout[-2000:], proc.stderr[-3000:]); sys.exit(1)
        log = open(os.path.join(out, "phamclust.log")).read()
        timing = json.loads([l for l in log.splitlines() if "timing: {" in l][-1].split("timing: ", 1)[1])
        timing["wall_of_the_command"] = round(wall, 3)
        timing["fill_line"] = [l.split("INFO: ", 1)[-1] for l in log.splitlines() if "genome-pairs/s" in l][-1]
        shutil.rmtree(out, ignore_errors=True)
        best = timing
    rows[(ranks, route)] = best
    lines.append(f"--gpus {ranks} ({route}): " + json.dumps(best))
    print(lines[-1], flush=True)
keys = ("interpreter_start", "imports", "load_genomes_for_estimate", "torch_import_and_process_group", "load_genomes", "pack", "process_group_and_context", "upload", "fill_exchange_d2h")
base = sum(rows[(1, "process")].get(k, 0.0) for k in keys)
for (ranks, route), t in rows.items():
    if ranks > 1:
        upto = sum(t.get(k, 0.0) for k in keys)
        lines.append(f"# --gpus {ranks} ({route}) against --gpus 1, command start -> matrix on the host: {upto:.2f} s against {base:.2f} s; fill stage alone "
                     f"{t.get('fill_exchange_d2h', 0.0):.3f} s against {rows[(1, 'process')].get('fill_exchange_d2h', 0.0):.3f} s (the devices are ONE GPU here) -> fixed cost ~ "
                     f"{upto - t.get('fill_exchange_d2h', 0.0) - (base - rows[(1, 'process')].get('fill_exchange_d2h', 0.0)):.2f} s")
        print(lines[-1])
if a.out:
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    open(a.out, "w").write("\n".join(lines) + "\n")
shutil.rmtree(work, ignore_errors=True)
