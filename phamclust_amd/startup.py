"""What ``--gpus N`` costs before the first pair is filled, and when it is not worth paying.

The reference pays its worker-pool start-up inside ``matrix_de_novo`` (matrix.py:471-472) and clamps the workers to the work
(matrix.py:460-462: never more CPUs than genome pairs).  Here ``--gpus N`` means one process per GPU under
``torch.distributed.run``: one interpreter, one torch import and one RCCL initialisation PER RANK before any rank fills a pair --
seconds, against a fill that takes 0.6 s for 5,000 genomes on one GPU.  So ``--gpus N`` by default keeps the job in ONE process
(``pc_multi_*``: the library drives a context per device; 0.1-0.2 s of start-up), the launcher route is opt-in
(``PHAMCLUST_MULTI=launcher``), and either way the CLI estimates the fill from the loaded genomes before it touches a GPU, compares
what N GPUs would save with what starting them costs (measured: ``profiles/r04/final/launch_cost.txt``), and stays on one GPU when
the job loses; the log says why.
"""

import json
import os
import time

import numpy as np

# Measured on the one-GPU box with the launcher and the gloo rehearsal transport (tools/launch_cost.py ->
# profiles/r04/final/launch_cost.txt, synth(5000,5000) -m peq): command start -> matrix on the host 2.0-2.7 s on one rank, 4.5-5.2 s as
# 2 ranks, 4.7-5.3 s as 4 -- the parent's own start, its load for this estimate and the launcher 1.7 s, the ranks' torch import and
# process group ~1.05 s -- i.e. 2.4-2.6 s of fixed cost, while the fill stage itself was no faster (the ranks shared one GPU there).  RCCL's
# communicator set-up, which gloo does not pay, is allowed one more second.  PHAMCLUST_LAUNCH_COST_S overrides.
LAUNCH_COST_S = 3.5
# one GPU, sustained (bench.py / profiles/): DP cells per second of the alignment kernels, genome pairs per second of the
# set-metric kernels, bytes per second of the D2H copy into pinned memory
CELLS_PER_S = 3.3e12
SET_PAIRS_PER_S = 5.0e10
D2H_BYTES_PER_S = 5.0e10


def process_start_time():
    """Epoch seconds at which this PROCESS was created (the interpreter's own start-up included)."""
    try:
        import psutil
        return psutil.Process().create_time()
    except Exception:                                          # noqa: BLE001
        return time.time()


class Timeline:
    """Stage stamps of one CLI run, logged as one parseable line: ``timing: {json}`` (tools/launch_cost.py reads it)."""

    def __init__(self):
        self.t_process = process_start_time()
        self.t_launch = float(os.environ.get("PHAMCLUST_T0", self.t_process))      # when the user's command started (the parent, under --gpus N)
        self.stamps = [("interpreter_start", self.t_process - self.t_launch)]
        self._last = self.t_process

    def mark(self, name):
        now = time.time()
        self.stamps.append((name, now - self._last))
        self._last = now

    def as_dict(self):
        d = {name: round(seconds, 4) for name, seconds in self.stamps}
        d["total_since_command_start"] = round(time.time() - self.t_launch, 4)
        return d

    def line(self):
        return "timing: " + json.dumps(self.as_dict())


# ... and of the in-process route (pc_multi_*: a context, an upload and a host thread per extra device, all in parallel; no
# launcher, no second interpreter, no process group).  Measured the same way (profiles/r04/final/launch_cost.txt; the "devices"
# were contexts on ONE GPU, so their uploads took turns): 0.08 s for 2 devices, 0.23 s for 4.
INPROCESS_COST_S = 0.25


def multi_gpu_route():
    """How ``phamclust --gpus N`` spreads the fill: "process" (default: this process drives all N GPUs through ``pc_multi_*``) or
    "launcher" (``PHAMCLUST_MULTI=launcher``: N ranks under torch.distributed.run, one RCCL gather -- the route ``bench.py``'s
    N > 1 contract measures)."""
    route = os.environ.get("PHAMCLUST_MULTI", "process")
    if route not in ("process", "launcher"):
        raise ValueError(f"PHAMCLUST_MULTI={route!r}: expected 'process' or 'launcher'")
    return route


def launch_cost_seconds(n_gpus, route="launcher"):
    """Fixed cost of spreading a fill over ``n_gpus`` devices instead of one."""
    if n_gpus <= 1:
        return 0.0
    env = os.environ.get("PHAMCLUST_LAUNCH_COST_S")
    return float(env) if env else (LAUNCH_COST_S if route == "launcher" else INPROCESS_COST_S)


def alignment_cells(packed):
    """Upper estimate of the DP cells an aai / peq fill aligns: every gene against every gene of the same pham in another
    genome, sum of la * lb = (L_p^2 - sum_g L_gp^2) / 2 over the phams (metrics.py:203-224; the anchor rule only drops
    paralog-against-paralog repeats).  Host arithmetic on the packed arrays -- no GPU."""
    lens = np.diff(packed.seq_off).astype(np.float64)
    pham = np.asarray(packed.gene_pham, dtype=np.int64)
    genome = np.repeat(np.arange(packed.n_genomes, dtype=np.int64), np.diff(packed.gene_off))
    per_pham = np.bincount(pham, weights=lens, minlength=packed.n_phams)
    key = genome * packed.n_phams + pham
    _, inverse = np.unique(key, return_inverse=True)
    per_entry = np.bincount(inverse, weights=lens)
    return float((np.square(per_pham).sum() - np.square(per_entry).sum()) / 2.0)


def estimate_fill_seconds(packed, metric):
    """Seconds ONE GPU needs for the whole matrix of ``metric`` (kernels + D2H), from the genomes alone."""
    pairs = packed.n_genomes * (packed.n_genomes - 1) / 2.0
    seconds = pairs * 8.0 / D2H_BYTES_PER_S + pairs / SET_PAIRS_PER_S
    if metric in ("aai", "peq"):
        seconds += alignment_cells(packed) / CELLS_PER_S
    return seconds


def choose_gpus(requested, packed, metric, route="launcher"):
    """How many GPUs to use for ``--gpus requested``: (n, reason).  N devices split the fill N ways at best and cost
    launch_cost_seconds(N, route) before they start; when that is more than they save, one GPU is faster."""
    requested = max(1, int(requested))
    if requested == 1:
        return 1, "one GPU requested"
    if os.environ.get("PHAMCLUST_FORCE_GPUS"):
        return requested, "PHAMCLUST_FORCE_GPUS is set: no estimate"
    one = estimate_fill_seconds(packed, metric)
    saved = one * (1.0 - 1.0 / requested)
    cost = launch_cost_seconds(requested, route)
    what = "interpreters, torch, process group" if route == "launcher" else "a context and an upload per device"
    if saved <= cost:
        return 1, (f"estimated {metric} fill of {packed.n_genomes} genomes on one GPU: {one:.2f} s; {requested} GPUs would save at most "
                   f"{saved:.2f} s and cost ~{cost:.2f} s to start ({what}): running on ONE GPU")
    return requested, (f"estimated {metric} fill on one GPU: {one:.2f} s; {requested} GPUs save up to {saved:.2f} s for ~{cost:.2f} s of start-up ({what})")
