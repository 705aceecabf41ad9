"""Multi-GPU fill: one process per GPU, static shard of the pair list, ONE gather.

Every pair (s, t), s < t, is independent (the reference treats them as an unordered task
bag, matrix.py:479-491).  Rank r of ``world`` is dealt target genomes t boustrophedon-wise
(0..w-1, w-1..0, ...), which balances the linear cost ramp in t; it fills the pairs of its
targets into a shard-local buffer on its own GPU, and a single ``torch.distributed.gather``
(RCCL over xGMI; every rank sends ``shard_stride`` doubles) brings the shards to rank 0,
where a device kernel permutes them into scipy condensed order.  No other collective.

The index arithmetic below mirrors ``apply_shard`` / ``k_assemble`` in csrc (closed forms),
so that host code and CPU tests can reason about the layout without a GPU.
"""

import os
import socket
import subprocess
import sys

import numpy as np


def shard_targets(n_genomes, rank, world):
    """Target genomes owned by ``rank``, ascending."""
    out = []
    j = 0
    while j * world < n_genomes:
        pos = (world - 1 - rank) if (j & 1) else rank
        t = j * world + pos
        if t < n_genomes:
            out.append(t)
        j += 1
    return np.asarray(out, dtype=np.int64)


def shard_layout(n_genomes, rank, world):
    """(owned targets, lbase) with shard-local index of pair (s, owned[k]) = lbase[k] + s."""
    owned = shard_targets(n_genomes, rank, world)
    lbase = np.zeros(owned.shape[0] + 1, dtype=np.int64)
    np.cumsum(owned, out=lbase[1:])
    return owned, lbase


def shard_pairs(n_genomes, rank, world):
    return int(shard_targets(n_genomes, rank, world).sum())


def shard_stride(n_genomes, world):
    """Equal-count gather size: the largest shard."""
    return max(shard_pairs(n_genomes, r, world) for r in range(world))


def condensed_index(n_genomes, s, t):
    """scipy condensed index of (s, t), s < t."""
    return s * n_genomes - s * (s + 1) // 2 + (t - s - 1)


def assemble_condensed_host(gathered, n_genomes, world):
    """numpy mirror of the device assembly: gathered[world, stride] -> condensed vector."""
    gathered = np.asarray(gathered).reshape(world, -1)
    out = np.empty(n_genomes * (n_genomes - 1) // 2, dtype=gathered.dtype)
    for rank in range(world):
        owned, lbase = shard_layout(n_genomes, rank, world)
        for k, t in enumerate(owned):
            t = int(t)
            if t == 0:
                continue
            s = np.arange(t, dtype=np.int64)
            out[condensed_index(n_genomes, s, t)] = gathered[rank, lbase[k]:lbase[k] + t]
    return out


def dist_mode():
    """How an aai / peq fill is spread over the ranks: "pairs" (default; each rank fills the genome pairs of its target
    genomes, one gather of matrix shards) or "alignments" (PHAMCLUST_DIST_MODE=alignments; every rank plans the whole
    fill and aligns a slice of the distinct alignments, one reduce of their results, the root builds the matrix:
    duplicates are merged across the whole job, which is what pays on collections full of identical proteins)."""
    mode = os.environ.get("PHAMCLUST_DIST_MODE", "pairs")
    if mode not in ("pairs", "alignments"):
        raise ValueError(f"PHAMCLUST_DIST_MODE={mode!r}: expected 'pairs' or 'alignments'")
    return mode


ALIGNED_METRICS = ("aai", "peq", "aai_ppos")


def force_exchange():
    """PHAMCLUST_DIST_FORCE_EXCHANGE=1: run the exchange step (gather / reduce) and the root's assembly even in a ONE-rank
    group.  A box with one GPU cannot host two RCCL ranks; this is how the `nccl` branches below -- the exact calls an
    8-GPU job makes -- execute on such a box (tests/test_gpu_parity.py::test_rccl_branches_with_one_rank)."""
    return os.environ.get("PHAMCLUST_DIST_FORCE_EXCHANGE") == "1"


def uses_alignment_slices(metric, mode=None):
    """True when a multi-rank fill of ``metric`` goes down the alignment-sliced route (one helper for fill_distributed and
    for bench.py's bookkeeping: every rank of that route reports the WHOLE job's counts, not its share)."""
    return (mode or dist_mode()) == "alignments" and metric in ALIGNED_METRICS


class _StageClock:
    """HIP-event (or, for host-staged rehearsal transports, host-clock) time of the exchange steps on this rank."""

    def __init__(self, device, on_device):
        import torch
        self.torch, self.device, self.on_device = torch, device, on_device
        self.ms = {}

    def __call__(self, name):
        return _Stage(self, name)


class _Stage:
    def __init__(self, clock, name):
        self.c, self.name = clock, name

    def __enter__(self):
        import time
        torch = self.c.torch
        if self.c.on_device:
            self.a, self.b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.a.record(torch.cuda.current_stream(self.c.device))
        else:
            torch.cuda.synchronize(self.c.device)
            self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        import time
        torch = self.c.torch
        if self.c.on_device:
            self.b.record(torch.cuda.current_stream(self.c.device))
            self.b.synchronize()
            self.c.ms[self.name] = self.a.elapsed_time(self.b)
        else:
            torch.cuda.synchronize(self.c.device)
            self.c.ms[self.name] = (time.perf_counter() - self.t0) * 1e3


def fill_distributed_alignments(ctx, metric, as_distance=True, group=None):
    """aai / peq over the job's ranks by slicing the ALIGNMENTS (see dist_mode).  Same contract as fill_distributed:
    (condensed f64 CUDA tensor on rank 0 | None elsewhere, stats of this rank's plan + slice)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    root = dist.get_global_rank(group, 0) if group is not None else 0
    ctx.set_shard(0, 1)
    device = torch.device("cuda", ctx.device_id)
    stream = torch.cuda.current_stream(device).cuda_stream
    plan = ctx.plan_dev(metric, stream)
    n = int(plan["n_distinct_alignments"])
    res = torch.empty(max(n, 1), dtype=torch.int64, device=device)       # (n_ident, aln_len) per distinct alignment
    stats = ctx.align_slice_dev(rank, world, res.data_ptr(), stream)
    stats["ms_plan"] = plan["ms_plan"]
    stats["ms_total"] = plan["ms_plan"] + stats["ms_align"]             # what THIS rank spent on the device (root: + ms_root_reduce)
    stats["dist_mode"] = "alignments"
    exchange = world > 1 or force_exchange()
    gloo = exchange and dist.get_backend(group) == "gloo"
    clock = _StageClock(device, on_device=not gloo)
    if exchange:
        with clock("ms_exchange"):
            if gloo:                                                      # rehearsal transport: staged through host memory
                host = res.cpu()
                dist.reduce(host, dst=root, op=dist.ReduceOp.SUM, group=group)
                if rank == 0:
                    res.copy_(host)
            else:
                dist.reduce(res, dst=root, op=dist.ReduceOp.SUM, group=group)  # exactly one rank holds a non-zero entry
    stats.update(clock.ms)
    stats["exchange_bytes"] = int(res.numel()) * 8
    if rank != 0:
        return None, stats
    out = torch.empty(max(ctx.n_pairs, 1), dtype=torch.float64, device=device)
    with clock("ms_root_reduce"):
        ctx.reduce_dev(metric, as_distance, res.data_ptr(), out.data_ptr(), stream)
    stats.update(clock.ms)
    stats["ms_reduce"] = clock.ms["ms_root_reduce"]
    stats["ms_total"] += clock.ms["ms_root_reduce"]
    return out[:ctx.n_pairs], stats


def fill_distributed(ctx, metric, as_distance=True, group=None, balanced=True, mode=None):
    """Sharded fill + the single gather.  ``torch.distributed`` must be initialised (backend
    "nccl" = RCCL) and ``ctx`` must hold the same uploaded genomes on every rank.
    ``balanced`` (default): target genomes are dealt by measured alignment work (``pc_set_shard_balanced``: one
    device pass per upload, identical on every rank); ``False``: the closed-form boustrophedon deal the helpers
    above describe.  Returns (condensed f64 CUDA tensor on rank 0 | None elsewhere, stats of this rank)."""
    import torch
    import torch.distributed as dist
    if uses_alignment_slices(metric, mode):
        return fill_distributed_alignments(ctx, metric, as_distance, group)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    root = dist.get_global_rank(group, 0) if group is not None else 0      # dist.gather's dst is a GLOBAL rank
    ctx.set_shard(rank, world, balanced=balanced and world > 1)
    stride = ctx.shard_stride()
    device = torch.device("cuda", ctx.device_id)                           # buffers live on the context's GPU, whatever
    stream = torch.cuda.current_stream(device).cuda_stream                 # torch's "current device" happens to be
    shard = torch.empty(max(stride, 1), dtype=torch.float64, device=device)
    stats = ctx.fill_shard_dev(metric, as_distance, shard.data_ptr(), stream)
    stats["dist_mode"] = "pairs"
    stats["shard_pairs"] = ctx.shard_pairs()
    exchange = world > 1 or force_exchange()
    gloo = exchange and dist.get_backend(group) == "gloo"
    clock = _StageClock(device, on_device=not gloo)
    if not exchange:
        gathered = shard
    else:
        with clock("ms_exchange"):
            if gloo:
                # rehearsal transport (several ranks sharing one GPU, or no RCCL): same shard / gather / assembly, staged
                # through host memory because gloo gathers CPU tensors only
                host = torch.empty(world * max(stride, 1), dtype=torch.float64) if rank == 0 else None
                dist.gather(shard.cpu(), list(host.chunk(world)) if rank == 0 else None, dst=root, group=group)
                gathered = host.to(device) if rank == 0 else None
            else:
                gathered = torch.empty(world * max(stride, 1), dtype=torch.float64, device=device) if rank == 0 else None
                dist.gather(shard, list(gathered.chunk(world)) if rank == 0 else None, dst=root, group=group)
    stats.update(clock.ms)
    stats["exchange_bytes"] = int(max(stride, 1)) * 8 * world
    if rank != 0:
        return None, stats
    out = torch.empty(max(ctx.n_pairs, 1), dtype=torch.float64, device=device)
    with clock("ms_assemble"):
        ctx.assemble_dev(gathered.data_ptr(), world, out.data_ptr(), stream)
    stats.update(clock.ms)
    return out[:ctx.n_pairs], stats


# ---------------------------------------------------------------------------------------------------------
# The product route: matrix_de_novo / the CLI under `torch.distributed.run` (one process per GPU).
# The reference parallelises inside matrix_de_novo through its `cpus` argument (matrix.py:432,471-472; flag -t,
# cli.py:113-115); here the same call shards over the ranks of the job it is running in.
# ---------------------------------------------------------------------------------------------------------
def env_world():
    """(rank, local_rank, world) as the launcher exported them; (0, 0, 1) outside a launcher."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def backend_name():
    """"nccl" (= RCCL over xGMI) unless PHAMCLUST_DIST_BACKEND says otherwise ("gloo": rehearsal transport for a box
    with fewer GPUs than ranks -- ranks share devices and the gather is staged through host memory)."""
    return os.environ.get("PHAMCLUST_DIST_BACKEND", "nccl")


def local_device():
    """HIP device ordinal of this rank: LOCAL_RANK; under the gloo rehearsal LOCAL_RANK modulo the devices present."""
    import torch
    _, local_rank, _ = env_world()
    if backend_name() != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    return int(os.environ.get("PHAMCLUST_DEVICE", local_rank))


def ensure_process_group():
    """Join the job's process group if the launcher started more than one rank and nothing has joined it yet.
    Must run before this process's first GPU call under nccl (RCCL binds the device at init).  Returns (rank, world)."""
    rank, _, world = env_world()
    if world <= 1:
        return 0, 1                                            # (one rank: torch is not even imported)
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        import datetime
        device = local_device()
        torch.cuda.set_device(device)
        # a rank that dies outside a collective leaves its peers waiting in the next one: bound that wait (the launcher
        # also tears the job down when a rank exits non-zero; this covers a rank that hangs)
        timeout = datetime.timedelta(seconds=int(os.environ.get("PHAMCLUST_DIST_TIMEOUT_S", "900")))
        if backend_name() == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device), timeout=timeout)
        else:
            dist.init_process_group(backend_name(), timeout=timeout)
    return dist.get_rank(), dist.get_world_size()


def broadcast_flag(value, src=0, error=None):
    """One small control-plane broadcast (e.g. "rank 0 found a cached matrix, nobody fills").  ``error``: the source rank
    failed while working out the flag -- every rank then raises instead of walking into a collective the source will
    never join."""
    import torch.distributed as dist
    box = [(bool(value), None if error is None else str(error))]
    dist.broadcast_object_list(box, src=src)
    flag, err = box[0]
    if err is not None:
        raise RuntimeError(f"rank {src} failed before the fill: {err}")
    return bool(flag)


def fill_condensed(ctx, metric, as_distance=True):
    """Sharded fill of the uploaded genomes over the job's ranks.  Rank 0 gets the host condensed vector and the
    stats of its shard; every other rank gets (None, stats)."""
    import torch
    out, stats = fill_distributed(ctx, metric, as_distance)
    if out is None:
        return None, stats
    host = torch.empty(out.shape, dtype=out.dtype, pin_memory=True)
    host.copy_(out, non_blocking=False)
    return host.numpy(), stats


def balanced_deal(costs, world, pair_floor=2000):
    """Host mirror of pc_set_shard_balanced (csrc/pc_api.hip): target t costs ``costs[t] + t * pair_floor`` (its DP
    cells plus a floor per pair); targets go, heaviest first (ties: lower index), to the rank with the least work so
    far (ties: lowest rank).  Returns (t_rank[N], t_lbase[N], stride): pair (s, t) lives at
    gathered[t_rank[t] * stride + t_lbase[t] + s]."""
    costs = np.asarray(costs, dtype=np.uint64)
    n = costs.shape[0]
    total = costs + np.arange(n, dtype=np.uint64) * np.uint64(pair_floor)
    order = sorted(range(n), key=lambda t: (-int(total[t]), t))
    load = [0] * world
    t_rank = np.zeros(n, dtype=np.int32)
    for t in order:
        best = min(range(world), key=lambda r: (load[r], r))
        t_rank[t] = best
        load[best] += int(total[t])
    t_lbase = np.zeros(n, dtype=np.int64)
    fill = [0] * world
    for t in range(n):
        t_lbase[t] = fill[t_rank[t]]
        fill[t_rank[t]] += t
    return t_rank, t_lbase, max(fill) if n else 0


def assemble_table_host(gathered, n_genomes, t_rank, t_lbase):
    """numpy mirror of k_assemble_table: gathered[world, stride] + the deal's tables -> condensed vector."""
    gathered = np.asarray(gathered)
    out = np.empty(n_genomes * (n_genomes - 1) // 2, dtype=gathered.dtype)
    for t in range(1, n_genomes):
        s = np.arange(t, dtype=np.int64)
        out[condensed_index(n_genomes, s, t)] = gathered[int(t_rank[t]), int(t_lbase[t]):int(t_lbase[t]) + t]
    return out


def free_port():
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def launch_ranks(n_ranks, module, argv, env=None):
    """Start ``python -m torch.distributed.run --nproc-per-node n_ranks -m <module> <argv>`` as a CHILD process and
    return its exit code.  Called before this process has touched the GPU (the CLI's ``--gpus N``): the children, not
    this process, own the devices."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), "-m", module] + list(argv)
    child_env = dict(os.environ)
    child_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    child_env.update(env or {})
    return subprocess.call(cmd, env=child_env)
