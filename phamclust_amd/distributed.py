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


def fill_distributed(ctx, metric, as_distance=True, group=None, balanced=True):
    """Sharded fill + the single gather.  ``torch.distributed`` must be initialised (backend
    "nccl" = RCCL) and ``ctx`` must hold the same uploaded genomes on every rank.
    ``balanced`` (default): target genomes are dealt by measured alignment work (``pc_set_shard_balanced``: one
    device pass per upload, identical on every rank); ``False``: the closed-form boustrophedon deal the helpers
    above describe.  Returns (condensed f64 CUDA tensor on rank 0 | None elsewhere, stats of this rank)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    ctx.set_shard(rank, world, balanced=balanced and world > 1)
    stride = ctx.shard_stride()
    device = torch.device("cuda", torch.cuda.current_device())
    stream = torch.cuda.current_stream().cuda_stream
    shard = torch.empty(max(stride, 1), dtype=torch.float64, device=device)
    stats = ctx.fill_shard_dev(metric, as_distance, shard.data_ptr(), stream)
    if world == 1:
        gathered = shard
    elif dist.get_backend(group) == "gloo":
        # rehearsal transport (several ranks sharing one GPU, or no RCCL): same shard / gather / assembly, staged
        # through host memory because gloo gathers CPU tensors only
        host = torch.empty(world * max(stride, 1), dtype=torch.float64) if rank == 0 else None
        dist.gather(shard.cpu(), list(host.chunk(world)) if rank == 0 else None, dst=0, group=group)
        gathered = host.to(device) if rank == 0 else None
    else:
        gathered = torch.empty(world * max(stride, 1), dtype=torch.float64, device=device) if rank == 0 else None
        dist.gather(shard, list(gathered.chunk(world)) if rank == 0 else None, dst=0, group=group)
    if rank != 0:
        return None, stats
    out = torch.empty(max(ctx.n_pairs, 1), dtype=torch.float64, device=device)
    ctx.assemble_dev(gathered.data_ptr(), world, out.data_ptr(), stream)
    return out[:ctx.n_pairs], stats
