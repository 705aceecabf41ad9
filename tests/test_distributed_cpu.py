"""The N>1 path on CPU: shard arithmetic, and a world_size-2 (and 3) gloo run in which each
rank fills its shard, one gather brings them to rank 0, and the assembled matrix equals the
unsharded one.  The per-rank "device" here is the oracle (this is a test of the host-side
sharding / gather / assembly logic, which is pure index arithmetic)."""

import os
import socket

import numpy as np
import pytest

from conftest import REPO


def test_shard_partition_properties():
    from phamclust_amd import distributed as D
    for n in (1, 2, 5, 8, 17, 64, 101):
        for world in (1, 2, 3, 4, 8):
            seen = np.zeros(n, dtype=int)
            tot = 0
            for r in range(world):
                owned, lbase = D.shard_layout(n, r, world)
                seen[owned] += 1
                assert (np.diff(owned) > 0).all() and lbase[-1] == owned.sum() == D.shard_pairs(n, r, world)
                tot += lbase[-1]
            assert (seen == 1).all() and tot == n * (n - 1) // 2
            assert D.shard_stride(n, world) >= (n * (n - 1) // 2 + world - 1) // world
    # boustrophedon balances the linear ramp: shards within ~1 column of each other
    sizes = [D.shard_pairs(5000, r, 8) for r in range(8)]
    assert max(sizes) - min(sizes) <= 5000


def test_assemble_host_is_a_permutation():
    from phamclust_amd import distributed as D
    n, world = 23, 4
    stride = D.shard_stride(n, world)
    gathered = np.full((world, stride), -1.0)
    for r in range(world):
        owned, lbase = D.shard_layout(n, r, world)
        for k, t in enumerate(owned):
            for s in range(int(t)):
                gathered[r, lbase[k] + s] = D.condensed_index(n, s, int(t))
    out = D.assemble_condensed_host(gathered, n, world)
    assert np.array_equal(out, np.arange(n * (n - 1) // 2, dtype=float))


def _worker(rank, world, port, metric, q):
    import sys
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    from phamclust_amd import distributed as D
    from phamclust_amd.synth import synth_packed
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        packed = synth_packed(31, 300, seed=4)
        n = packed.n_genomes
        owned, lbase = D.shard_layout(n, rank, world)
        stride = D.shard_stride(n, world)
        shard = torch.zeros(stride, dtype=torch.float64)
        for k, t in enumerate(owned):                       # this rank's pairs, shard-local order
            for s in range(int(t)):
                shard[lbase[k] + s] = O.pair(packed, metric, s, int(t), as_distance=True)
        gathered = torch.empty(world * stride, dtype=torch.float64) if rank == 0 else None
        dist.gather(shard, list(gathered.chunk(world)) if rank == 0 else None, dst=0)   # the ONE collective
        if rank == 0:
            got = D.assemble_condensed_host(gathered.numpy(), n, world)
            want = O.fill(packed, metric, as_distance=True)
            q.put(bool(np.array_equal(got, want)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_shard_gather_assemble(native_built, world):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, "jc", q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert q.get(timeout=10) is True
