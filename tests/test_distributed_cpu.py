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


def _balanced_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    from phamclust_amd import distributed as D
    from phamclust_amd.synth import synth_packed
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      PHAMCLUST_DIST_BACKEND="gloo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        packed = synth_packed(37, 300, seed=9)
        n = packed.n_genomes
        costs = _target_cells(packed)          # DP cells behind each target genome, as the device pass counts them
        t_rank, t_lbase, stride = D.balanced_deal(costs, world)
        assert (np.bincount(t_rank, minlength=world) > 0).all()
        shard = torch.zeros(max(stride, 1), dtype=torch.float64)
        for t in range(1, n):
            if t_rank[t] == rank:
                for s_ in range(t):
                    shard[t_lbase[t] + s_] = O.pair(packed, "peq", s_, t, as_distance=True)
        gathered = torch.empty(world * max(stride, 1), dtype=torch.float64) if rank == 0 else None
        dist.gather(shard, list(gathered.chunk(world)) if rank == 0 else None, dst=0)
        flag = D.broadcast_flag(rank == 0 and True, src=0)          # the pipeline's one control-plane broadcast
        if rank == 0:
            got = D.assemble_table_host(gathered.numpy().reshape(world, -1), n, t_rank, t_lbase)
            want = O.fill(packed, "peq", as_distance=True)
            loads = [int(costs[t_rank == r].sum()) for r in range(world)]
            q.put((bool(np.array_equal(got, want)), flag, max(loads) / (sum(loads) / world)))
    finally:
        dist.destroy_process_group()


def _target_cells(packed):
    """sum over s < t of the DP cells of pair (s, t): per shared pham, (summed gene length in s) x (summed gene length in t)."""
    n = packed.n_genomes
    length = {}                                             # (genome, pham) -> summed translation length
    for g in range(n):
        for k in range(int(packed.gene_off[g]), int(packed.gene_off[g + 1])):
            key = (g, int(packed.gene_pham[k]))
            length[key] = length.get(key, 0) + int(packed.seq_off[k + 1] - packed.seq_off[k])
    by_genome = [{} for _ in range(n)]
    for (g, pham), ln in length.items():
        by_genome[g][pham] = ln
    costs = np.zeros(n, dtype=np.uint64)
    for t in range(1, n):
        costs[t] = sum(ln * by_genome[s][pham] for s in range(t) for pham, ln in by_genome[t].items() if pham in by_genome[s])
    return costs


def test_gloo_world2_balanced_deal(native_built):
    """World-2 gloo run of the COST-BALANCED deal (pc_set_shard_balanced's host mirror, distributed.balanced_deal):
    each rank fills the targets it was dealt, one gather, table-driven assembly == the unsharded matrix, and the
    dealt alignment work is level."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_balanced_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    equal, flag, imbalance = q.get(timeout=10)
    assert equal is True and flag is True and imbalance < 1.05


def test_balanced_deal_mirror_properties():
    from phamclust_amd import distributed as D
    rng = np.random.default_rng(3)
    for n, world in ((1, 2), (7, 3), (200, 8), (53, 64)):
        costs = rng.integers(0, 10**9, n).astype(np.uint64)
        t_rank, t_lbase, stride = D.balanced_deal(costs, world)
        fill = [0] * world
        for t in range(n):
            assert t_lbase[t] == fill[t_rank[t]]
            fill[t_rank[t]] += t
        assert stride == max(fill) and sum(fill) == n * (n - 1) // 2
    t_rank, _, _ = D.balanced_deal(np.array([5, 5, 5, 5], dtype=np.uint64), 2, pair_floor=0)
    assert t_rank.tolist() == [0, 1, 0, 1]                  # ties: lower index first, lowest rank first


def test_dist_mode_switch(monkeypatch):
    """PHAMCLUST_DIST_MODE picks how an aai / peq fill is split over the ranks; anything else is refused."""
    from phamclust_amd import distributed as D
    monkeypatch.delenv("PHAMCLUST_DIST_MODE", raising=False)
    assert D.dist_mode() == "pairs"
    monkeypatch.setenv("PHAMCLUST_DIST_MODE", "alignments")
    assert D.dist_mode() == "alignments"
    monkeypatch.setenv("PHAMCLUST_DIST_MODE", "rows")
    with pytest.raises(ValueError):
        D.dist_mode()


# ---------------------------------------------------------------------------------------------------------
# `python bench.py --gpus N` by itself: the parent starts the ranks as a CHILD launcher before it has touched a GPU
# (it replaces the reference's in-process worker pool, matrix.py:471-472,488-491, whose parent also only hands out work).
# PC_BENCH_CHILD_PROBE=1 makes the ranks stop after joining their (gloo) group, so the plumbing runs where there is no GPU.
# ---------------------------------------------------------------------------------------------------------
def _run_bench_parent(extra_env, *argv):
    import json
    import subprocess
    import sys
    env = dict(os.environ, PC_BENCH_CHILD_PROBE="1", PC_BENCH_BACKEND="gloo", **extra_env)
    for key in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    done = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=300)
    lines = [json.loads(x) for x in done.stdout.splitlines() if x.startswith("{")]
    return done, lines


def test_bench_parent_spawns_the_ranks_as_a_child_without_touching_the_gpu():
    done, lines = _run_bench_parent({}, "--gpus", "2", "--steps", "1", "--warmup", "0", "--genomes", "64")
    assert done.returncode == 0, done.stderr[-2000:]
    assert len(lines) == 1, done.stdout                         # ONE line on the parent's stdout: rank 0's, relayed
    line = lines[0]
    assert line["ranks_seen"] == 2 and line["n_gpus"] == 2
    assert sorted(r["rank"] for r in line["ranks"]) == [0, 1] and len({r["pid"] for r in line["ranks"]}) == 2
    for r in line["ranks"]:                                     # the parent's own arguments reached every rank unchanged
        assert r["argv"] == ["--gpus", "2", "--steps", "1", "--warmup", "0", "--genomes", "64"]
    by = line["launched_by"]
    assert by["parent_modules_touching_gpu"] == [] and by["child_exit_status"] == 0
    assert "torch.distributed.run" in by["how"]


def test_bench_parent_relays_a_failing_rank():
    done, lines = _run_bench_parent({"PC_BENCH_CHILD_PROBE_FAIL": "1"}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert done.returncode != 0
    assert lines and lines[0]["launched_by"]["child_exit_status"] != 0


def test_bench_parent_source_makes_no_gpu_call():
    """spawn_ranks itself: no torch, no hip binding, no build -- read off its source."""
    import ast
    src = open(os.path.join(REPO, "bench.py")).read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "spawn_ranks")
    imported = set()
    for node in ast.walk(fn):
        if isinstance(node, ast.Import):
            imported |= {a.name for a in node.names}
        if isinstance(node, ast.ImportFrom):
            imported.add(node.module)
    assert imported == {"subprocess", "phamclust_amd.distributed"}, imported
    main = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "main")
    first_import = min(n.lineno for n in ast.walk(main) if isinstance(n, (ast.Import, ast.ImportFrom)))
    spawn_call = min(n.lineno for n in ast.walk(main) if isinstance(n, ast.Call) and getattr(n.func, "id", "") == "spawn_ranks")
    assert spawn_call < first_import                             # main() hands over before it imports anything
