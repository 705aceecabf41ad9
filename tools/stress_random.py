#!/usr/bin/env python3
"""Randomised stress of all six metrics (and aai with ppos=True) against the oracle on many small collections (paralogs, byte-identical
sequences, odd residues, tie-heavy 3-letter sequences, lengths up to 1,500), every collection under a randomly chosen
row of the tie-rule table (kernel and oracle switched together): `python tools/stress_random.py SEED TRIALS`.
r01: seed 4242, 1,500 collections, 9,000 fills, 0 mismatches; seed 20261004, 4,000 collections, 24,000 fills, 0 mismatches.
r02 (64-bit lexicographic-max cell, all 8 rules): see profiles/r02/final/stress.txt.
r03: every collection additionally draws a plan budget (aai / peq fills in one piece or in many chunks), a two-part or a full
upload, both popcount tile kernels (PC_POPC_TILE, read per launch), every pocp / af kernel (PC_SET_KERNEL) and, every fourth one, a shard of a 2- or 3-rank deal compared with
the same pairs of the unsharded matrix: see profiles/r03/final/stress.txt.
r04: one collection in twelve holds genes of 4,100-8,100 residues (strip-mined passes on the wide and the narrow variants, the
percent-positives passes), PC_S64_CHUNKS is drawn per collection, and buckets of one or
two rows -- most of what these tiny collections hold -- take the one- / two-wave workgroups and the tier launches; the strip-mined
launches run one row per wave or pipelined over 2 / 3 / 8 waves (PC_PIPE, drawn per collection): see profiles/r04/final/stress.txt."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from phamclust_amd import hip
from phamclust_amd.genome import Genome
from phamclust_amd.pack import pack_genomes
ctx = hip.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 777)
letters = np.array(list("ACDEFGHIKLMNPQRSTVWY" * 3 + "BZX*Uacdw-"))
few = np.array(list("AGS"))                                  # co-optimal alignments in almost every pair
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 600):
    n_genomes, n_phams = int(rng.integers(2, 14)), int(rng.integers(1, 9))
    maxlen = int(rng.choice([12, 90, 400, 1500]))
    long_genes = trial % 12 == 11                            # r04: column genes beyond the widest variant's 4,096 columns
    alpha = few if rng.random() < 0.4 else letters
    pool = ["".join(alpha[rng.integers(0, alpha.size, int(rng.integers(1, maxlen + 1)))]) for _ in range(int(rng.integers(1, 10)))]
    if long_genes:
        pool += ["".join(alpha[rng.integers(0, alpha.size, int(rng.integers(4100, 8101)))]) for _ in range(2)]
    rule = int(rng.integers(0, 8))
    ctx.set_tie_rule(rule); O.set_tie_rule(rule)
    genomes = []
    for g in range(n_genomes):
        genome = Genome(f"g{g:02d}")
        for p in rng.permutation(n_phams)[:int(rng.integers(1, n_phams + 1))]:
            for _ in range(int(rng.integers(1, 5)) if rng.random() < 0.3 else 1):
                seq = pool[int(rng.integers(0, len(pool)))]
                if rng.random() < 0.5:
                    cut = int(rng.integers(0, len(seq)))
                    seq = seq[:cut] + "W" + seq[cut + 1:] if rng.random() < 0.5 else seq[:max(1, cut)]
                genome.add(f"pham{p}", seq)
        genomes.append(genome)
    packed = pack_genomes(genomes)
    ctx.upload(packed, residues=bool(rng.random() < 0.5))              # two-part upload: the residues follow on demand
    ctx.set_plan_budget(int(rng.choice([0, 56, 56 * 7, 56 * 60, 56 * 2000])))   # 0: automatic; tiny: one target genome per chunk
    os.environ["PC_POPC_TILE"] = str(rng.choice(["32", "64"]))
    os.environ["PC_SET_KERNEL"] = str(rng.choice(["popc", "sparse", "sparse64", "sparsecol", "walker"]))   # pocp / af kernel, read per fill
    os.environ["PC_S64_CHUNKS"] = str(rng.choice(["1", "2", "3"]))
    os.environ["PC_PIPE"] = str(rng.choice(["", "0", "2", "3", "8"]))       # strip-mined launches: the launcher's choice, one row per wave, pipelined over n waves
    for metric in ("gcs", "jc", "pocp", "af", "aai", "peq", "aai_ppos"):
        got = ctx.fill(metric, as_distance=bool(trial & 1))
        want = O.fill(packed, metric, as_distance=bool(trial & 1))
        if not np.array_equal(got, want):
            bad += 1; print("MISMATCH", trial, metric, "rule", rule, flush=True)
    if trial % 4 == 3 and n_genomes >= 3:                               # a shard of a multi-rank deal == the same pairs of the matrix
        import torch
        world = int(rng.integers(2, 4)); rank = int(rng.integers(0, world)); metric = str(rng.choice(["jc", "pocp", "af", "peq"]))
        want = ctx.fill(metric, as_distance=True)
        ctx.set_shard(rank, world, balanced=bool(rng.random() < 0.5))
        stride = ctx.shard_stride()
        buf = torch.full((max(stride, 1),), -1.0, dtype=torch.float64, device="cuda:0")
        ctx.fill_shard_dev(metric, True, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        t_rank, t_lbase = ctx.shard_table()
        got = buf.cpu().numpy()
        n = packed.n_genomes
        for t in range(1, n):
            if t_rank[t] == rank:
                s_idx = np.arange(t)
                if not np.array_equal(got[t_lbase[t]:t_lbase[t] + t], want[s_idx * n - s_idx * (s_idx + 1) // 2 + (t - s_idx - 1)]):
                    bad += 1; print("SHARD MISMATCH", trial, metric, rank, world, flush=True); break
        ctx.set_shard(0, 1)
    if trial % 100 == 99: print("trial", trial + 1, "mismatches", bad, flush=True)
print("done, mismatches", bad)
sys.exit(1 if bad else 0)
