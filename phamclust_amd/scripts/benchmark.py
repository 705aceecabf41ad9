"""``python -m phamclust_amd.scripts.benchmark <infile> <outdir> [-m METRIC]`` -- the reference's scaling protocol
(scripts/benchmark.py:15-23, 96-113) for this build: seed 42, samples of 2,000 / 1,000 / 500 genomes of the input, each
sample filled ``--iterations`` times with the chosen metric, the matrix written in squareform, one
``Genomes / GPUs / Iter / Elapsed`` line per fill.

Where the reference sweeps worker processes (CPUS = 1, 2, 4, 6, 8, 12, 16), this sweeps what parallelises the fill here:
GPUs.  A single process drives one GPU, so the sweep is [1] unless the script runs under
``python -m torch.distributed.run --nproc-per-node N -m phamclust_amd.scripts.benchmark ...``, where every fill is
sharded over the N ranks (``matrix_de_novo`` does that by itself) and rank 0 reports.  Sample files and matrices carry
the metric's NAME (the reference formats the function object into the file name, scripts/benchmark.py:103-104).
"""

import argparse
import datetime
import pathlib
import random
import sys

from phamclust_amd import distributed
from phamclust_amd.cli import METRICS
from phamclust_amd.matrix import LAST_FILL, matrix_de_novo, matrix_to_squareform
from phamclust_amd.scripts.phamclust import load_genomes_from_tsv

SAMPLE_SIZES = [2000, 1000, 500]
ITER = 3
SEED = 42


def parse_args(argv=None):
    p = argparse.ArgumentParser(description=__doc__, prog="phamclust-benchmark", formatter_class=argparse.RawTextHelpFormatter)
    p.add_argument("infile", type=pathlib.Path, help="TSV mapping phages to phams and translations")
    p.add_argument("outdir", type=pathlib.Path, help="where the sample files and matrices are written")
    p.add_argument("-m", "--metric", type=str, default="peq", choices=METRICS.keys(), help="metric of the fills [default: %(default)s]")
    p.add_argument("-i", "--iterations", type=int, default=ITER, help="fills per sample [default: %(default)s]")
    p.add_argument("-s", "--seed", type=int, default=SEED, help="random seed of the sampling [default: %(default)s]")
    return p.parse_args(argv)


def dump_genomes_to_tsv(genomes, filepath):
    with open(filepath, "w") as writer:
        for genome in genomes:
            for pham, translations in genome:
                for translation in translations:
                    writer.write(f"{genome.name}\t{pham}\t{translation}\n")
    return filepath


def main(argv=None):
    if argv is None and len(sys.argv) == 1:
        sys.argv.append("-h")
    args = parse_args(argv)
    metric = METRICS[args.metric]
    random.seed(args.seed)
    rank, world = distributed.ensure_process_group()
    if rank == 0:
        args.outdir.mkdir(parents=True, exist_ok=True)
    genomes = load_genomes_from_tsv(args.infile)
    for k in SAMPLE_SIZES:
        if k > len(genomes):
            if rank == 0:
                print(f"Genomes: {k}: skipped, the input holds {len(genomes)}")
            continue
        sample = sorted(random.sample(population=genomes, k=k), key=lambda g: g.name)      # every rank draws the same sample
        if rank == 0:
            dump_genomes_to_tsv(sample, args.outdir / f"{k}_genomes.tsv")
        for i in range(args.iterations):
            t_start = datetime.datetime.now()
            matrix = matrix_de_novo(sample, metric, world)
            t_stop = datetime.datetime.now()
            if rank != 0:
                continue
            matrix_to_squareform(matrix, args.outdir / f"{k}_genomes-{world}_gpus-iter_{i}-pairwise_{args.metric}_distances.tsv")
            print(f"Genomes: {k}, GPUs: {world}, Iter: {i}, Elapsed: {str(t_stop - t_start)} "
                  f"(upload {LAST_FILL.get('upload_s', 0.0):.3f} s, fill {LAST_FILL.get('fill_s', 0.0):.3f} s, "
                  f"kernels {LAST_FILL.get('ms_total', 0.0):.1f} ms)", flush=True)
    if world > 1:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
