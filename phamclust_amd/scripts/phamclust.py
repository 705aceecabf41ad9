"""The phamclust pipeline around the GPU matrix fill (reference scripts/phamclust.py:140-545).

Same stages, file names and formats as the reference so that outputs and the on-disk cache
interoperate: load genomes -> sort by name -> md5 of the FASTA text -> ``{md5}.tmp`` ->
``01_genomes/*.fasta`` -> ``02_distmats/{metric}_distance_matrix.tsv`` (lower triangle; reused when
present) -> integrity check -> three rounds of hierarchical clustering -> per-cluster directories,
similarity matrices, heatmaps (when plotly is available) -> pairwise similarity outputs.
The only change is stage 2: ``matrix_de_novo`` runs on the GPU.
"""

import datetime
import hashlib
import logging
import pathlib
import shutil
import sys

from phamclust_amd.cli import METRICS, parse_args
from phamclust_amd.clustering import hierarchical_clustering
from phamclust_amd.genome import Genome
from phamclust_amd.heatmap import CSS_COLORS, draw_heatmap
from phamclust_amd.matrix import matrix_de_novo, matrix_from_squareform, matrix_to_adjacency, matrix_to_squareform

LOG_STR_FMT = "phamclust: %(asctime)s.%(msecs)03d: %(levelname)s: %(message)s"
LOG_TIME_FMT = "%H:%M:%S"
FASTA_SUFFIXES = (".fasta", ".faa", ".fa")


def load_genomes_from_tsv(filepath):
    """2- or 3-column TSV -> list[Genome] in first-appearance order (2 columns: translation "M")."""
    genomes = dict()
    with open(filepath, "r") as handle:
        for line in handle:
            row = line.rstrip().split("\t")
            if len(row) == 3:
                name, pham, translation = row
            elif len(row) == 2:
                (name, pham), translation = row, "M"
            else:
                raise ValueError("input file must either 2 or 3 columns")
            if name not in genomes:
                genomes[name] = Genome(name)
            genomes[name].add(pham, translation)
    return list(genomes.values())


def load_genomes_from_fasta_dir(filepath):
    """One FASTA per genome, headers ``>name=..|pham=..|n=..``; the file stem names the genome."""
    genomes = dict()
    for f in filepath.iterdir():
        if f.suffix not in FASTA_SUFFIXES:
            logging.debug(f"{f} does not appear to be FASTA - skipping")
            continue
        if f.is_dir():
            logging.debug(f"{f} appears to be a directory - skipping")
            continue
        if f.stem in genomes:
            raise ValueError(f"duplicate genome name detected {f.stem}")
        genomes[f.stem] = Genome(f.stem)
        genomes[f.stem].load(f)
    return list(genomes.values())


def _hash_genomes(genomes):
    """md5 over the FASTA text of the genomes in the given (name-sorted) order."""
    digest = hashlib.new("md5")
    for genome in genomes:
        digest.update(str(genome).encode())
    return digest.hexdigest()


def check_matrix_integrity(matrix):
    """(edge-count error, diagonal-sum error, unfilled edges) - all zero for a sound matrix
    (reference scripts/phamclust.py:109-137).  The array-backed matrix always holds N(N-1)/2 edge slots,
    so the first term can only be zero; unset slots (NaN) are what can still go wrong."""
    import numpy as np
    n = len(matrix)
    full = matrix.to_ndarray()
    upper = full[np.triu_indices(n, k=0)]
    unfilled = int(np.isnan(upper).sum())
    diag_sum = float(np.nansum(full.diagonal()))
    return 0, diag_sum - (1.0 - matrix.is_distance) * n, unfilled


def _mkdir(path):
    if not path.is_dir():
        path.mkdir()
    return path


def _fresh_dir(path):
    if path.is_dir():
        shutil.rmtree(path)
    path.mkdir()
    return path


def phamclust(infile, outdir, is_genome_dir, metric, nr_distance, nr_linkage, clu_distance, clu_linkage, sub_distance,
              sub_linkage, k_min, no_sub, colors, midpoint, cpus, rm_tmp, debug):
    if nr_distance >= clu_distance:
        nr_distance = 0.0
    log = logging.info
    log("=======================")
    log(" 0: runtime parameters ")
    log("=======================")
    for key, value in (("infile", infile), ("outdir", outdir), ("debug", debug), ("subcluster", not no_sub),
                       ("remove tmp", rm_tmp), ("sub dist", sub_distance), ("sub link", sub_linkage),
                       ("clu dist", clu_distance), ("clu link", clu_linkage), ("nr dist", nr_distance),
                       ("nr link", nr_linkage), ("metric", metric), ("cpus", cpus), ("colors", ",".join(colors)),
                       ("midpoint", midpoint)):
        log(f"{key + ':':<11} {value}")

    log("====================")
    log(" 1: parsing genomes ")
    log("====================")
    if is_genome_dir:
        genomes = load_genomes_from_fasta_dir(infile)
        log(f"loaded {len(genomes)} genomes from input directory")
    else:
        genomes = load_genomes_from_tsv(infile)
        log(f"loaded {len(genomes)} genomes from input TSV")
    genomes.sort(key=lambda g: g.name)
    by_name = {g.name: g for g in genomes}
    hashsum = _hash_genomes(genomes)
    log(f"md5 hashsum is {hashsum}")
    tmpdir = _mkdir(outdir.joinpath(f"{hashsum}.tmp"))
    log(f"using temp directory {tmpdir.name}")
    tmp_genomes = _mkdir(tmpdir.joinpath("01_genomes"))
    for genome in genomes:
        fasta = tmp_genomes.joinpath(f"{genome.name}.fasta")
        if not fasta.is_file():
            genome.save(fasta)
    log(f"genomes stashed in {tmpdir.name}/{tmp_genomes.name}")

    log("==========================")
    log(" 2: build distance matrix ")
    log("==========================")
    log(f"selected metric: {metric}")
    tmp_distmats = _mkdir(tmpdir.joinpath("02_distmats"))
    dist_file = tmp_distmats.joinpath(f"{metric}_distance_matrix.tsv")
    start = datetime.datetime.now()
    if not dist_file.is_file():
        log("cached distance matrix not found - computing de novo")
        dist_mat = matrix_de_novo(genomes, METRICS[metric], cpus)
        log(f"computed distance matrix in {datetime.datetime.now() - start}")
        log("caching distance matrix so it can be re-used")
        matrix_to_squareform(dist_mat, dist_file, lower_triangle=True)
    else:
        log("found cached distance matrix - importing it")
        dist_mat = matrix_from_squareform(dist_file)
        log(f"loaded distance matrix in {datetime.datetime.now() - start}")
    if not dist_mat.is_distance:
        log("matrix is not a distance matrix - flipping it")
        dist_mat.invert()
    status = check_matrix_integrity(dist_mat)
    if not any(status):
        log("matrix passed all integrity checks")
    else:
        logging.error("matrix failed the following integrity check(s):")
        if status[0]:
            logging.error(f"found {abs(status[0])} too {'many' if status[0] > 0 else 'few'} edges")
        if status[1] != 0:
            logging.error(f"diagonal sum is off by {status[1]}")
        if status[2] != 0:
            logging.error(f"found {status[2]} unfilled edges")
        print("matrix validation failed - check log for details")
        sys.exit(1)

    log("====================")
    log(" 3: cluster genomes ")
    log("====================")
    log(f"grouping highly redundant genomes with distance <= {nr_distance} by {nr_linkage} linkage")
    seeds = hierarchical_clustering(dist_mat, eps=nr_distance, linkage=nr_linkage)
    seed_map = {m.medoid[0]: m for m in seeds}
    repr_mat = dist_mat.extract_submatrix(list(seed_map.keys()))
    log(f"found {len(repr_mat)} groups of similar genomes")
    log(f"clustering non-redundant genomes with distance <= {clu_distance} by {clu_linkage} linkage")
    clu_mats = hierarchical_clustering(repr_mat, eps=clu_distance, linkage=clu_linkage)
    for i, clu_mat in enumerate(clu_mats):
        nodes = []
        for representative in clu_mat.nodes:
            nodes.extend(seed_map[representative].nodes)
        clu_mats[i] = dist_mat.extract_submatrix(nodes)
    single_mats = [m for m in clu_mats if len(m) == 1]
    clu_mats = [m for m in clu_mats if len(m) > 1]
    log(f"found {len(clu_mats)} clusters and {len(single_mats)} singletons")
    tmp_clusters = _mkdir(tmpdir.joinpath("03_clusters"))

    log("========================")
    log(" 4: sub-cluster genomes ")
    log("========================")
    for i, clu_mat in enumerate(sorted(clu_mats, reverse=True)):
        log(f"cluster {i + 1} has {len(clu_mat)} nodes")
        cluster_dir = _fresh_dir(tmp_clusters.joinpath(f"cluster_{i + 1}"))
        genome_dir = _mkdir(cluster_dir.joinpath("genomes"))
        members = set(clu_mat.nodes)
        for genome in genomes:
            if genome.name in members:
                genome.save(genome_dir.joinpath(f"{genome.name}.faa"))
        matfile = cluster_dir.joinpath(f"{metric}_similarity.tsv")
        if no_sub or len(clu_mat) < k_min:
            logging.debug(f"not sub-clustering {len(clu_mat)} genomes")
            clu_mat.reorder()
        else:
            order = []
            sub_mats = sorted(hierarchical_clustering(clu_mat, eps=sub_distance, linkage=sub_linkage), reverse=True)
            for j, sub_mat in enumerate(sub_mats):
                if len(sub_mat) > 2:
                    sub_mat.reorder()
                order.extend(sub_mat.nodes)
                sub_mat.invert()
                matrix_to_squareform(sub_mat, cluster_dir.joinpath(f"subcluster_{j + 1}_similarity.tsv"))
            clu_mat.reorder(order)
        clu_mat.invert()
        matrix_to_squareform(clu_mat, matfile)
        draw_heatmap(clu_mat, colors=colors, midpoint=midpoint, filename=cluster_dir.joinpath(f"{metric}_heatmap.svg"))
        draw_heatmap(clu_mat, colors=colors, midpoint=midpoint, filename=cluster_dir.joinpath(f"{metric}_heatmap.html"))

    if single_mats:
        genome_dir = _mkdir(_fresh_dir(tmp_clusters.joinpath("singletons")).joinpath("genomes"))
        for single in single_mats:
            node = single.nodes[0]
            by_name[node].save(genome_dir.joinpath(f"{node}.faa"))

    log("========================")
    log(" 5: draw dataset heatmap")
    log("========================")
    log("putting matrix in cluster order")
    order = []
    for clu_mat in sorted(clu_mats, reverse=True):
        order.extend(clu_mat.nodes)
    for single in single_mats:
        order.extend(single.nodes)
    dist_mat.reorder(order)
    log("cast to similarity matrix for easier visualization")
    dist_mat.invert()
    if len(dist_mat) > 1500:
        log("full pairwise matrix is too large to visualize")
    else:
        for suffix in ("html", "svg"):
            target = tmp_clusters.joinpath(f"{metric}_heatmap.{suffix}")
            log(f"drawing heatmap {suffix.upper()} and saving to {target}")
            draw_heatmap(dist_mat, colors=colors, midpoint=1.0 - clu_distance, filename=target)

    log("========================")
    log(" 6: move output files   ")
    log("========================")
    log(f"removing contents from existing output directory {outdir}")
    for fp in outdir.iterdir():
        if fp.name == tmpdir.name or fp.suffix == ".log":
            continue
        if fp.is_file():
            fp.unlink()
        elif fp.is_dir():
            shutil.rmtree(fp)
        else:
            logging.warning(f"skip removal of unknown filetype {fp}")
    target = outdir.joinpath(f"pairwise_{metric}_similarities.tsv")
    log(f"writing pairwise {metric} similarities to {target}")
    matrix_to_squareform(dist_mat, target)
    target = outdir.joinpath(f"pairwise_{metric}_adjacency.tsv")
    log(f"writing pairwise {metric} adjacency to {target}")
    matrix_to_adjacency(dist_mat, target, skip_zero=True)
    log(f"moving output files from temporary directory to {outdir}")
    shutil.copytree(tmp_clusters, outdir, dirs_exist_ok=True)
    shutil.rmtree(tmp_clusters)
    if rm_tmp:
        log(f"cleaning up temporary files in {tmpdir}")
        shutil.rmtree(tmpdir)


def main(argv=None):
    if argv is None and len(sys.argv) == 1:
        sys.argv.append("-h")
    args = parse_args(argv)
    if args.genome_dir and not args.infile.is_dir():
        print(f"genome directory '{args.infile}' does not exist")
        sys.exit(1)
    if not args.genome_dir and not args.infile.is_file():
        print(f"input TSV '{args.infile}' does not exist")
        sys.exit(1)
    if not args.outdir.is_dir():
        args.outdir.mkdir(parents=True)
    logging.basicConfig(filename=args.outdir.joinpath("phamclust.log"), filemode="w",
                        level=logging.DEBUG if args.debug else logging.INFO, format=LOG_STR_FMT, datefmt=LOG_TIME_FMT,
                        force=True)
    logging.getLogger().addHandler(logging.StreamHandler(sys.stdout))

    colors = args.heatmap_colors.split(",")
    if not 2 <= len(colors) <= 3:
        logging.error(f"expected either two or three colors, got {len(colors)}")
        sys.exit(1)
    if len(colors) == 2:
        logging.warning("2-color scale ignores `--heatmap-midpoint`")
    unknown = [c for c in colors if c not in CSS_COLORS]
    if unknown:
        for color in unknown:
            logging.error(f"unknown color specified: '{color}'")
        logging.error(f"got {len(unknown)} unrecognized colors")
        logging.error("valid colors:")
        logging.error(" ".join(sorted(CSS_COLORS)))
        sys.exit(1)

    phamclust(infile=args.infile, outdir=args.outdir, is_genome_dir=args.genome_dir, metric=args.metric,
              nr_distance=round(1.0 - args.nr_thresh, 6), nr_linkage=args.nr_linkage,
              clu_distance=round(1.0 - args.clu_thresh, 6), clu_linkage=args.clu_linkage,
              sub_distance=round(1.0 - args.sub_thresh, 6), sub_linkage=args.sub_linkage,
              k_min=max([1, args.k_min]), colors=colors, midpoint=round(args.heatmap_midpoint, 6), cpus=args.threads,
              no_sub=args.no_sub, rm_tmp=args.remove_tmp, debug=args.debug)


if __name__ == "__main__":
    main()
