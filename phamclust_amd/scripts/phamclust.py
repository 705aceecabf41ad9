"""The pipeline around the GPU matrix fill.

What it must stay compatible with is the reference's on-disk contract (scripts/phamclust.py:140-473), so that
caches and results interoperate: genomes sorted by name; cache directory ``<outdir>/<md5 of the FASTA text>.tmp``
holding ``01_genomes/<name>.fasta`` and ``02_distmats/<metric>_distance_matrix.tsv`` (lower triangle, reused when
present); three agglomerative passes (pre-group at ``nr``, cluster the group medoids at ``clu``, sub-cluster at
``sub``); ``cluster_<i>/`` (largest first) with ``genomes/*.faa``, ``<metric>_similarity.tsv``,
``subcluster_<j>_similarity.tsv`` and two heatmaps; ``singletons/genomes``; ``pairwise_<metric>_similarities.tsv`` and
``pairwise_<metric>_adjacency.tsv``.  Stage 2 is the only thing that changed: it is one GPU call.
"""

import hashlib
import logging
import pathlib
import shutil
import os
import sys
import time

import numpy as np

from phamclust_amd import distributed
from phamclust_amd import matrix as _matrix
from phamclust_amd import metrics as _metrics
from phamclust_amd.cli import METRICS, parse_args
from phamclust_amd.clustering import hierarchical_clustering
from phamclust_amd.genome import Genome
from phamclust_amd.heatmap import CSS_COLORS, draw_heatmap
from phamclust_amd.matrix import matrix_de_novo, matrix_from_squareform, matrix_to_adjacency, matrix_to_squareform
from phamclust_amd.pack import load_tsv_genomes, packed_behind
from phamclust_amd import startup

LOG_STR_FMT = "phamclust: %(asctime)s.%(msecs)03d: %(levelname)s: %(message)s"
LOG_TIME_FMT = "%H:%M:%S"
FASTA_SUFFIXES = {".fasta", ".faa", ".fa"}
log = logging.getLogger()
TIMELINE = None                    # startup.Timeline of this run (main() makes it; library callers of phamclust() run without)
_PRELOADED = {}                    # infile -> genomes the CLI already loaded for its --gpus estimate


def _mark(name):
    if TIMELINE is not None:
        TIMELINE.mark(name)


# ---- input ---------------------------------------------------------------------------------
def load_genomes_from_tsv(filepath):
    """Rows ``genome, pham[, translation]``; a missing translation is "M"; genomes in first-seen order."""
    found = {}
    with open(filepath) as handle:
        for number, line in enumerate(handle, start=1):
            fields = line.rstrip().split("\t")
            if len(fields) == 2:
                fields.append("M")
            if len(fields) != 3:
                raise ValueError("input file must either 2 or 3 columns")
            name, pham, translation = fields
            genome = found.get(name)
            if genome is None:
                genome = found[name] = Genome(name)
            genome.add(pham, translation)
    return list(found.values())


def load_genomes_from_fasta_dir(filepath):
    """One FASTA file per genome (stem = genome name; headers carry ``pham=``)."""
    found = {}
    for path in filepath.iterdir():
        if path.suffix not in FASTA_SUFFIXES or path.is_dir():
            log.debug(f"skipping {path}")
            continue
        if path.stem in found:
            raise ValueError(f"duplicate genome name detected {path.stem}")
        found[path.stem] = genome = Genome(path.stem)
        genome.load(path)
    return list(found.values())


def load_genomes(filepath):
    """TSV -> name-sorted genomes through the C loader (csrc/pc_pack.c): same genomes, same order, same FASTA text as
    ``sorted(load_genomes_from_tsv(filepath), key=name)`` (scripts/phamclust.py:21-47, 221), without a Python object
    per gene.  Translations with bytes outside ASCII take the Python loader."""
    try:
        return load_tsv_genomes(filepath)
    except ValueError as exc:
        if "non-ASCII" not in str(exc):
            raise
        return sorted(load_genomes_from_tsv(filepath), key=lambda g: g.name)


def _hash_genomes(genomes):
    """md5 of the concatenated FASTA text, genomes in the order given (the cache key)."""
    md5 = hashlib.md5()
    for genome in genomes:
        md5.update(genome.fasta_bytes() if hasattr(genome, "fasta_bytes") else str(genome).encode())
    return md5.hexdigest()


def check_matrix_integrity(matrix):
    """(edge-count error, diagonal-sum error, unfilled slots); all zero for a usable matrix.  The array-backed
    matrix cannot have the wrong number of edges, so the first term is always 0 here."""
    full = matrix.to_ndarray()
    upper = full[np.triu_indices(len(matrix))]
    return 0, float(np.nansum(full.diagonal())) - (0.0 if matrix.is_distance else float(len(matrix))), int(np.isnan(upper).sum())


# ---- the run -----------------------------------------------------------------------------------
class _Run:
    def __init__(self, outdir, metric, colors, midpoint):
        self.outdir, self.metric, self.colors, self.midpoint = outdir, metric, colors, midpoint
        self.genomes, self.by_name, self.cache, self.stage = [], {}, None, None
        self.rank, self.world = 0, 1          # this process's place in the job (one process per GPU)

    @staticmethod
    def _dir(path, fresh=False):
        if fresh and path.is_dir():
            shutil.rmtree(path)
        path.mkdir(exist_ok=True)
        return path

    def banner(self, number, title):
        now = time.perf_counter()
        if getattr(self, "_stage_t0", None) is not None:
            log.info(f"    ({now - self._stage_t0:.1f} s)")          # wall time of the stage that just ended
        self._stage_t0 = now
        log.info(f"--- {number}: {title} ---")

    # 1
    def read(self, infile, is_genome_dir):
        self.banner(1, "genomes")
        t0 = time.perf_counter()
        if is_genome_dir:
            self.genomes = sorted(load_genomes_from_fasta_dir(infile), key=lambda g: g.name)
        else:
            self.genomes = _PRELOADED.pop(str(infile), None) or load_genomes(infile)
        self.by_name = {g.name: g for g in self.genomes}
        log.info(f"loaded {len(self.genomes)} genomes in {time.perf_counter() - t0:.3f} s")
        _mark("load_genomes")
        if self.rank != 0:                    # the other ranks only need the genomes: rank 0 owns the output tree
            return
        digest = _hash_genomes(self.genomes)
        self.cache = self._dir(self.outdir / f"{digest}.tmp")
        log.info(f"{len(self.genomes)} genomes, md5 {digest}, cache {self.cache.name}")
        stash = self._dir(self.cache / "01_genomes")
        for genome in self.genomes:
            target = stash / f"{genome.name}.fasta"
            if not target.is_file():
                genome.save(target)

    # 2
    def distances(self, cpus):
        self.banner(2, f"{self.metric} distance matrix")
        cached, have_cache, failure = None, False, None
        if self.rank == 0:
            try:
                cached = self._dir(self.cache / "02_distmats") / f"{self.metric}_distance_matrix.tsv"
                have_cache = cached.is_file()
            except Exception as exc:          # noqa: BLE001 -- the other ranks must hear of it before they enter the gather
                if self.world == 1:
                    raise
                failure = exc
        if self.world > 1:                    # rank 0 owns the cache; the others learn whether there is work for them
            have_cache = distributed.broadcast_flag(have_cache, src=0, error=failure)
        t0 = time.perf_counter()
        if have_cache:
            if self.rank != 0:
                return None
            matrix = matrix_from_squareform(cached)
            log.info(f"read cached matrix in {time.perf_counter() - t0:.3f} s")
        else:
            matrix = matrix_de_novo(self.genomes, METRICS[self.metric], cpus)
            if matrix is None:                # not rank 0: this rank's shard went into the gather, nothing else to do
                return None
            wall = time.perf_counter() - t0
            st = _matrix.LAST_FILL
            pairs = st.get("genome_pairs", 0)
            line = (f"filled {len(matrix)} x {len(matrix)} matrix on {st.get('n_gpus', 1)} GPU(s) in {wall:.3f} s "
                    f"(pack {st.get('pack_s', 0.0):.3f}, upload {st.get('upload_s', 0.0):.3f}, fill+gather+D2H {st.get('fill_s', 0.0):.3f}): "
                    f"{pairs / max(st.get('fill_s', 0.0), 1e-9):.3e} genome-pairs/s")
            if st.get("n_cells"):
                line += (f"; rank 0: {st['n_alignments']} alignments, {st['n_cells']:.3e} DP cells, "
                         f"{st['n_distinct_cells'] / max(st['ms_align'], 1e-6) / 1e6:.0f} GCUPS in the alignment kernels")
            log.info(line)
            peers = st.get("peer_access")                    # one process, N GPUs: how every shard reached the root -- never silent
            if peers:
                ways = ", ".join(f"{d['device']}: {d['access']}" for d in peers["devices"])
                log.info(f"exchange: device -> root copies {ways}; {st.get('ms_exchange', 0.0):.2f} ms (slowest copy)"
                         + (f"; STAGED THROUGH HOST for {peers['staged_through_host']} device(s): {peers['note']} "
                            f"(PHAMCLUST_MULTI=launcher gathers over RCCL instead)" if peers["staged_through_host"] else "")
                         + ("" if any(d["code"] == 1 for d in peers["devices"]) or peers["staged_through_host"] else
                            " [all contexts on the root's GPU: the cross-device branch did not run]"))
            if self.metric in _metrics.PARITY_NOTE:         # say it wherever these numbers leave the program
                log.info(f"parity: {_metrics.parity_note(self.metric)}")
            if TIMELINE is not None:                         # the matrix stage, split (matrix_de_novo's own clocks)
                TIMELINE.stamps += [("pack", st.get("pack_s", 0.0)), ("process_group_and_context", max(0.0, wall - st.get("pack_s", 0.0) - st.get("upload_s", 0.0) - st.get("fill_s", 0.0))),
                                    ("upload", st.get("upload_s", 0.0)), ("fill_exchange_d2h", st.get("fill_s", 0.0))]
                TIMELINE._last = time.time()
            matrix_to_squareform(matrix, cached, lower_triangle=True)
        if not matrix.is_distance:
            matrix.invert()
        edges, diagonal, unfilled = check_matrix_integrity(matrix)
        if edges or diagonal or unfilled:
            log.error(f"matrix failed integrity checks: edge count off by {edges}, diagonal off by {diagonal}, {unfilled} unfilled")
            print("matrix validation failed - check log for details")
            sys.exit(1)
        return matrix

    # 3
    def clusters(self, matrix, nr, clu):
        self.banner(3, "clusters")
        groups = hierarchical_clustering(matrix, eps=nr[0], linkage=nr[1])          # near-identical genomes
        by_medoid = {group.medoid[0]: group for group in groups}
        medoids = matrix.extract_submatrix(list(by_medoid))
        merged = []
        for cluster in hierarchical_clustering(medoids, eps=clu[0], linkage=clu[1]):
            members = [node for medoid in cluster.nodes for node in by_medoid[medoid].nodes]
            merged.append(matrix.extract_submatrix(members))
        multi = sorted((m for m in merged if len(m) > 1), reverse=True)
        single = [m for m in merged if len(m) == 1]
        log.info(f"{len(groups)} pre-groups -> {len(multi)} clusters + {len(single)} singletons")
        return multi, single

    # 4
    def write_cluster(self, number, cluster, sub, k_min, no_sub):
        root = self._dir(self.stage / f"cluster_{number}", fresh=True)
        faa = self._dir(root / "genomes")
        members = set(cluster.nodes)
        for genome in self.genomes:
            if genome.name in members:
                genome.save(faa / f"{genome.name}.faa")
        if no_sub or len(cluster) < k_min:
            cluster.reorder()
        else:
            order = []
            parts = sorted(hierarchical_clustering(cluster, eps=sub[0], linkage=sub[1]), reverse=True)
            for j, part in enumerate(parts, start=1):
                if len(part) > 2:
                    part.reorder()
                order += part.nodes
                matrix_to_squareform(part.invert(), root / f"subcluster_{j}_similarity.tsv")
            cluster.reorder(order)
        matrix_to_squareform(cluster.invert(), root / f"{self.metric}_similarity.tsv")
        for suffix in ("svg", "html"):
            draw_heatmap(cluster, colors=self.colors, midpoint=self.midpoint, filename=root / f"{self.metric}_heatmap.{suffix}")

    # 5 + 6
    def finish(self, matrix, multi, single, clu_distance, rm_tmp):
        self.banner(5, "dataset outputs")
        matrix.reorder([n for m in multi for n in m.nodes] + [n for m in single for n in m.nodes])
        matrix.invert()
        if len(matrix) > 1500:
            log.info("matrix too large for a dataset heatmap")
        else:
            for suffix in ("html", "svg"):
                draw_heatmap(matrix, colors=self.colors, midpoint=1.0 - clu_distance,
                             filename=self.stage / f"{self.metric}_heatmap.{suffix}")
        for old in self.outdir.iterdir():                      # clear earlier results, keep cache and log
            if old.name == self.cache.name or old.suffix == ".log":
                continue
            shutil.rmtree(old) if old.is_dir() else old.unlink()
        matrix_to_squareform(matrix, self.outdir / f"pairwise_{self.metric}_similarities.tsv")
        matrix_to_adjacency(matrix, self.outdir / f"pairwise_{self.metric}_adjacency.tsv", skip_zero=True)
        shutil.copytree(self.stage, self.outdir, dirs_exist_ok=True)
        shutil.rmtree(self.stage)
        if rm_tmp:
            shutil.rmtree(self.cache)


def phamclust(infile, outdir, is_genome_dir, metric, nr_distance, nr_linkage, clu_distance, clu_linkage, sub_distance,
              sub_linkage, k_min, no_sub, colors, midpoint, cpus, rm_tmp, debug):
    """Same signature as the reference's ``phamclust()`` (distances, not similarities, for the thresholds)."""
    if nr_distance >= clu_distance:          # pre-grouping must be tighter than clustering, else switch it off
        nr_distance = 0.0
    settings = dict(infile=infile, outdir=outdir, metric=metric, nr=(nr_distance, nr_linkage), clu=(clu_distance, clu_linkage),
                    sub=(sub_distance, sub_linkage), k_min=k_min, subcluster=not no_sub, colors=",".join(colors),
                    midpoint=midpoint, cpus=cpus, remove_tmp=rm_tmp, debug=debug)
    log.info("--- 0: settings ---")
    for key, value in settings.items():
        log.info(f"{key:<11}{value}")
    run = _Run(outdir, metric, colors, midpoint)
    run.rank, run.world = distributed.ensure_process_group()      # (0, 1) unless started by torch.distributed.run
    _mark("torch_import_and_process_group")
    run.read(infile, is_genome_dir)
    matrix = run.distances(cpus)
    if matrix is None:                        # ranks other than 0 are done once their shard is gathered
        return
    multi, single = run.clusters(matrix, (nr_distance, nr_linkage), (clu_distance, clu_linkage))
    run.banner(4, "sub-clusters")
    run.stage = run._dir(run.cache / "03_clusters")
    for number, cluster in enumerate(multi, start=1):
        log.info(f"cluster {number}: {len(cluster)} genomes")
        run.write_cluster(number, cluster, (sub_distance, sub_linkage), k_min, no_sub)
    if single:
        lone = run._dir(run._dir(run.stage / "singletons", fresh=True) / "genomes")
        for one in single:
            name = one.nodes[0]
            run.by_name[name].save(lone / f"{name}.faa")
    run.finish(matrix, multi, single, clu_distance, rm_tmp)


def _colors(text):
    colors = text.split(",")
    if not 2 <= len(colors) <= 3:
        log.error(f"expected either two or three colors, got {len(colors)}")
        sys.exit(1)
    bad = [c for c in colors if c not in CSS_COLORS]
    if bad:
        log.error(f"unknown colors {bad}; valid colors: {' '.join(sorted(CSS_COLORS))}")
        sys.exit(1)
    if len(colors) == 2:
        log.warning("2-color scale ignores `--heatmap-midpoint`")
    return colors


def main(argv=None):
    if argv is None and len(sys.argv) == 1:
        sys.argv.append("-h")
    args = parse_args(argv)
    if not (args.infile.is_dir() if args.genome_dir else args.infile.is_file()):
        kind = "genome directory" if args.genome_dir else "input TSV"
        print(f"{kind} '{args.infile}' does not exist")
        sys.exit(1)
    global TIMELINE
    TIMELINE = startup.Timeline()
    TIMELINE.mark("imports")
    # What this run tells the library layers through the environment (PHAMCLUST_GPUS / _NO_TORCH / _DEVICE) is THIS RUN's: the old
    # values come back when main() returns, so that a caller of main() -- a test, a notebook -- does not find its later
    # matrix_de_novo calls redirected to N GPUs, or its later `import torch` next to a HIP runtime bound without it.
    saved_env = {key: os.environ.get(key) for key in ("PHAMCLUST_GPUS", "PHAMCLUST_GPU_IDS", "PHAMCLUST_NO_TORCH", "PHAMCLUST_DEVICE")}
    try:
        _run(args, argv)
    finally:
        for key, value in saved_env.items():
            if value is None:
                os.environ.pop(key, None)
            else:
                os.environ[key] = value


def _run(args, argv):
    rank, _, world = distributed.env_world()
    gpus_note = None
    if args.gpus > 1 and world == 1:
        # `--gpus N`.  Two routes (startup.multi_gpu_route): by default THIS process drives the N GPUs (pc_multi_*: nothing to
        # launch); PHAMCLUST_MULTI=launcher re-runs the command as N ranks under torch.distributed.run, which costs seconds
        # before the first pair (an interpreter, a torch import and a process group per rank).  Either way the reference's rule
        # applies -- never more workers than work (matrix.py:460-462): load the genomes (host code only, no GPU call), estimate
        # the fill, and spread it only when N GPUs can win their start-up back.
        route = startup.multi_gpu_route()
        n_gpus = args.gpus
        if not args.genome_dir:
            genomes = load_genomes(args.infile)
            TIMELINE.mark("load_genomes_for_estimate")
            packed = packed_behind(genomes)
            if packed is not None:
                n_gpus, gpus_note = startup.choose_gpus(args.gpus, packed, args.metric, route)
                _PRELOADED[str(args.infile)] = genomes
        if n_gpus > 1 and route == "launcher":
            # as N ranks, one per GPU -- as a child process, before this process has made a single GPU call
            passed = list(sys.argv[1:] if argv is None else argv)
            sys.exit(distributed.launch_ranks(n_gpus, "phamclust_amd", [str(x) for x in passed],
                                              env={"PHAMCLUST_T0": repr(TIMELINE.t_launch), "PHAMCLUST_FORCE_GPUS": "1"}))
        if n_gpus > 1:
            os.environ["PHAMCLUST_GPUS"] = str(n_gpus)             # matrix_de_novo: devices 0..N-1 from this process (PHAMCLUST_GPU_IDS names others)
            if args.device is not None:
                gpus_note = (gpus_note or "") + (f"; --device {args.device} is IGNORED: {n_gpus} GPUs from one process are devices 0..{n_gpus - 1} "
                                                 f"(name others with PHAMCLUST_GPU_IDS)")
        else:
            os.environ.pop("PHAMCLUST_GPUS", None); os.environ.pop("PHAMCLUST_GPU_IDS", None)
    if world == 1:
        os.environ.setdefault("PHAMCLUST_NO_TORCH", "1")           # one rank: nothing needs torch (hip.load() then skips its import)
    if args.device is not None and world == 1:
        os.environ["PHAMCLUST_DEVICE"] = str(args.device)           # matrix.default_device reads it when the context is made
    args.outdir.mkdir(parents=True, exist_ok=True)
    if rank == 0:
        logging.basicConfig(filename=args.outdir / "phamclust.log", filemode="w", format=LOG_STR_FMT, datefmt=LOG_TIME_FMT,
                            level=logging.DEBUG if args.debug else logging.INFO, force=True)
        log.addHandler(logging.StreamHandler(sys.stdout))
    else:                                     # rank 0 owns the log file and the output tree
        logging.basicConfig(stream=sys.stderr, format=f"phamclust[rank {rank}]: %(levelname)s: %(message)s", level=logging.WARNING, force=True)
    if gpus_note and rank == 0:
        log.info(f"--gpus {args.gpus}: {gpus_note}")
    as_distance = lambda similarity: round(1.0 - similarity, 6)          # noqa: E731
    try:
        phamclust(infile=args.infile, outdir=args.outdir, is_genome_dir=args.genome_dir, metric=args.metric,
                  nr_distance=as_distance(args.nr_thresh), nr_linkage=args.nr_linkage,
                  clu_distance=as_distance(args.clu_thresh), clu_linkage=args.clu_linkage,
                  sub_distance=as_distance(args.sub_thresh), sub_linkage=args.sub_linkage, k_min=max(1, args.k_min),
                  no_sub=args.no_sub, colors=_colors(args.heatmap_colors), midpoint=round(args.heatmap_midpoint, 6),
                  cpus=args.threads, rm_tmp=args.remove_tmp, debug=args.debug)
    finally:
        if rank == 0:
            TIMELINE.mark("clustering_and_outputs")
            log.info(TIMELINE.line())
        if world > 1:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()


if __name__ == "__main__":
    main()
