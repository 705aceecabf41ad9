"""Command line of the pipeline: same flags, defaults and ``METRICS`` selector as the reference
(cli.py:10-117), declared here as a table so that the defaults are inspectable (``DEFAULTS``)."""

import argparse
import pathlib
from multiprocessing import cpu_count

from phamclust_amd import metrics as _m
from phamclust_amd.metrics import *          # noqa: F401,F403  re-export the six metric callables

# name -> callable, in the reference's order; `-m` picks one of these keys
METRICS = dict(gcs=_m.gene_content_similarity, jc=_m.jaccard_coefficient, pocp=_m.percentage_of_conserved_proteins,
               af=_m.alignment_fraction, aai=_m.average_aminoacid_identity, peq=_m.proteomic_equivalence_quotient)
LINKAGES = {"single", "average", "complete"}
CPUS = cpu_count()
GPUS = 1                                     # the fill's counterpart of -t: how many GPUs share the pair list
COLORS = "red,yellow,green"
METRIC, K_MIN = "peq", 6
NR_THRESH, NR_LINKAGE = 0.75, "complete"     # 1st pass: glue near-identical genomes together
CLU_THRESH, CLU_LINKAGE = 0.25, "average"    # 2nd pass: clusters
SUB_THRESH, SUB_LINKAGE = 0.6, "single"      # 3rd pass: sub-clusters

_PAPERS = (("gcs", "gene content similarity", "10.1038/nmicrobiol.2017.112"),
           ("jc", "jaccard coefficient", "10.1111/j.1469-8137.1912.tb05611.x"),
           ("pocp", "percentage of conserved proteins", "10.1128/JB.01688-14"),
           ("af", "alignment fraction", "10.1093/nar/gkv657"),
           ("aai", "average aminoacid identity", "10.1073/pnas.0409727102"),
           ("peq", "proteomic equivalence quotient", "10.1128/msystems.00443-23"))
EPILOG = "\nAvailable metrics:\n\n" + "\n".join(
    f"({i}) {key:<6}{title:<36}https://doi.org/{doi}" for i, (key, title, doi) in enumerate(_PAPERS, start=1)) + "\n"

# (group, short, long, kwargs); "%(default)s" is filled in by argparse
_OPTIONS = (
    (None, "-g", "--genome-dir", dict(action="store_true", help="`infile` is a directory with one FASTA per genome, not a TSV")),
    ("clustering arguments:", "-k", "--k-min", dict(type=int, default=K_MIN, help="smallest cluster that gets sub-clustered")),
    ("clustering arguments:", "-s", "--sub-thresh", dict(type=float, default=SUB_THRESH, help="similarity threshold of the sub-clustering pass")),
    ("clustering arguments:", "-sl", "--sub-linkage", dict(type=str, choices=LINKAGES, default=SUB_LINKAGE, help="linkage of the sub-clustering pass")),
    ("clustering arguments:", "-c", "--clu-thresh", dict(type=float, default=CLU_THRESH, help="similarity threshold of the clustering pass")),
    ("clustering arguments:", "-cl", "--clu-linkage", dict(type=str, choices=LINKAGES, default=CLU_LINKAGE, help="linkage of the clustering pass")),
    ("clustering arguments:", "-nr", "--nr-thresh", dict(type=float, default=NR_THRESH, help="similarity above which genomes are pre-grouped and can never be split")),
    ("clustering arguments:", "-nl", "--nr-linkage", dict(type=str, choices=LINKAGES, default=NR_LINKAGE, help="linkage of that pre-grouping pass")),
    ("clustering arguments:", "-m", "--metric", dict(type=str, choices=METRICS, default=METRIC, help="pairwise relatedness index (see below)")),
    ("heatmap arguments:", "-hc", "--heatmap-colors", dict(type=str, default=COLORS, help="2 or 3 comma-separated CSS colour names")),
    ("heatmap arguments:", "-hm", "--heatmap-midpoint", dict(type=float, default=0.5, help="where the middle colour sits on a 3-colour scale")),
    (None, "-d", "--debug", dict(action="store_true", help="log at DEBUG level")),
    (None, "-n", "--no-sub", dict(action="store_true", help="skip the sub-clustering pass")),
    (None, "-r", "--remove-tmp", dict(action="store_true", help="delete the cache directory at the end (re-runs then recompute the matrix)")),
    (None, "-t", "--threads", dict(type=int, default=CPUS, help="accepted for compatibility; the six metrics run on the GPU (see --gpus)")),
    (None, "-D", "--device", dict(type=int, default=None, help="HIP device ordinal of a single-GPU run (default: PHAMCLUST_DEVICE, else 0); "
                                                               "with --gpus N the ranks take devices 0..N-1")),
    (None, "-G", "--gpus", dict(type=int, default=GPUS, help="GPUs of this node to spread the matrix fill over (one process per GPU "
                                                              "under torch.distributed.run, static pair shard, one RCCL gather)")),
)
DEFAULTS = {long.lstrip("-").replace("-", "_"): kw.get("default", False) for _, _, long, kw in _OPTIONS}


def build_parser():
    parser = argparse.ArgumentParser(prog="phamclust", epilog=EPILOG, formatter_class=argparse.RawTextHelpFormatter,
                                     description="Cluster phage genomes using gene content similarity-based metrics.")
    parser.add_argument("infile", type=pathlib.Path, help="TSV: genome <tab> pham <tab> translation (translation optional)")
    parser.add_argument("outdir", type=pathlib.Path, help="where results (and the cache directory) are written")
    groups = {}
    for group, short, long, kwargs in _OPTIONS:
        target = parser if group is None else groups.setdefault(group, parser.add_argument_group(group))
        kwargs = dict(kwargs)
        if "default" in kwargs:
            kwargs["help"] += " [default: %(default)s]"
            kwargs["metavar"] = ""
        target.add_argument(short, long, **kwargs)
    return parser


def parse_args(argv=None):
    return build_parser().parse_args(argv)
