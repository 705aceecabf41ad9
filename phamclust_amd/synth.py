"""Synthetic pham/translation data: ``synth(N, P, seed)`` of SURVEY.md section 8(d).

The generator itself is plain C (``csrc/pc_synth.c``, built by ``phamclust_amd.build``);
this module wraps it and returns the same :class:`~phamclust_amd.pack.PackedGenomes` that
``pack_genomes`` would build from the equivalent ``Genome`` list / 3-column TSV.
"""

import ctypes
import os

import numpy as np

from phamclust_amd.pack import PackedGenomes, unpack_genomes



def _lib_path():
    from phamclust_amd.build import native_path
    return native_path("libpc_synth.so")



class _Data(ctypes.Structure):
    _fields_ = [("n_genomes", ctypes.c_int32), ("n_phams", ctypes.c_int32),
                ("n_genes", ctypes.c_int64), ("n_residues", ctypes.c_int64),
                ("gene_off", ctypes.POINTER(ctypes.c_int64)), ("gene_pham", ctypes.POINTER(ctypes.c_int32)),
                ("seq_off", ctypes.POINTER(ctypes.c_int64)), ("residues", ctypes.POINTER(ctypes.c_uint8))]


def default_seed(n_genomes):
    return 20241218 + n_genomes


def synth_packed(n_genomes, n_phams=5000, seed=None):
    """Generate synth(N, P, seed) directly in packed form (fast: ~1 s for N=5,000)."""
    path = _lib_path()
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing - run `python -m phamclust_amd.build` first")
    lib = ctypes.CDLL(path)
    lib.pcs_generate.restype = ctypes.POINTER(_Data)
    lib.pcs_generate.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_uint64]
    lib.pcs_free.argtypes = [ctypes.POINTER(_Data)]
    seed = default_seed(n_genomes) if seed is None else seed
    handle = lib.pcs_generate(n_genomes, n_phams, seed)
    if not handle:
        raise RuntimeError("pcs_generate failed")
    try:
        d = handle.contents
        N, G, R = d.n_genomes, d.n_genes, d.n_residues
        gene_off = np.ctypeslib.as_array(d.gene_off, shape=(N + 1,)).copy()
        raw_pham = np.ctypeslib.as_array(d.gene_pham, shape=(G,)).copy()
        seq_off = np.ctypeslib.as_array(d.seq_off, shape=(G + 1,)).copy()
        residues = np.ctypeslib.as_array(d.residues, shape=(R,)).copy()
    finally:
        lib.pcs_free(handle)

    present, gene_pham = np.unique(raw_pham, return_inverse=True)     # ids = ranks of present pham names
    gene_pham = gene_pham.astype(np.int32)
    P = int(present.shape[0])
    W = max(1, (P + 63) // 64)
    gene_genome = np.repeat(np.arange(N, dtype=np.int64), np.diff(gene_off))
    bitmap = np.zeros(N * W, dtype=np.uint64)
    np.bitwise_or.at(bitmap, gene_genome * W + (gene_pham >> 6),
                     np.uint64(1) << (gene_pham & 63).astype(np.uint64))
    first = np.ones(G, dtype=bool)
    first[1:] = (gene_pham[1:] != gene_pham[:-1]) | (gene_genome[1:] != gene_genome[:-1])
    nph = np.bincount(gene_genome[first], minlength=N).astype(np.int32)
    ngen = np.diff(gene_off).astype(np.int32)
    tlen = np.add.reduceat(np.diff(seq_off), gene_off[:-1]).astype(np.int64) if G else np.zeros(N, np.int64)
    return PackedGenomes(
        names=[f"synth_{g:06d}" for g in range(N)], pham_names=[f"pham_{int(p):06d}" for p in present],
        n_genomes=N, n_phams=P, words_per_row=W, bitmap=bitmap, nph=nph, ngen=ngen, tlen=tlen,
        gene_off=gene_off.astype(np.int64), gene_pham=gene_pham, seq_off=seq_off.astype(np.int64),
        residues=residues).validate()


def synth_genomes(n_genomes, n_phams=5000, seed=None):
    """The same data as ``Genome`` objects (small N only: builds Python strings)."""
    return unpack_genomes(synth_packed(n_genomes, n_phams, seed))


def write_tsv(genomes, filepath):
    """3-column TSV ``genome<TAB>pham<TAB>translation`` (reference scripts/phamclust.py:21-47)."""
    with open(filepath, "w") as handle:
        for genome in genomes:
            for pham, translations in genome:
                for translation in translations:
                    handle.write(f"{genome.name}\t{pham}\t{translation}\n")
    return filepath


def write_tsv_packed(packed, filepath):
    """The same TSV straight from packed arrays (no Genome objects): genes in packed order, i.e. per genome sorted
    by pham id -- one of the many line orders that load back to the same genomes."""
    res = packed.residues.tobytes()
    seq_off = packed.seq_off.tolist()
    gene_pham = packed.gene_pham.tolist()
    phams = [name.encode() for name in packed.pham_names]
    with open(filepath, "wb") as handle:
        for g, name in enumerate(packed.names):
            prefix = name.encode() + b"\t"
            k0, k1 = int(packed.gene_off[g]), int(packed.gene_off[g + 1])
            handle.write(b"".join(prefix + phams[gene_pham[k]] + b"\t" + res[seq_off[k]:seq_off[k + 1]] + b"\n" for k in range(k0, k1)))
    return filepath
