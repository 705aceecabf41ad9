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


def synth_real(n_genomes, seed=None, m_genome_fraction=0.01):
    """A second workload, shaped like a real phage collection rather than like ``synth(N, P)`` (every threshold in
    the library was tuned on that one family; the reference's own benchmark_data.tsv is a missing blob,
    ``.MISSING_LARGE_BLOBS:1``, protocol ``scripts/benchmark.py:96-113``).  What differs, on purpose:

    * clusters with power-law sizes (Zipf over N/25 clusters: the largest holds >= 15 % of the genomes, most hold a handful),
      genomes of a cluster scattered over the name order;
    * P ~ 6 N phams with a heavy-tailed holder distribution: per cluster ~50 core phams (90 % of its members) and ~50 accessory
      ones (15-25 %), two orphan phams per genome shared with at most two others (most phams have 1-3 holders), and five
      "universal" phams held by 25-60 % of ALL genomes;
    * >= 60 % of the proteins of a cluster byte-identical to the cluster's variant (phage collections are full of identical
      proteins), the rest 3 % substitutions away; the universal phams' cluster variants 25 % substitutions + indels away from
      their ancestor (homologs across clusters that still align);
    * paralog runs: 94 % single genes, 4 % pairs, 1.5 % triples, 0.5 % runs of 4-8;
    * lognormal lengths (median 180, up to 1,500) and, in one cluster out of ten, a tape-measure-like core protein of
      5,000-8,000 residues (beyond the systolic variants' 4,096 columns: strip-mined passes);
    * ``m_genome_fraction`` of the genomes as the reference's 2-column input (every translation "M", scripts/phamclust.py:35-38).
    Deterministic in (N, seed); numpy only.  Returns :class:`PackedGenomes`."""
    N = int(n_genomes)
    rng = np.random.default_rng(977 + N if seed is None else seed)
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    K = max(1, N // 25)
    w = 1.0 / np.arange(1, K + 1)
    size = np.maximum(1, np.floor(N * w / w.sum()).astype(np.int64))
    size[0] += N - int(size.sum()) if size.sum() <= N else 0
    while size.sum() > N:                                        # (tiny N: trim the tail)
        size[np.argmax(size)] -= 1
    cluster_of = np.repeat(np.arange(K), size)[:N]
    cluster_of = cluster_of[rng.permutation(N)]                  # clusters scattered over the (name-sorted) genome order

    def length(n, lo=30, hi=1500):
        return np.clip(np.rint(rng.lognormal(np.log(180.0), 0.6, n)), lo, hi).astype(np.int64)

    # ---- the sequence store: one entry per (cluster, pham) variant / orphan / universal variant
    store, store_off = [], [0]

    def put(seq):
        store.append(seq)
        store_off.append(store_off[-1] + seq.size)
        return len(store) - 1

    def mutate(seq, sub, indel):
        out = seq.copy()
        hit = rng.random(out.size) < sub
        out[hit] = aa[rng.integers(0, 20, int(hit.sum()))]
        if indel > 0 and out.size > 40:
            for _ in range(rng.poisson(indel * out.size)):
                at = int(rng.integers(0, out.size)); n = int(rng.geometric(0.5))
                out = np.delete(out, slice(at, at + n)) if rng.random() < 0.5 else np.insert(out, at, aa[rng.integers(0, 20, n)])
        return out if out.size else seq[:1].copy()

    n_core, n_acc, n_uni = 50, 50, 5
    pham_next = 0
    uni_anc = [aa[rng.integers(0, 20, int(l))] for l in length(n_uni, 150, 900)]
    uni_pham = list(range(n_uni)); pham_next += n_uni
    uni_share = rng.uniform(0.25, 0.6, n_uni)
    genes_g, genes_p, genes_src, genes_ident = [], [], [], []     # per gene: genome, pham, store entry, byte-identical to it?
    for k in range(K):
        members = np.flatnonzero(cluster_of == k)
        lens = length(n_core + n_acc)
        if k % 10 == 3:
            lens[0] = int(rng.integers(5000, 8001))              # a tape-measure-like core protein (clusters 3, 13, ...: mid-sized ones)
        phams = np.arange(pham_next, pham_next + n_core + n_acc); pham_next += n_core + n_acc
        variants = [put(mutate(aa[rng.integers(0, 20, int(l))], 0.0, 0.0)) for l in lens]   # the cluster's own phams: the variant IS the ancestor
        uni_var = [put(mutate(a, 0.25, 0.01)) for a in uni_anc]
        share = np.concatenate([np.full(n_core, 0.9), rng.uniform(0.15, 0.25, n_acc)])
        for g in members:
            held = np.flatnonzero(rng.random(n_core + n_acc) < share)
            for j in held:
                u = rng.random()
                copies = 1 if u < 0.94 else 2 if u < 0.98 else 3 if u < 0.995 else int(rng.integers(4, 9))
                for _ in range(copies):
                    genes_g.append(g); genes_p.append(int(phams[j])); genes_src.append(variants[j]); genes_ident.append(rng.random() < 0.62)
            for j in np.flatnonzero(rng.random(n_uni) < uni_share):
                genes_g.append(g); genes_p.append(uni_pham[j]); genes_src.append(uni_var[j]); genes_ident.append(rng.random() < 0.62)
    # orphans: two new phams per genome, each shared with 0, 1 or 2 other genomes (anywhere in the collection)
    for g in range(N):
        for _ in range(2):
            entry = put(aa[rng.integers(0, 20, int(length(1)[0]))])
            holders = {g} | set(int(x) for x in rng.integers(0, N, int(rng.choice([0, 0, 1, 1, 2]))))
            for h2 in holders:
                genes_g.append(h2); genes_p.append(pham_next); genes_src.append(entry); genes_ident.append(h2 == g)
            pham_next += 1
    genes_g = np.asarray(genes_g, np.int64); genes_p = np.asarray(genes_p, np.int64)
    genes_src = np.asarray(genes_src, np.int64); genes_ident = np.asarray(genes_ident, bool)
    order = np.lexsort((np.arange(genes_g.size), genes_p, genes_g))           # by genome, then pham, paralogs in creation order
    genes_g, genes_p, genes_src, genes_ident = genes_g[order], genes_p[order], genes_src[order], genes_ident[order]
    m_genomes = rng.random(N) < m_genome_fraction                             # the reference's 2-column input: translation "M"
    store_flat = np.concatenate(store) if store else np.zeros(0, np.uint8)
    store_off = np.asarray(store_off, np.int64)
    glen = (store_off[genes_src + 1] - store_off[genes_src])
    glen = np.where(m_genomes[genes_g], 1, glen)
    seq_off = np.concatenate([[0], np.cumsum(glen)]).astype(np.int64)
    residues = np.empty(int(seq_off[-1]), np.uint8)
    G = genes_g.size
    for c0 in range(0, G, 200000):                                           # gather in chunks (the index arrays are 8 B per residue)
        c1 = min(G, c0 + 200000)
        ln = glen[c0:c1]
        pos = np.arange(int(ln.sum()), dtype=np.int64) - np.repeat(seq_off[c0:c1] - seq_off[c0], ln)
        src = np.repeat(store_off[genes_src[c0:c1]], ln) + pos
        part = store_flat[src]
        gene_of = np.repeat(np.arange(c0, c1), ln)
        hit = (rng.random(part.size) < 0.03) & ~genes_ident[gene_of]
        part[hit] = aa[rng.integers(0, 20, int(hit.sum()))]
        part[m_genomes[genes_g[gene_of]]] = ord("M")
        residues[seq_off[c0]:seq_off[c1]] = part
    present, gene_pham = np.unique(genes_p, return_inverse=True)
    gene_pham = gene_pham.astype(np.int32)
    P = int(present.shape[0]); W = max(1, (P + 63) // 64)
    gene_off = np.concatenate([[0], np.cumsum(np.bincount(genes_g, minlength=N))]).astype(np.int64)
    bitmap = np.zeros(N * W, dtype=np.uint64)
    np.bitwise_or.at(bitmap, genes_g * W + (gene_pham >> 6), np.uint64(1) << (gene_pham & 63).astype(np.uint64))
    first = np.ones(G, dtype=bool)
    first[1:] = (gene_pham[1:] != gene_pham[:-1]) | (genes_g[1:] != genes_g[:-1])
    nph = np.bincount(genes_g[first], minlength=N).astype(np.int32)
    ngen = np.diff(gene_off).astype(np.int32)
    tlen = np.bincount(genes_g, weights=glen, minlength=N).astype(np.int64)
    return PackedGenomes(
        names=[f"real_{g:06d}" for g in range(N)], pham_names=[f"pham_{int(p):06d}" for p in present],
        n_genomes=N, n_phams=P, words_per_row=W, bitmap=bitmap, nph=nph, ngen=ngen, tlen=tlen,
        gene_off=gene_off, gene_pham=gene_pham, seq_off=seq_off, residues=residues).validate()


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
