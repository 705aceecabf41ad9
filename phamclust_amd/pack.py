"""Pack a name-sorted ``list[Genome]`` into the flat arrays the C-ABI consumes.

This is the hand-over format of the drop-in boundary (``pc_packed`` in
``include/phamclust_hip.h``).  List order is pair orientation: for a pair (s, t) with
s < t, genome s is the reference's ``source`` and t its ``target``
(reference scripts/phamclust.py:221 sorts by name before matrix_de_novo, and
matrix.py:479-486 pairs ``genomes[i]`` with ``genomes[i+1:]``).

Layout
  bitmap     u64[N*W]   bit p of row g set iff genome g holds pham id p (W = ceil(P/64))
  nph        i32[N]     distinct phams          == len(g.phams)
  ngen       i32[N]     genes, paralogs incl.   == len(g)
  tlen       i64[N]     summed translation length
  gene_off   i64[N+1]   CSR: genes of genome g are [gene_off[g], gene_off[g+1])
  gene_pham  i32[G]     pham id per gene; within a genome sorted ascending, stable in
                        the genome's own list order for paralogs
  seq_off    i64[G+1]   residues of gene k are [seq_off[k], seq_off[k+1])
  residues   u8[R]      raw bytes of the translations (one byte per character)
Pham ids are ranks of the sorted unique pham names.
"""

from dataclasses import dataclass, field

import numpy as np

from phamclust_amd.genome import Genome


@dataclass
class PackedGenomes:
    names: list
    pham_names: list
    n_genomes: int
    n_phams: int
    words_per_row: int
    bitmap: np.ndarray
    nph: np.ndarray
    ngen: np.ndarray
    tlen: np.ndarray
    gene_off: np.ndarray
    gene_pham: np.ndarray
    seq_off: np.ndarray
    residues: np.ndarray
    gene_order: np.ndarray = None              # C loader only: input-line rank of each gene (restores insertion order)
    _keepalive: list = field(default_factory=list, repr=False)

    @property
    def n_genes(self):
        return int(self.gene_pham.shape[0])

    @property
    def n_pairs(self):
        return self.n_genomes * (self.n_genomes - 1) // 2

    def validate(self):
        """Shape/consistency checks done on the host before any kernel sees the data."""
        N, W, G = self.n_genomes, self.words_per_row, self.n_genes
        assert self.bitmap.dtype == np.uint64 and self.bitmap.shape == (N * W,)
        assert self.nph.dtype == np.int32 and self.nph.shape == (N,)
        assert self.ngen.dtype == np.int32 and self.ngen.shape == (N,)
        assert self.tlen.dtype == np.int64 and self.tlen.shape == (N,)
        assert self.gene_off.dtype == np.int64 and self.gene_off.shape == (N + 1,)
        assert self.gene_pham.dtype == np.int32
        assert self.seq_off.dtype == np.int64 and self.seq_off.shape == (G + 1,)
        assert self.residues.dtype == np.uint8
        assert int(self.gene_off[-1]) == G and int(self.seq_off[-1]) == self.residues.shape[0]
        assert W == max(1, (self.n_phams + 63) // 64)
        for arr in (self.bitmap, self.nph, self.ngen, self.tlen, self.gene_off, self.gene_pham,
                    self.seq_off, self.residues):
            assert arr.flags["C_CONTIGUOUS"]
        return self


def _encode(translation):
    try:
        return translation.encode("latin-1")
    except UnicodeEncodeError as exc:          # one byte per character is part of the contract
        raise ValueError(f"translation contains a character outside latin-1: {exc}") from None


def pack_genomes(genomes):
    """Pack genomes (in the given order) into a :class:`PackedGenomes`."""
    if len(genomes) == 0:
        raise ValueError("need at least 1 genome to pack")
    vocab = sorted({pham for g in genomes for pham in g.phams})
    pham_id = {name: k for k, name in enumerate(vocab)}
    N, P = len(genomes), len(vocab)
    W = max(1, (P + 63) // 64)

    bitmap = np.zeros((N, W), dtype=np.uint64)
    nph = np.zeros(N, dtype=np.int32)
    ngen = np.zeros(N, dtype=np.int32)
    tlen = np.zeros(N, dtype=np.int64)
    gene_off = np.zeros(N + 1, dtype=np.int64)
    gene_pham, seq_len, chunks = [], [], []

    for g_idx, genome in enumerate(genomes):
        ids = sorted((pham_id[p], p) for p in genome.phams)
        nph[g_idx] = len(ids)
        count, total = 0, 0
        cols = np.fromiter((i for i, _ in ids), dtype=np.int64, count=len(ids))
        if len(ids):
            np.bitwise_or.at(bitmap[g_idx], cols >> 6, np.uint64(1) << (cols & 63).astype(np.uint64))
        for pid, pham in ids:
            for translation in genome.phams[pham]:
                raw = _encode(translation)
                gene_pham.append(pid)
                seq_len.append(len(raw))
                chunks.append(raw)
                count += 1
                total += len(raw)
        ngen[g_idx] = count
        tlen[g_idx] = total
        gene_off[g_idx + 1] = gene_off[g_idx] + count

    G = len(gene_pham)
    seq_off = np.zeros(G + 1, dtype=np.int64)
    if G:
        np.cumsum(np.asarray(seq_len, dtype=np.int64), out=seq_off[1:])
    residues = np.frombuffer(b"".join(chunks), dtype=np.uint8).copy() if G else np.zeros(0, np.uint8)
    return PackedGenomes(
        names=[g.name for g in genomes], pham_names=vocab, n_genomes=N, n_phams=P, words_per_row=W,
        bitmap=np.ascontiguousarray(bitmap.reshape(-1)), nph=nph, ngen=ngen, tlen=tlen, gene_off=gene_off,
        gene_pham=np.asarray(gene_pham, dtype=np.int32), seq_off=seq_off, residues=residues).validate()


def unpack_genomes(packed):
    """Inverse of :func:`pack_genomes` (used by the synthetic generator and tests)."""
    genomes = []
    res = packed.residues.tobytes()
    for g_idx, name in enumerate(packed.names):
        genome = Genome(name)
        for k in range(int(packed.gene_off[g_idx]), int(packed.gene_off[g_idx + 1])):
            seq = res[int(packed.seq_off[k]):int(packed.seq_off[k + 1])].decode("latin-1")
            genome.add(packed.pham_names[int(packed.gene_pham[k])], seq)
        genomes.append(genome)
    return genomes


# ---------------------------------------------------------------------------------------
# TSV -> PackedGenomes in C (csrc/pc_pack.c): no Python object per gene
# ---------------------------------------------------------------------------------------
_PACK_LIB = None


def _pack_lib():
    import ctypes
    import os
    global _PACK_LIB
    if _PACK_LIB is not None:
        return _PACK_LIB
    from phamclust_amd.build import native_path
    lib_path = native_path("libpc_pack.so")
    if not os.path.exists(lib_path):
        raise RuntimeError(f"{lib_path} is missing - run `python -m phamclust_amd.build` first")

    class _Data(ctypes.Structure):
        _fields_ = [("n_genomes", ctypes.c_int32), ("n_phams", ctypes.c_int32), ("words_per_row", ctypes.c_int32),
                    ("status", ctypes.c_int32), ("n_genes", ctypes.c_int64), ("n_residues", ctypes.c_int64),
                    ("names_bytes", ctypes.c_int64), ("pham_names_bytes", ctypes.c_int64),
                    ("bitmap", ctypes.POINTER(ctypes.c_uint64)), ("nph", ctypes.POINTER(ctypes.c_int32)),
                    ("ngen", ctypes.POINTER(ctypes.c_int32)), ("tlen", ctypes.POINTER(ctypes.c_int64)),
                    ("gene_off", ctypes.POINTER(ctypes.c_int64)), ("gene_pham", ctypes.POINTER(ctypes.c_int32)),
                    ("seq_off", ctypes.POINTER(ctypes.c_int64)), ("residues", ctypes.POINTER(ctypes.c_uint8)),
                    ("names", ctypes.POINTER(ctypes.c_char)), ("name_off", ctypes.POINTER(ctypes.c_int64)),
                    ("pham_names", ctypes.POINTER(ctypes.c_char)), ("pham_name_off", ctypes.POINTER(ctypes.c_int64)),
                    ("file", ctypes.c_void_p), ("error", ctypes.c_char * 256),
                    ("gene_order", ctypes.POINTER(ctypes.c_int64))]

    lib = ctypes.CDLL(lib_path)
    lib.pcp_load_tsv.restype = ctypes.POINTER(_Data)
    lib.pcp_load_tsv.argtypes = [ctypes.c_char_p]
    lib.pcp_free.argtypes = [ctypes.POINTER(_Data)]
    lib.pcp_genome_fasta.restype = ctypes.c_int64
    lib.pcp_genome_fasta.argtypes = [ctypes.POINTER(_Data), ctypes.c_int32, ctypes.c_char_p, ctypes.c_int64]
    _PACK_LIB = lib
    return lib


class _LoaderHandle:
    """Owns the C loader's result for as long as lazy genomes may ask it for their FASTA text."""

    def __init__(self, lib, handle):
        self.lib, self.handle = lib, handle

    def fasta(self, index):
        import ctypes
        need = self.lib.pcp_genome_fasta(self.handle, index, None, 0)
        if need < 0:
            raise RuntimeError("pcp_genome_fasta failed")
        buf = ctypes.create_string_buffer(int(need) + 1)
        size = self.lib.pcp_genome_fasta(self.handle, index, buf, need + 1)
        return buf.raw[:size]

    def __del__(self):
        try:
            if self.handle:
                self.lib.pcp_free(self.handle)
                self.handle = None
        except Exception:
            pass


def _load_tsv(filepath, keep_handle):
    import ctypes
    import os
    lib = _pack_lib()
    handle = lib.pcp_load_tsv(os.fsencode(str(filepath)))
    if not handle:
        raise MemoryError("pcp_load_tsv: out of memory")
    owner = _LoaderHandle(lib, handle)
    d = handle.contents
    if d.status != 0:
        raise ValueError(d.error.decode("utf-8", "replace"))
    N, P, W, G, R = d.n_genomes, d.n_phams, d.words_per_row, d.n_genes, d.n_residues

    def arr(ptr, n):
        return np.ctypeslib.as_array(ptr, shape=(max(n, 1),))[:n].copy()

    def strings(buf, offs, n, nbytes):
        raw = ctypes.string_at(buf, nbytes)
        off = np.ctypeslib.as_array(offs, shape=(n + 1,))
        return [raw[int(off[i]):int(off[i + 1])].decode("utf-8") for i in range(n)]

    packed = PackedGenomes(
        names=strings(d.names, d.name_off, N, d.names_bytes),
        pham_names=strings(d.pham_names, d.pham_name_off, P, d.pham_names_bytes),
        n_genomes=N, n_phams=P, words_per_row=W, bitmap=arr(d.bitmap, N * W), nph=arr(d.nph, N), ngen=arr(d.ngen, N),
        tlen=arr(d.tlen, N), gene_off=arr(d.gene_off, N + 1), gene_pham=arr(d.gene_pham, G), seq_off=arr(d.seq_off, G + 1),
        residues=arr(d.residues, R)).validate()
    if keep_handle:
        packed.gene_order = arr(d.gene_order, G)
        packed._keepalive.append(owner)
    return packed, owner


def load_tsv_packed(filepath):
    """Parse the reference's 2/3-column TSV (scripts/phamclust.py:21-47) straight into packed form,
    genomes sorted by name (scripts/phamclust.py:221).  Equals pack_genomes(sorted(load_genomes_from_tsv))."""
    return _load_tsv(filepath, keep_handle=False)[0]


def _plain_genome(name, phams):
    genome = Genome(name)
    genome.phams = phams
    return genome


class LazyGenome(Genome):
    """A ``Genome`` whose genes still live in the loader's packed arrays.  ``phams`` (pham -> translations, in the
    genome's own insertion order) is built on first use; the FASTA text -- what the pipeline hashes and stashes --
    comes straight from the loader.  Touching ``phams`` detaches the object from the packed form (it may have been
    edited), after which it behaves like any other ``Genome``."""

    def __init__(self, name, packed, owner, index):
        self.name = name
        self._packed, self._owner, self._index = packed, owner, index
        self._phams = None

    def _read_phams(self):
        pk, g = self._packed, self._index
        k0, k1 = int(pk.gene_off[g]), int(pk.gene_off[g + 1])
        genes = sorted(range(k0, k1), key=lambda k: int(pk.gene_order[k]))
        text = pk.residues[int(pk.seq_off[k0]):int(pk.seq_off[k1])].tobytes().decode("latin-1")
        base = int(pk.seq_off[k0])
        found = {}
        for k in genes:
            found.setdefault(pk.pham_names[int(pk.gene_pham[k])], []).append(
                text[int(pk.seq_off[k]) - base:int(pk.seq_off[k + 1]) - base])
        return found

    @property
    def phams(self):
        if self._phams is None:
            self._phams = self._read_phams()
        return self._phams

    def __reduce__(self):
        """Pickled (joblib workers of the generic ``matrix_de_novo`` path), a lazy genome travels as a plain ``Genome``: the
        loader's arrays and its C handle stay in this process, and the object itself stays attached to them."""
        return _plain_genome, (self.name, self._read_phams() if self._phams is None else self._phams)

    @phams.setter
    def phams(self, value):
        self._phams = value

    def is_packed(self):
        return self._phams is None

    def fasta_bytes(self):
        if self._phams is None:
            return self._owner.fasta(self._index)
        return super().__str__().encode()

    def __str__(self):
        return self.fasta_bytes().decode() if self._phams is None else super().__str__()

    __repr__ = __str__

    def __len__(self):
        return int(self._packed.ngen[self._index]) if self._phams is None else super().__len__()

    def save(self, filepath):
        with open(filepath, "wb") as handle:
            handle.write(self.fasta_bytes())
        return filepath


def load_tsv_genomes(filepath):
    """The pipeline's loader: ``list[Genome]`` sorted by name, as scripts/phamclust.py:21-47,221 leave it, but the
    parsing, sorting, pham vocabulary, bitmap and residue packing are one pass of C (csrc/pc_pack.c), and the
    genomes are lazy views of that result -- ``matrix_de_novo`` uploads the packed arrays as they are."""
    packed, owner = _load_tsv(filepath, keep_handle=True)
    return [LazyGenome(name, packed, owner, k) for k, name in enumerate(packed.names)]


def packed_behind(genomes):
    """The loader's PackedGenomes if ``genomes`` is exactly its full, ordered, untouched list of lazy genomes."""
    first = genomes[0] if len(genomes) else None
    if not isinstance(first, LazyGenome):
        return None
    packed = first._packed
    if len(genomes) != packed.n_genomes:
        return None
    for k, genome in enumerate(genomes):
        if not isinstance(genome, LazyGenome) or genome._packed is not packed or genome._index != k or not genome.is_packed() \
                or genome.name != packed.names[k]:
            return None
    return packed
