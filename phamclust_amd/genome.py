"""``Genome``: the reference's dict/set hybrid (genome.py:12-225), same surface.

A genome maps pham name -> list of translations (insertion order kept, paralogs
appended).  Set algebra (``& | - ^``) works on pham names and returns ``set[str]``.
The GPU path never touches these objects pair by pair: ``phamclust_amd.pack`` turns a
name-sorted ``list[Genome]`` into flat arrays once, and the kernels work on those.
"""

from phamclust_amd.fasta import read_fasta


class GenomeLoadError(Exception):
    """Raised when a genome cannot be loaded from FASTA (reference genome.py:7-9)."""


def _require_genome(other):
    if not isinstance(other, Genome):
        raise TypeError(f"cannot compare Genome to '{type(other)}'")


class Genome:
    def __init__(self, name):
        self.name = name
        self.phams = dict()

    # -- construction -----------------------------------------------------
    def add(self, pham, translation="M"):
        """Append one gene (reference genome.py:31-49; 2-column inputs use "M")."""
        if not isinstance(pham, str):
            raise TypeError(f"type(pham) should be 'str', not '{type(pham)}'")
        if not isinstance(translation, str):
            raise TypeError(f"type(translation) should be 'str', not '{type(translation)}'")
        self.phams.setdefault(pham, []).append(translation)

    def load(self, fasta):
        """Add every record of a single-genome FASTA file; the pham of a record is the first ``pham=<id>`` item among the
        ``|``-separated ``key=value`` items of its header (the format ``save`` writes; reference genome.py:67-83).
        Items are read only up to that one, and each of them must hold exactly one ``=``."""
        def items(header):
            for item in header.split("|"):
                key, value = item.split("=")
                yield key, value

        for header, translation in read_fasta(fasta):
            pham = next((value for key, value in items(header) if key == "pham"), None)
            if pham is None:
                raise GenomeLoadError("unable to get pham from FASTA header")
            self.add(pham, translation)

    def pop(self, pham):
        if pham not in self:
            return None
        return self.phams.pop(pham)

    def save(self, filepath):
        with open(filepath, "w") as handle:
            handle.write(str(self))
        return filepath

    # -- set algebra on pham names ------------------------------------------
    def intersection(self, other):
        return self & other

    def union(self, other):
        return self | other

    def difference(self, other):
        return self - other

    def symmetric_difference(self, other):
        return self ^ other

    def __and__(self, other):
        _require_genome(other)
        return set(self.phams).intersection(other.phams)

    def __or__(self, other):
        _require_genome(other)
        return set(self.phams).union(other.phams)

    def __sub__(self, other):
        _require_genome(other)
        return set(self.phams).difference(other.phams)

    def __xor__(self, other):
        _require_genome(other)
        return set(self.phams).symmetric_difference(other.phams)

    # -- container protocol ---------------------------------------------------
    def __contains__(self, item):
        if not isinstance(item, str):
            raise TypeError(f"type(item) should be 'str', not '{type(item)}'")
        return item in self.phams

    def __getitem__(self, item):
        if item not in self:
            raise KeyError(f"node '{item}' not in matrix")
        return self.phams[item]

    def __iter__(self):
        yield from self.phams.items()

    def __len__(self):
        """Total number of genes, paralogs included (reference genome.py:168-169)."""
        return sum(len(genes) for genes in self.phams.values())

    def __lt__(self, other):
        _require_genome(other)
        return len(self) < len(other)

    def __str__(self):
        """FASTA text; also what the pipeline hashes (reference genome.py:192-199)."""
        parts = []
        for pham, translations in self:
            for n, translation in enumerate(translations, start=1):
                parts.append(f">name={self.name}|pham={pham}|n={n}\n{translation}\n")
        return "".join(parts)

    __repr__ = __str__
