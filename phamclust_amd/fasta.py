"""Minimal FASTA reader/writer (reference fasta.py:4-73 surface)."""


def read_fasta(filename):
    """Yield ``(header, sequence)`` records; the header has its ``>`` stripped and
    multi-line sequences are joined.  Raises ``ValueError`` when the file does not
    start with a ``>`` record (reference fasta.py:21-22)."""
    header, chunks = None, []
    with open(filename, "r") as handle:
        for line in handle:
            if line.startswith(">"):
                if header is not None:
                    yield header, "".join(chunks)
                header, chunks = line[1:].rstrip(), []
            else:
                if header is None:
                    raise ValueError("records in FASTA files must start with '>'")
                chunks.append(line.rstrip())
    if header is not None:
        yield header, "".join(chunks)


def write_fasta(records, filename, mode="w", wrap=None):
    """Write ``(header, sequence)`` records, optionally wrapping sequence lines."""
    with open(filename, mode) as handle:
        for header, sequence in records:
            handle.write(f">{header}\n")
            if wrap is None or wrap <= 0:
                handle.write(f"{sequence}\n")
            else:
                for start in range(0, len(sequence), wrap):
                    handle.write(sequence[start:start + wrap] + "\n")
    return filename
