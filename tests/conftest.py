import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")
SET_METRICS = ["gcs", "jc", "pocp", "af"]
ALL_METRICS = SET_METRICS + ["aai", "peq"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_tsv_genomes(path):
    """Same parsing rules as the reference loader (scripts/phamclust.py:21-47): 3 columns, or 2
    columns with the translation defaulting to "M"; returned sorted by name (:221)."""
    from phamclust_amd.genome import Genome
    genomes = {}
    with open(path) as handle:
        for line in handle:
            row = line.rstrip().split("\t")
            if len(row) == 3:
                name, pham, translation = row
            elif len(row) == 2:
                (name, pham), translation = row, "M"
            else:
                raise ValueError("input file must either 2 or 3 columns")
            genomes.setdefault(name, Genome(name)).add(pham, translation)
    return sorted(genomes.values(), key=lambda g: g.name)


def read_lower_triangle(path):
    """(names, condensed scipy-order vector) from a reference lower-triangle matrix file."""
    names, rows = [], []
    with open(path) as handle:
        n = int(handle.readline().split("\t")[0])
        for line in handle:
            fields = line.rstrip().split("\t")
            names.append(fields[0])
            rows.append([float(x) for x in fields[1:]])
    assert len(names) == n
    full = np.zeros((n, n))
    for i, row in enumerate(rows):
        assert len(row) == i + 1
        full[i, :i + 1] = row
    full = full + full.T - np.diag(np.diag(full))
    return names, full[np.triu_indices(n, k=1)], np.diag(full).copy()


def golden_file(metric, kind="distance"):
    tag = "" if metric in SET_METRICS else ".oracle_nw"
    if kind == "distance":
        return os.path.join(GOLDEN, f"{metric}_distance_matrix{tag}.tsv")
    return os.path.join(GOLDEN, f"pairwise_{metric}_similarities{tag}.tsv")


def read_adjacency_condensed(path, names):
    idx = {name: k for k, name in enumerate(names)}
    n = len(names)
    full = np.full((n, n), np.nan)
    with open(path) as handle:
        for line in handle:
            s, t, w = line.rstrip().split("\t")
            full[idx[s], idx[t]] = full[idx[t], idx[s]] = float(w)
    return full[np.triu_indices(n, k=1)], np.diag(full).copy()


@pytest.fixture(scope="session")
def small_genomes():
    return load_tsv_genomes(os.path.join(GOLDEN, "small_input.tsv"))


@pytest.fixture(scope="session")
def small_packed(small_genomes):
    from phamclust_amd.pack import pack_genomes
    return pack_genomes(small_genomes)


@pytest.fixture(scope="session")
def native_built():
    """Build the native pieces once (hipcc cross-compiles without a GPU)."""
    from phamclust_amd import build
    build.build_all()
    from oracle import oracle
    oracle.build()
    return True


@pytest.fixture(scope="session")
def gpu_ctx(native_built):
    from phamclust_amd import hip
    ctx = hip.Context(0)
    yield ctx
    ctx.close()
