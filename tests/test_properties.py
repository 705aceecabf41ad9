"""Property tests (hypothesis; CPU): what must hold for ANY input, checked on generated ones -- the reference has no tests
at all, so these stand where its invariants would.  The checker and the host code only; GPU parity is in
test_gpu_parity.py."""

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

AA = "ACDEFGHIKLMNPQRSTVWY"
seqs = st.text(alphabet=AA + "BZX*acdJU", min_size=1, max_size=40)
few_letter_seqs = st.text(alphabet="AGW", min_size=1, max_size=30)          # co-optimal ties everywhere
genome = st.dictionaries(st.sampled_from([f"p{i}" for i in range(12)]), st.lists(seqs, min_size=1, max_size=3), min_size=1, max_size=8)
SETTINGS = dict(deadline=None, max_examples=60, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture], derandomize=True)   # the same examples on every run: a CI gate must not flip on a dice roll (new seeds: --hypothesis-seed=N)


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    oracle.build()
    return oracle


def _genomes(dicts):
    from phamclust_amd.genome import Genome
    out = []
    for k, phams in enumerate(dicts):
        g = Genome(f"g{k:02d}")
        for pham, translations in phams.items():
            for t in translations:
                g.add(pham, t)
        out.append(g)
    return out


@settings(**SETTINGS)
@given(st.lists(genome, min_size=2, max_size=5))
def test_oracle_c_equals_python_restatement_and_metric_identities(O, dicts):
    """Packed closed forms (C) == per-pair dict/set semantics (Python, metrics.py:26-253) for all six metrics; the four set
    metrics are symmetric in (source, target); peq is round(round(af) * round(aai)) (metrics.py:247-253); a genome against
    itself scores 1 everywhere."""
    from phamclust_amd.pack import pack_genomes
    genomes = _genomes(dicts)
    packed = pack_genomes(genomes)
    n = len(genomes)
    sim = {m: O.fill(packed, m, as_distance=False) for m in ("gcs", "jc", "pocp", "af", "aai", "peq")}
    k = 0
    for i in range(n):
        for j in range(i + 1, n):
            for m in ("gcs", "jc", "pocp", "af", "aai", "peq"):
                assert sim[m][k] == O.PY_METRICS[m](genomes[i], genomes[j], as_distance=False), m
            for m in ("gcs", "jc", "pocp", "af"):
                assert O.PY_METRICS[m](genomes[j], genomes[i]) == sim[m][k], m
            # (Python floats: numpy's float64.__round__ is rint(x * 1e6) / 1e6, not CPython's correctly rounded round())
            assert sim["peq"][k] == round(float(sim["af"][k]) * float(sim["aai"][k]), 6)
            assert (sim["gcs"][k] == 0.0) == (sim["jc"][k] == 0.0) == (sim["pocp"][k] == 0.0)
            assert 0.0 <= sim["jc"][k] <= sim["gcs"][k] <= 1.0
            k += 1
    for g in genomes:
        for m in ("gcs", "jc", "pocp", "af", "aai", "peq"):
            assert O.PY_METRICS[m](g, g) == 1.0, m
    dist = O.fill(packed, "peq", as_distance=True)                      # metrics.py:250-253: 1 - af * aai of the two ROUNDED factors, rounded once
    assert np.array_equal(dist, np.array([round(1.0 - float(a) * float(b), 6) for a, b in zip(sim["af"], sim["aai"])]))


@settings(**SETTINGS)
@given(st.one_of(seqs, few_letter_seqs), st.one_of(seqs, few_letter_seqs))
def test_aligner_and_certificate(O, a, b):
    """Both aligner formulations agree; the certificate's optimum is the aligner's score, symmetric in (a, b) like the matrix
    and the gap costs, its count too; every tie rule's statistics lie inside its ranges; n_ident <= n_diag <= min(la, lb)."""
    sc, ident, diag = O.nw_stats(a, b)
    tb = O.nw_traceback(a, b)
    assert (tb.score, tb.comp.count("|"), len(tb.query)) == (sc, ident, len(a) + len(b) - diag)
    assert tb.query.replace("-", "") == a and tb.ref.replace("-", "") == b
    score, count, (id_lo, id_hi), (dg_lo, dg_hi) = O.cooptimal(a, b)
    score2, count2, idr2, dgr2 = O.cooptimal(b, a)
    assert score == sc == score2 and count == count2 >= 1 and (id_lo, id_hi) == idr2 and (dg_lo, dg_hi) == dgr2
    assert 0 <= id_lo <= id_hi <= dg_hi <= min(len(a), len(b)) and dg_lo <= dg_hi
    for rule in range(8):
        with O.tie_rule(rule):
            s2, i2, d2 = O.nw_stats(a, b)
        assert s2 == sc and id_lo <= i2 <= id_hi and dg_lo <= d2 <= dg_hi
    if count == 1:
        assert (id_lo, dg_lo) == (id_hi, dg_hi) == (ident, diag)


@settings(**SETTINGS)
@given(st.floats(min_value=0.0, max_value=1.0, allow_nan=False), st.integers(0, 10 ** 6), st.integers(1, 10 ** 6))
def test_round6_is_pythons(O, x, num, den):
    assert O.round6(x) == round(x, 6)
    q = min(num, den) / den
    assert O.round6(q) == round(q, 6) and O.round6(1.0 - q) == round(1.0 - q, 6)
    tie = (num % 10 ** 6 + 0.5) / 1e6
    assert O.round6(tie) == round(tie, 6)


@settings(**SETTINGS)
@given(st.lists(st.integers(0, 2 ** 40), min_size=0, max_size=200), st.integers(1, 2 ** 41))
def test_chunk_plan_properties(native_built, counts, limit):
    """The cutting rule of memory-bounded fills: the ranges tile [0, n), none is empty, a range exceeds the limit only when it
    is a single element, and no two neighbours could have been merged."""
    from phamclust_amd import hip
    cuts = hip.Context.chunk_plan(counts, limit)
    n = len(counts)
    assert cuts[0] == 0 and cuts[-1] == n and (np.diff(cuts) > 0).all() if n else cuts.tolist() == [0]
    sums = [sum(counts[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    for (a, b), s in zip(zip(cuts[:-1], cuts[1:]), sums):
        assert s <= limit or b - a == 1
    for k in range(len(sums) - 1):
        assert sums[k] + counts[cuts[k + 1]] > limit


line = st.tuples(st.text(alphabet="abcXYZ_09", min_size=1, max_size=6), st.text(alphabet="pq12", min_size=1, max_size=3),
                 st.one_of(st.none(), st.text(alphabet=AA + "xb*", min_size=1, max_size=25)))


@settings(**SETTINGS)
@given(st.lists(line, min_size=1, max_size=40))
def test_c_loader_equals_python_loader(native_built, tmp_path_factory, rows):
    """csrc/pc_pack.c == the reference's loading rules (scripts/phamclust.py:21-47, 221) + pack_genomes, on generated 2- and
    3-column files (a missing translation defaults to "M")."""
    from conftest import load_tsv_genomes
    from phamclust_amd import pack
    path = tmp_path_factory.mktemp("prop") / "in.tsv"
    with open(path, "w") as fh:
        for name, pham, translation in rows:
            fh.write(f"{name}\t{pham}\n" if translation is None else f"{name}\t{pham}\t{translation}\n")
    want = pack.pack_genomes(load_tsv_genomes(path))
    got = pack.load_tsv_packed(path)
    assert got.names == want.names and got.pham_names == want.pham_names
    for name in ("bitmap", "nph", "ngen", "tlen", "gene_off", "gene_pham", "seq_off", "residues"):
        assert np.array_equal(getattr(got, name), getattr(want, name)), name
    lazy = pack.load_tsv_genomes(path)
    assert [str(g) for g in lazy] == [str(g) for g in load_tsv_genomes(path)]
