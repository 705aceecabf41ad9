"""CPU tests of the co-optimal certificate (oracle/pc_cooptimal.c): an independent three-state DP that counts every
optimal global alignment of a sequence pair and the range of (n_ident, n_diag) over them.  Where the range is a single
point, ANY correct Needleman-Wunsch -- parasail.nw_trace_diag_16 included (reference metrics.py:160-175, 216-217) --
reports exactly these statistics: such alignments are pinned by mathematics, not by recalled tie rules."""

import json
import os
import random

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    oracle.build()
    return oracle


def _all_alignments(a, b):
    """Every global alignment as a string over M (pair), I (gap in a: consumes b), D (gap in b: consumes a)."""
    out = []

    def rec(i, j, ops):
        if i == len(a) and j == len(b):
            out.append("".join(ops))
            return
        if i < len(a) and j < len(b):
            rec(i + 1, j + 1, ops + ["M"])
        if j < len(b):
            rec(i, j + 1, ops + ["I"])
        if i < len(a):
            rec(i + 1, j, ops + ["D"])
    rec(0, 0, [])
    return out


def _stats(O, a, b, ops, open_=11, ext=1, ppos=False):
    L = O.lib()
    i = j = 0
    score = ident = diag = 0
    prev = None
    for op in ops:
        if op == "M":
            s = L.pco_blosum62(L.pco_map(ord(a[i])), L.pco_map(ord(b[j])))
            score += s
            ident += (a[i].upper() == b[j].upper()) or (ppos and s > 0)
            diag += 1
            i += 1
            j += 1
        else:
            score -= ext if prev == op else open_
            if op == "I":
                j += 1
            else:
                i += 1
        prev = op
    return score, ident, diag


@pytest.mark.parametrize("ppos", [False, True])
def test_certificate_equals_brute_force(O, ppos):
    """Score, number of optimal alignments and both ranges against exhaustive enumeration (<= 7 x 6 residues; a
    two-letter alphabet and poly-residue runs make most cases co-optimal)."""
    rng = random.Random(7)
    alphabets = ["AG", "LIV", "ACDEFGHIKLMNPQRSTVWY", "WwXBZ*J"]
    n_multi = n_uncertified = 0
    for it in range(700):
        alpha = alphabets[it % len(alphabets)]
        a = "".join(rng.choice(alpha) for _ in range(rng.randint(1, 7)))
        b = "".join(rng.choice(alpha) for _ in range(rng.randint(1, 6)))
        scored = [_stats(O, a, b, ops, ppos=ppos) for ops in _all_alignments(a, b)]
        best = max(s for s, _, _ in scored)
        opt = [(i, d) for s, i, d in scored if s == best]
        score, count, (id_lo, id_hi), (dg_lo, dg_hi) = O.cooptimal(a, b, ppos=ppos)
        assert score == best, (a, b)
        assert count == len(opt), (a, b, count, len(opt))
        assert (id_lo, id_hi) == (min(i for i, _ in opt), max(i for i, _ in opt)), (a, b)
        assert (dg_lo, dg_hi) == (min(d for _, d in opt), max(d for _, d in opt)), (a, b)
        n_multi += count > 1
        n_uncertified += (id_lo, dg_lo) != (id_hi, dg_hi)
    assert n_multi > 100 and n_uncertified > 20          # the enumeration really exercised ties


def test_other_gap_costs_against_brute_force(O):
    rng = random.Random(11)
    try:
        for open_, ext in ((12, 1), (5, 2), (3, 3)):
            O.set_gap(open_, ext)
            for _ in range(120):
                a = "".join(rng.choice("AGW") for _ in range(rng.randint(1, 6)))
                b = "".join(rng.choice("AGW") for _ in range(rng.randint(1, 6)))
                scored = [_stats(O, a, b, ops, open_, ext) for ops in _all_alignments(a, b)]
                best = max(s for s, _, _ in scored)
                opt = [(i, d) for s, i, d in scored if s == best]
                score, count, idr, dgr = O.cooptimal(a, b)
                assert (score, count) == (best, len(opt))
                assert idr == (min(i for i, _ in opt), max(i for i, _ in opt))
                assert O.nw_stats(a, b)[0] == best          # the oracle's aligner follows the same run-time gap costs
    finally:
        O.set_gap(11, 1)


def test_every_tie_rule_returns_an_optimal_alignment(O):
    """All 16 rule combinations of the oracle's aligner trace SOME optimal alignment: their statistics lie inside the
    certificate's ranges, their score is the optimum; on certified pairs all 16 agree."""
    rng = random.Random(3)
    aa = "ACDEFGHIKLMNPQRSTVWY"
    certified = uncertified = 0
    for it in range(600):
        alpha = aa[:3] if it % 2 else aa
        a = "".join(rng.choice(alpha) for _ in range(rng.randint(1, 60)))
        b = "".join(rng.choice(alpha) for _ in range(rng.randint(1, 60)))
        score, count, (id_lo, id_hi), (dg_lo, dg_hi) = O.cooptimal(a, b)
        seen = set()
        for rule in range(O.N_TIE_RULES):
            with O.tie_rule(rule):
                sc, ident, diag = O.nw_stats(a, b)
            assert sc == score
            assert id_lo <= ident <= id_hi and dg_lo <= diag <= dg_hi, (a, b, rule)
            seen.add((ident, diag))
        if (id_lo, dg_lo) == (id_hi, dg_hi):
            certified += 1
            assert seen == {(id_lo, dg_lo)}
        else:
            uncertified += 1
        if count == 1:
            assert (id_lo, dg_lo) == (id_hi, dg_hi)
    assert certified > 100 and uncertified > 50


def test_unique_optimum_report_is_current(O):
    """tests/golden/unique_optimum.json (tools/unique_optimum.py) on the small fixture: recomputed here, must agree."""
    path = os.path.join(GOLDEN, "unique_optimum.json")
    with open(path) as fh:
        report = json.load(fh)
    from phamclust_amd.pack import pack_genomes
    from conftest import load_tsv_genomes
    packed = pack_genomes(load_tsv_genomes(os.path.join(GOLDEN, "small_input.tsv")))
    iu = np.triu_indices(packed.n_genomes, 1)
    a, b, q = O.enumerate_alignments(packed, iu[0], iu[1])
    _, count, rng = O.cooptimal_batch(packed.residues, packed.seq_off, a, b)
    cert = (rng[:, 0] == rng[:, 1]) & (rng[:, 2] == rng[:, 3])
    ds = next(d for d in report["datasets"] if d["name"].endswith("small_input.tsv"))
    assert ds["alignments"] == a.shape[0]
    assert ds["alignments_unique_optimum"] == int((count == 1).sum())
    assert ds["alignments_certified"] == int(cert.sum())
    bad_pairs = np.unique(q[~cert])
    with_aln = np.unique(q)
    assert ds["genome_pairs_with_alignments"] == with_aln.shape[0]
    assert ds["genome_pairs_fully_certified"] == with_aln.shape[0] - bad_pairs.shape[0]
