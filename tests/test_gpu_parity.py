"""GPU parity tests (run with ``-m gpu`` on the MI355X box).  Every value computed here comes
through the C-ABI of libphamclust_hip.so; the oracle and the golden fixtures are the checkers.

Bars (north_star): gcs / jc / pocp bit-exact; af / aai / peq within 1e-6 (in practice the
HIP path reproduces the oracle bit for bit, which these tests also record); integer
alignment outputs (n_ident, n_diag) exactly equal to the oracle's.
aai / peq fixtures are "oracle_nw" class: parity vs parasail's co-optimal ties is unpinned.
"""

import os

import numpy as np
import pytest

from conftest import ALL_METRICS, SET_METRICS, golden_file, read_adjacency_condensed, read_lower_triangle

pytestmark = pytest.mark.gpu


def _py_round6(arr):
    """CPython's ``round(v, 6)`` of every element -- the reference's rounding (e.g. metrics.py:50-53) -- without 12.5 M interpreter
    calls: ``rint(v * 1e6) / 1e6`` IS that value whenever v * 1e6 is not within 1e-3 of a half (the scaled value's rounding error is
    < 1e-9 there, so both pick the same integer k, and k / 1e6 is one correctly-rounded division either way); the elements that are
    near a half, plus a random 50,000 as a guard on this argument, go through CPython's round itself."""
    arr = np.asarray(arr, dtype=np.float64)
    scaled = arr * 1e6
    out = np.rint(scaled) / 1e6
    near_half = np.abs(np.abs(scaled - np.floor(scaled)) - 0.5) < 1e-3
    idx = np.flatnonzero(near_half)
    out[idx] = [round(v, 6) for v in arr[idx].tolist()]
    guard = np.random.default_rng(6).integers(0, arr.size, min(arr.size, 50000))
    assert np.array_equal(out[guard], np.array([round(v, 6) for v in arr[guard].tolist()]))
    return out
TOL = 1e-6


def _oracle():
    from oracle import oracle
    return oracle


@pytest.mark.parametrize("metric", ALL_METRICS)
def test_golden_distance(gpu_ctx, small_packed, metric):
    names, gold, _ = read_lower_triangle(golden_file(metric))
    assert names == small_packed.names
    got = gpu_ctx.upload(small_packed).fill(metric, as_distance=True)
    if metric in ("gcs", "jc", "pocp"):
        assert np.array_equal(got, gold)
    else:
        assert np.max(np.abs(got - gold)) <= TOL
        assert np.array_equal(got, gold), "HIP path is expected to reproduce the fixture bit for bit"


@pytest.mark.parametrize("metric", ALL_METRICS)
def test_golden_similarity(gpu_ctx, small_packed, metric):
    gold, _ = read_adjacency_condensed(golden_file(metric, "similarity"), small_packed.names)
    got = gpu_ctx.upload(small_packed).fill(metric, as_distance=False)
    if metric in ("gcs", "jc", "pocp"):
        assert np.array_equal(got, gold)
    else:
        assert np.max(np.abs(got - gold)) <= TOL


@pytest.mark.parametrize("n,p,seed", [(97, 600, 3), (200, 5000, None)])
@pytest.mark.parametrize("metric", ALL_METRICS)
def test_synth_vs_oracle(gpu_ctx, native_built, metric, n, p, seed):
    from phamclust_amd.synth import synth_packed
    O = _oracle()
    packed = synth_packed(n, p, seed)
    got = gpu_ctx.upload(packed).fill(metric, as_distance=True)
    want = O.fill(packed, metric, as_distance=True)
    assert got.shape == want.shape == (n * (n - 1) // 2,)
    assert np.array_equal(got, want)
    assert got.min() >= 0.0 and got.max() <= 1.0


@pytest.mark.parametrize("variant", [0, -1])
def test_align_pairs_integer_outputs(gpu_ctx, native_built, variant):
    """(n_ident, n_diag) of every alignment == the oracle's, exactly."""
    from phamclust_amd.synth import synth_packed
    O = _oracle()
    packed = synth_packed(60, 300, seed=11)
    rng = np.random.default_rng(5)
    g = packed.n_genes
    a = rng.integers(0, g, 4000).astype(np.int32)
    b = rng.integers(0, g, 4000).astype(np.int32)
    b[:500] = b[0]                                   # a long bucket: many rows against one column gene
    gpu_ctx.upload(packed)
    ident, diag = gpu_ctx.align_pairs(a, b, variant=variant)
    _, want_i, want_d = O.nw_batch(packed.residues, packed.seq_off, a, b)
    assert np.array_equal(ident, want_i)
    assert np.array_equal(diag, want_d)


@pytest.mark.parametrize("w", [2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 22, 24, 32, 48, 64])
def test_every_systolic_variant(gpu_ctx, native_built, w):
    """Each compiled columns-per-lane variant, forced, on column genes from 1 residue up to its
    64*w limit, many rows per column gene (streams of back-to-back alignments, several segments)."""
    from phamclust_amd.genome import Genome
    from phamclust_amd.pack import pack_genomes
    O = _oracle()
    rng = np.random.default_rng(100 + w)
    aa = np.array(list("ACDEFGHIKLMNPQRSTVWY"))
    cap = min(64 * w, 900 if w < 32 else 64 * w)
    lens = sorted(set([1, 2, w, w + 1, 2 * w - 1, 16 * w, 16 * w + 1, 21 * w, 32 * w, 32 * w + 1, cap - 1, cap]
                      + rng.integers(1, cap + 1, 6).tolist()))
    lens = [x for x in lens if 1 <= x <= cap]
    g = Genome("cols")
    for i, ln in enumerate(lens):
        g.add(f"c{i:02d}", "".join(aa[rng.integers(0, 20, ln)]))
    h = Genome("rows")
    for i in range(40):
        base = g.phams[f"c{i % len(lens):02d}"][0]
        cut = rng.integers(0, max(1, len(base)))
        h.add(f"r{i:02d}", (base[:cut] + "".join(aa[rng.integers(0, 20, rng.integers(0, 4))]) + base[cut + rng.integers(0, 3):]) or "M")
    pk = pack_genomes([g, h])
    ncol = len(lens)
    a = np.repeat(np.arange(ncol, ncol + 40, dtype=np.int32), ncol)      # rows: genes of "rows"
    b = np.tile(np.arange(ncol, dtype=np.int32), 40)                      # cols: genes of "cols"
    gpu_ctx.upload(pk)
    ident, diag = gpu_ctx.align_pairs(a, b, variant=w)
    _, wi, wd = O.nw_batch(pk.residues, pk.seq_off, a, b)
    assert np.array_equal(ident, wi) and np.array_equal(diag, wd)


_CELL_CHECK = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import oracle as O
from phamclust_amd import hip
from phamclust_amd.genome import Genome
from phamclust_amd.pack import pack_genomes
ctx = hip.Context(0)
rng = np.random.default_rng(42)
aa = np.array(list("ACDEFGHIKLMNPQRSTVWYBZX*"))
bad = 0
for w in (3, 8, 12, 13, 16, 19, 20, 24):
    lens = sorted({1, w, 8 * w, 8 * w + 1, 16 * w, 16 * w + 1, 32 * w, 32 * w + 1, min(64 * w, 1200)})
    g = Genome("cols")
    for i, ln in enumerate(lens):
        g.add(f"c{i:02d}", "".join(aa[rng.integers(0, aa.size, ln)]))
    h = Genome("rows")
    for i in range(70):
        base = g.phams[f"c{i % len(lens):02d}"][0]
        cut = int(rng.integers(0, max(1, len(base))))
        h.add(f"r{i:02d}", (base[:cut] + "".join(aa[rng.integers(0, aa.size, int(rng.integers(0, 4)))]) + base[cut + int(rng.integers(0, 3)):]) or "M")
    pk = pack_genomes([g, h])
    n = len(lens)
    a = np.repeat(np.arange(n, n + 70, dtype=np.int32), n); b = np.tile(np.arange(n, dtype=np.int32), 70)
    ctx.upload(pk)
    ident, diag = ctx.align_pairs(a, b, variant=w)
    _, wi, wd = O.nw_batch(pk.residues, pk.seq_off, a, b)
    ok = bool(np.array_equal(ident, wi) and np.array_equal(diag, wd))
    bad += not ok
    print("w", w, "ok" if ok else "MISMATCH", flush=True)
sys.exit(1 if bad else 0)
"""


@pytest.mark.parametrize("inc16", ["0", "1"])
def test_both_cells_everywhere(native_built, inc16):
    """The systolic variants up to W = 24 exist with two cells (residue compare, 11 instructions; increments from a
    16-bit profile, 10) and the launcher picks one per launch class.  PC_INC16 forces one of them on every class that
    can run it -- segments of 1..32 lanes for the profile cell (workgroups of 4 and 8 waves), all for the compare
    cell -- in a process of its own (the switch is read once)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PC_INC16=inc16)
    p = subprocess.run([sys.executable, "-c", _CELL_CHECK, root], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert p.stdout.count(" ok") == 8


def test_bytes_outside_the_alphabet(gpu_ctx, native_built):
    """Residues that are no BLOSUM62 letter (U, O, J, digits ...) score as '*' and are identical only to the same byte
    (metrics.py:216 counts '|' of a character comparison).  As COLUMN residues they send the gene to its variant's
    "any byte" launch class, which runs the residue-compare cell (the 16-bit increment profile has one row for all of
    them); as row residues they change nothing."""
    from phamclust_amd.genome import Genome
    from phamclust_amd.pack import pack_genomes
    O = _oracle()
    rng = np.random.default_rng(99)
    aa = np.array(list("ACDEFGHIKLMNPQRSTVWY" * 4 + "UOJ7u"))
    pool = ["".join(aa[rng.integers(0, aa.size, int(rng.integers(5, 260)))]) for _ in range(12)]
    pool += ["UUUUUUUU", "AUAUAUAUOJ", "ACDEFGHIKL" * 20]
    genomes = []
    for gi in range(9):
        g = Genome(f"g{gi}")
        for p in range(6):
            seq = pool[int(rng.integers(0, len(pool)))]
            if rng.random() < 0.5:
                cut = int(rng.integers(0, len(seq)))
                seq = seq[:cut] + "U" + seq[cut + 1:]
            g.add(f"pham{p}", seq)
        genomes.append(g)
    pk = pack_genomes(genomes)
    gpu_ctx.upload(pk)
    for metric in ("aai", "peq"):
        assert np.array_equal(gpu_ctx.fill(metric, True), O.fill(pk, metric, True)), metric
    n = pk.n_genes
    a = rng.integers(0, n, 600).astype(np.int32); b = rng.integers(0, n, 600).astype(np.int32)
    for variant in (0, 13, 19, 32, -1):
        ident, diag = gpu_ctx.align_pairs(a, b, variant=variant)
        _, wi, wd = O.nw_batch(pk.residues, pk.seq_off, a, b)
        assert np.array_equal(ident, wi) and np.array_equal(diag, wd), variant


@pytest.mark.parametrize("w", [4, 13])
def test_big_buckets_all_segment_counts(gpu_ctx, native_built, w):
    """Column genes whose lanes-per-segment run over 1..64 (so 1..16 segments per wave), each with
    300 row sequences: tasks of 208 + 92 rows spread over 4 waves x nseg segments."""
    from phamclust_amd.genome import Genome
    from phamclust_amd.pack import pack_genomes
    O = _oracle()
    rng = np.random.default_rng(7 + w)
    aa = np.array(list("ACDEFGHIKLMNPQRSTVWY"))
    gs = (1, 2, 3, 4, 5, 6, 7, 9, 12, 13, 16, 21, 22, 32, 33, 64)
    g = Genome("cols")
    for i, lanes in enumerate(gs):
        g.add(f"c{i:02d}", "".join(aa[rng.integers(0, 20, w * lanes - (i % 2))]))
    h = Genome("rows")
    for i in range(300):
        h.add(f"r{i:03d}", "".join(aa[rng.integers(0, 20, int(rng.integers(1, 60)))]))
    pk = pack_genomes([g, h])
    ncol = len(gs)
    a = np.repeat(np.arange(ncol, ncol + 300, dtype=np.int32), ncol)
    b = np.tile(np.arange(ncol, dtype=np.int32), 300)
    gpu_ctx.upload(pk)
    ident, diag = gpu_ctx.align_pairs(a, b, variant=w)
    _, wi, wd = O.nw_batch(pk.residues, pk.seq_off, a, b)
    assert np.array_equal(ident, wi) and np.array_equal(diag, wd)


@pytest.mark.parametrize("rule", list(range(8)))
def test_tie_rule_table(gpu_ctx, native_built, rule):
    """Every row of the aligner's tie-rule table (include/phamclust_hip.h, pc_set_tie_rule): the systolic variants,
    the general kernel and a whole aai / peq fill equal the oracle switched to the same rule (pco_set_tie_rule), on
    tie-heavy sequences (3-letter alphabet: co-optimal alignments in most pairs).  Rule 0 = SURVEY 8c as recalled;
    parasail itself is absent, so which row it follows is unpinned -- tests/golden/tie_sensitivity.json holds what the
    choice is worth."""
    from phamclust_amd.genome import Genome
    from phamclust_amd.pack import pack_genomes
    from phamclust_amd.synth import synth_packed
    O = _oracle()
    rng = np.random.default_rng(900 + rule)
    g, h = Genome("cols"), Genome("rows")
    for i in range(24):
        alpha = np.array(list("AGS" if i % 2 == 0 else "ACDEFGHIKLMNPQRSTVWY"))
        g.add(f"c{i:02d}", "".join(alpha[rng.integers(0, len(alpha), int(rng.integers(1, 190)))]))
    for i in range(60):
        alpha = np.array(list("AGS" if i % 3 else "ACDEFGHIKLMNPQRSTVWY"))
        h.add(f"r{i:02d}", "".join(alpha[rng.integers(0, len(alpha), int(rng.integers(1, 190)))]))
    pk = pack_genomes([g, h])
    a = np.repeat(np.arange(24, 84, dtype=np.int32), 24)
    b = np.tile(np.arange(24, dtype=np.int32), 60)
    small = synth_packed(40, 200, seed=77)
    try:
        gpu_ctx.set_tie_rule(rule)
        assert gpu_ctx.tie_rule() == rule
        O.set_tie_rule(rule)
        _, wi, wd = O.nw_batch(pk.residues, pk.seq_off, a, b)
        gpu_ctx.upload(pk)
        for variant in (0, -1, 3, 8, 13, 24, 48):
            ident, diag = gpu_ctx.align_pairs(a, b, variant=variant)
            assert np.array_equal(ident, wi) and np.array_equal(diag, wd), f"rule {rule} variant {variant}"
        gpu_ctx.upload(small)
        for metric in ("aai", "peq"):
            assert np.array_equal(gpu_ctx.fill(metric, as_distance=True), O.fill(small, metric, as_distance=True)), f"rule {rule} {metric}"
        if rule:                                     # the switch is not a no-op: this data has ties the rule decides
            O.set_tie_rule(0)
            _, zi, zd = O.nw_batch(pk.residues, pk.seq_off, a, b)
            assert not (np.array_equal(zi, wi) and np.array_equal(zd, wd))
    finally:
        O.set_tie_rule(0)
        gpu_ctx.set_tie_rule(0)


def test_int16_saturation_threshold(gpu_ctx, native_built):
    """The reference aligns with parasail's 16-bit kernel (`nw_trace_diag_16`, metrics.py:174) and never looks at its
    saturation flag: once a DP cell passes +32,767 (BLOSUM62's best per-residue score is W/W = 11, so from 2,979
    identical tryptophans; about 6,000 residues of ordinary protein identical to itself) parasail's table, and with it
    the reference's identity and length, is garbage.  This build keeps scores in 32 bits on purpose: below
    the threshold the two cannot differ (every int16 operation is exact there), above it this path stays exact.  Checked
    here on both sides of the line against the 32-bit oracle and against closed forms."""
    from phamclust_amd.genome import Genome
    from phamclust_amd.pack import pack_genomes
    O = _oracle()
    lens = (2970, 2978, 2979, 3300)                         # 11 * 2978 = 32,758 < 32,767 < 32,769 = 11 * 2979
    g, h = Genome("cols"), Genome("rows")
    for i, n in enumerate(lens):
        g.add(f"c{i}", "W" * n)
        h.add(f"r{i}a", "W" * n)                             # identical: score 11 n, identity n / n
        h.add(f"r{i}b", "W" * (n - 9) + "A" * 9)             # 9 substitutions at the end: no gap can beat them (W/A = -3 > -11)
        h.add(f"r{i}c", "W" * (n - 40))                      # a 40-residue deletion: one gap run
    pk = pack_genomes([g, h])
    ncol = len(lens)
    a = np.arange(ncol, ncol + 3 * ncol, dtype=np.int32)
    b = np.repeat(np.arange(ncol, dtype=np.int32), 3)
    gpu_ctx.upload(pk)
    ident, diag = gpu_ctx.align_pairs(a, b)
    score, wi, wd = O.nw_batch(pk.residues, pk.seq_off, a, b)
    assert np.array_equal(ident, wi) and np.array_equal(diag, wd)
    for k, n in enumerate(lens):
        assert score[3 * k] == 11 * n and (score[3 * k] > 32767) == (n >= 2979)
        assert ident[3 * k] == n and diag[3 * k] == n                                   # 100 % identity, no gaps
        assert ident[3 * k + 1] == n - 9 and diag[3 * k + 1] == n                       # (n - 9) / n
        assert ident[3 * k + 2] == n - 40 and diag[3 * k + 2] == n - 40                 # (n - 40) / n, aln_len == n


def test_round6_matches_python(gpu_ctx):
    rng = np.random.default_rng(9)
    xs = np.concatenate([
        rng.random(200000),
        (rng.integers(0, 1000000, 100000) + 0.5) / 1e6,                 # decimal ties (inexact in binary)
        rng.integers(0, 2 ** 20, 100000) / 2.0 ** rng.integers(1, 24, 100000),   # exact binary ties
        rng.integers(0, 400, 100000) / np.maximum(1, rng.integers(1, 700, 100000)),
        np.array([0.0, 1.0, 0.5, 1 / 128, 3 / 128, 1 / 640, 5e-7, 4.9999999e-7, 1e-300, 0.9999995, 0.9999994999]),
    ])
    xs = xs[xs <= 1.0]
    got = gpu_ctx.round6(xs)
    want = np.array([round(float(x), 6) for x in xs])
    assert np.array_equal(got, want)


def test_round6_every_millionth(gpu_ctx):
    """round(x, 6) of every k / 1e6, k in [0, 2 * 10^6] (it rounds to itself), and of x just off each of them, against CPython."""
    k = np.arange(0, 2_000_001, dtype=np.float64)
    xs = k / 1e6
    assert np.array_equal(gpu_ctx.round6(xs), xs)                        # x = k / 1e6 rounds to itself
    rng = np.random.default_rng(21)
    near = xs[:1_000_001] + rng.uniform(-4e-7, 4e-7, 1_000_001)
    near = near[(near >= 0.0) & (near <= 1.0)][::7]
    assert np.array_equal(gpu_ctx.round6(near), np.array([round(float(x), 6) for x in near]))


def test_edge_cases(gpu_ctx, native_built):
    from phamclust_amd.genome import Genome
    from phamclust_amd.pack import pack_genomes
    O = _oracle()
    # single genome: no pairs
    g0 = Genome("only"); g0.add("p1", "MKV")
    out = gpu_ctx.upload(pack_genomes([g0])).fill("peq")
    assert out.shape == (0,)
    # two genomes sharing nothing: every metric similarity 0 -> distance 1
    g1 = Genome("a"); g1.add("p1", "MKV")
    g2 = Genome("b"); g2.add("p2", "MKV")
    pk = pack_genomes([g1, g2])
    for metric in ALL_METRICS:
        assert gpu_ctx.upload(pk).fill(metric, as_distance=True)[0] == 1.0
        assert gpu_ctx.upload(pk).fill(metric, as_distance=False)[0] == 0.0
    # all-"M" two-column style input: aai == 1 wherever phams are shared, peq == af
    gs = []
    for k in range(6):
        g = Genome(f"m{k}")
        for p in range(k, k + 4):
            g.add(f"p{p}")
        gs.append(g)
    pk = pack_genomes(gs)
    gpu_ctx.upload(pk)
    aai = gpu_ctx.fill("aai", as_distance=False)
    af = gpu_ctx.fill("af", as_distance=False)
    pocp = gpu_ctx.fill("pocp", as_distance=False)
    peq = gpu_ctx.fill("peq", as_distance=False)
    shared = gpu_ctx.fill("gcs", as_distance=False) > 0
    assert np.array_equal(aai[shared], np.ones(shared.sum()))
    assert np.array_equal(aai[~shared], np.zeros((~shared).sum()))
    assert np.array_equal(af, pocp)
    assert np.array_equal(peq, af)
    # empty translation under aai: refused loudly, like the reference (parasail cannot align it)
    from phamclust_amd.hip import HipLibraryError
    g3 = Genome("e1"); g3.add("p1", "")
    g4 = Genome("e2"); g4.add("p1", "MK")
    gpu_ctx.upload(pack_genomes([g3, g4]))
    with pytest.raises(HipLibraryError):
        gpu_ctx.fill("aai")
    assert gpu_ctx.fill("af", as_distance=False)[0] == 1.0


def test_random_small_sets_all_metrics(gpu_ctx, native_built):
    """150 random little collections (2-9 genomes, few phams so that sharing, paralogs and byte-identical
    sequences are common; lengths 1-90; residues drawn from the alphabet plus B Z X * U, lower case and '-'):
    all six metrics, both polarities, against the oracle.  Exercises the plan's corner cases (one alignment,
    one distinct sequence, every alignment identical, single-row buckets)."""
    from phamclust_amd.genome import Genome
    from phamclust_amd.pack import pack_genomes
    O = _oracle()
    rng = np.random.default_rng(2024)
    letters = np.array(list("ACDEFGHIKLMNPQRSTVWY" * 3 + "BZX*Uacdw-"))
    for trial in range(150):
        n_genomes, n_phams = int(rng.integers(2, 10)), int(rng.integers(1, 7))
        pool = ["".join(letters[rng.integers(0, letters.size, int(rng.integers(1, 91)))]) for _ in range(int(rng.integers(1, 8)))]
        genomes = []
        for g in range(n_genomes):
            genome = Genome(f"g{g:02d}")
            for p in rng.permutation(n_phams)[:int(rng.integers(1, n_phams + 1))]:
                for _ in range(int(rng.integers(1, 4)) if rng.random() < 0.3 else 1):
                    seq = pool[int(rng.integers(0, len(pool)))]
                    if rng.random() < 0.4:                                   # a point mutant or a truncation of a pool member
                        cut = int(rng.integers(0, len(seq)))
                        seq = seq[:cut] + "W" + seq[cut + 1:] if rng.random() < 0.5 else seq[:max(1, cut)]
                    genome.add(f"pham{p}", seq)
            genomes.append(genome)
        packed = pack_genomes(genomes)
        gpu_ctx.upload(packed)
        for metric in ALL_METRICS:
            for as_distance in (True, False):
                got = gpu_ctx.fill(metric, as_distance=as_distance)
                want = O.fill(packed, metric, as_distance=as_distance)
                assert np.array_equal(got, want), (trial, metric, as_distance)


def test_long_and_ragged_sequences(gpu_ctx, native_built):
    """Column sequences from 1 residue up to beyond the systolic kernels' reach."""
    from phamclust_amd.genome import Genome
    from phamclust_amd.pack import pack_genomes
    O = _oracle()
    rng = np.random.default_rng(21)
    aa = np.array(list("ACDEFGHIKLMNPQRSTVWY"))
    lens = [1, 2, 3, 15, 16, 17, 63, 64, 65, 255, 256, 257, 300, 513, 1279, 1280, 1281, 1535, 1536, 1537, 2100, 4096, 4097, 4500]
    gs = []
    for gi in range(3):
        g = Genome(f"g{gi}")
        for k, ln in enumerate(lens):
            base = "".join(aa[rng.integers(0, 20, ln)])
            g.add(f"p{k:02d}", base)
            if gi == 1 and ln > 3:                 # a paralog with an indel, so anchors have choices
                g.add(f"p{k:02d}", base[: ln // 2] + base[ln // 2 + 2:])
        gs.append(g)
    pk = pack_genomes(gs)
    gpu_ctx.upload(pk)
    for metric in ("aai", "peq"):
        assert np.array_equal(gpu_ctx.fill(metric), O.fill(pk, metric))
    g = pk.n_genes
    a, b = np.meshgrid(np.arange(g, dtype=np.int32), np.arange(g, dtype=np.int32))
    a, b = a.ravel(), b.ravel()
    keep = rng.random(a.shape[0]) < 0.07
    a, b = a[keep], b[keep]
    ident, diag = gpu_ctx.align_pairs(a, b)
    _, wi, wd = O.nw_batch(pk.residues, pk.seq_off, a, b)
    assert np.array_equal(ident, wi) and np.array_equal(diag, wd)


def _mutated(rng, seq, alphabet, sub=0.1, indel=0.01):
    out = []
    for ch in seq:
        u = rng.random()
        if u < indel:
            continue                                          # deletion
        out.append(alphabet[rng.integers(0, len(alphabet))] if u < indel + sub else ch)
        if rng.random() < indel:
            out.append(alphabet[rng.integers(0, len(alphabet))])   # insertion
    return "".join(out)


def test_strip_mined_long_column_genes(gpu_ctx, native_built):
    """Column genes beyond 4,096 residues run strip-mined on the systolic kernel (k_nw_strip: passes of 64 x W columns, the
    last column's (Ho, E) of every row step handed to the next pass through an HBM line) -- the reference's aligner has no length
    cliff (metrics.py:160-175).  Homolog pairs of 4,097 ... 20,000 residues, both sides long, long columns against short rows and
    the reverse, a tie-heavy 3-letter alphabet among them, under tie rules 0 and 3 (the cyclic one carries two differently tagged
    copies of Ho across the pass boundary too): (n_ident, n_diag) equal the oracle's through the chooser and through each wide
    variant forced; aai / peq fills over genomes holding such genes, and percent-positives (profile cell, passes of 1,536
    columns, column genes up to 8,191 residues) likewise.  Both forms of the kernel: one row per wave, and the passes of one
    alignment pipelined over the waves of a workgroup (a wave reads the boundary line the wave before it is still writing)."""
    from phamclust_amd.genome import Genome
    from phamclust_amd.pack import pack_genomes
    O = _oracle()
    rng = np.random.default_rng(404)
    aa = np.array(list("ACDEFGHIKLMNPQRSTVWY"))
    ags = np.array(list("AGS"))
    g, h = Genome("g0"), Genome("g1")
    lens = [150, 1200, 3000, 4097, 5000, 6200, 9000, 20000]
    for k, ln in enumerate(lens):
        alpha = ags if ln in (4097, 6200) else aa
        base = "".join(alpha[rng.integers(0, len(alpha), ln)])
        g.add(f"p{k}", base)
        h.add(f"p{k}", _mutated(rng, base, alpha))
    g.add("p4", "".join(aa[rng.integers(0, 20, 5100)]))       # a paralog of the 5,000-residue gene: the anchor has a choice
    pk = pack_genomes([g, h])
    lens_all = np.diff(pk.seq_off)
    G = pk.n_genes
    a, b = np.meshgrid(np.arange(G, dtype=np.int32), np.arange(G, dtype=np.int32))
    a, b = a.ravel(), b.ravel()
    keep = (lens_all[b] > 4096) & ((lens_all[a] < 4000) | (np.abs(lens_all[a].astype(np.int64) - lens_all[b]) < 1500) | (rng.random(a.shape[0]) < 0.15))
    a, b = a[keep], b[keep]
    assert a.size >= 40 and (lens_all[a] > 9000).any()
    gp, hp = Genome("q0"), Genome("q1")
    for k, ln in enumerate([700, 1600, 3000, 5000, 8100]):
        base = "".join(aa[rng.integers(0, 20, ln)])
        gp.add(f"p{k}", base)
        hp.add(f"p{k}", _mutated(rng, base, aa))
    pk_ppos = pack_genomes([gp, hp])
    a_pp, b_pp = np.meshgrid(np.arange(10, dtype=np.int32), np.arange(10, dtype=np.int32))
    a_pp, b_pp = a_pp.ravel(), b_pp.ravel()
    try:
        for rule in (0, 3):
            gpu_ctx.set_tie_rule(rule)
            O.set_tie_rule(rule)
            _, wi, wd = O.nw_batch(pk.residues, pk.seq_off, a, b)
            gpu_ctx.upload(pk)
            # PC_PIPE (read per launch): "0" one row per wave (the wide variants forced too); unset: the launcher's choice -- these few
            # tasks run pipelined, the passes of each alignment dealt over eight waves; 4, 3 (passes not a multiple of the waves), 1
            want_fill = {metric: O.fill(pk, metric) for metric in ("peq",)}      # (peq = round(af x aai): the same alignments and reduce as aai; one oracle pass over 20,000 x 20,000 cells less)
            for pipe in ("0", None, "4", "3", "1"):
                if pipe is None: os.environ.pop("PC_PIPE", None)
                else: os.environ["PC_PIPE"] = pipe
                for variant in ((0, 32, 48, 64) if pipe == "0" else (0, 48)):
                    ident, diag = gpu_ctx.align_pairs(a, b, variant=variant)
                    assert np.array_equal(ident, wi) and np.array_equal(diag, wd), f"rule {rule} variant {variant} PC_PIPE {pipe}"
                for metric in want_fill:
                    assert np.array_equal(gpu_ctx.fill(metric), want_fill[metric]), f"rule {rule} {metric} PC_PIPE {pipe}"
            os.environ.pop("PC_PIPE", None)
            if rule == 0:
                sel = (lens_all[b] < 4200) & (lens_all[a] < 5500)          # the one-lane-per-alignment kernel agrees (it stays as the fallback;
                ident, diag = gpu_ctx.align_pairs(a[sel], b[sel], variant=-1)   # 2 x 10^7 cells on ONE lane take seconds: only the 4,097-column pairs)
                assert sel.sum() >= 4 and np.array_equal(ident, wi[sel]) and np.array_equal(diag, wd[sel])
            # percent-positives: the profile cell, passes of 64 x 24 columns, for column genes of 1,537 ... 8,191 residues
            gpu_ctx.upload(pk_ppos)
            assert np.array_equal(gpu_ctx.fill("aai_ppos"), O.fill(pk_ppos, "aai_ppos")), f"rule {rule} aai_ppos"
            ident, diag = gpu_ctx.align_pairs(a_pp, b_pp)
            _, wi, wd = O.nw_batch(pk_ppos.residues, pk_ppos.seq_off, a_pp, b_pp)
            assert np.array_equal(ident, wi) and np.array_equal(diag, wd)
    finally:
        os.environ.pop("PC_PIPE", None)
        O.set_tie_rule(0)
        gpu_ctx.set_tie_rule(0)


def test_shard_and_assemble_single_gpu(gpu_ctx, native_built):
    """Shard-local fills of every rank of a 3-way and a 4-way split, gathered by hand and
    assembled on the device, equal the unsharded condensed result."""
    import torch
    from phamclust_amd.synth import synth_packed
    packed = synth_packed(83, 500, seed=2)
    gpu_ctx.upload(packed)
    for metric in ("jc", "af", "peq"):
        want = gpu_ctx.fill(metric)
        for world in (3, 4):
            strides, parts = set(), []
            for rank in range(world):
                gpu_ctx.set_shard(rank, world)
                stride = gpu_ctx.shard_stride()
                strides.add(stride)
                buf = torch.full((stride,), -1.0, dtype=torch.float64, device="cuda:0")
                gpu_ctx.fill_shard_dev(metric, True, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                parts.append(buf)
            assert len(strides) == 1
            gathered = torch.cat(parts)
            out = torch.empty(packed.n_pairs, dtype=torch.float64, device="cuda:0")
            gpu_ctx.assemble_dev(gathered.data_ptr(), world, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), want)
        gpu_ctx.set_shard(0, 1)


def test_aai_percent_positives(gpu_ctx, native_built, small_genomes, small_packed):
    """ppos=True (metrics.py:218-220): '+' columns count too.  Pairwise callable, bulk fill and both oracle
    restatements agree."""
    from phamclust_amd.metrics import average_aminoacid_identity
    O = _oracle()
    got = gpu_ctx.upload(small_packed).fill("aai_ppos", as_distance=False)
    plain = gpu_ctx.fill("aai", as_distance=False)
    n = small_packed.n_genomes
    want_c = np.array([O.pair(small_packed, "aai_ppos", s, t, as_distance=False) for s in range(n) for t in range(s + 1, n)])
    assert np.array_equal(got, want_c)
    assert (got >= plain).all() and (got > plain).any()
    for s, t in ((0, 5), (3, 17), (16, 18), (20, 22)):
        py = O.py_aai(small_genomes[s], small_genomes[t], ppos=True)
        assert py == O.pair(small_packed, "aai_ppos", s, t, as_distance=False)
        assert average_aminoacid_identity(small_genomes[s], small_genomes[t], ppos=True) == py
        assert average_aminoacid_identity(small_genomes[s], small_genomes[t], True, True) == O.py_aai(small_genomes[s], small_genomes[t], True, True)


def test_percent_positives_on_the_systolic_kernel(gpu_ctx, native_built):
    """aai with ppos=True at some size: the launch classes whose profile cell can run (segments of up to 32 lanes, W <= 24)
    take "identical or positive" from the 16-bit increment table, the others fall back to the general kernel; column
    genes with bytes outside the alphabet included ('*' against '*' scores +1, so the table is exact for them too)."""
    from phamclust_amd.genome import Genome
    from phamclust_amd.pack import pack_genomes
    from phamclust_amd.synth import synth_packed
    O = _oracle()
    pk = synth_packed(150, 900, seed=21)
    gpu_ctx.upload(pk)
    got, st = gpu_ctx.fill("aai_ppos", True, want_stats=True)
    assert np.array_equal(got, O.fill(pk, "aai_ppos", True))
    assert st["n_align_launches"] > 3
    rng = np.random.default_rng(8)
    aa = np.array(list("ACDEFGHIKLMNPQRSTVWY" * 3 + "BZX*UJ"))
    genomes = []
    for gi in range(8):
        g = Genome(f"g{gi}")
        for p in range(5):
            g.add(f"pham{p}", "".join(aa[rng.integers(0, aa.size, int(rng.integers(3, 1500)))]))
        genomes.append(g)
    pk2 = pack_genomes(genomes)
    gpu_ctx.upload(pk2)
    assert np.array_equal(gpu_ctx.fill("aai_ppos", False), O.fill(pk2, "aai_ppos", False))


def test_matrix_de_novo_drop_in(native_built, small_genomes):
    """The reference-shaped entry point: matrix_de_novo(genomes, METRICS[m], cpus)."""
    from phamclust_amd.cli import METRICS
    from phamclust_amd.matrix import matrix_de_novo
    for metric in ("gcs", "peq"):
        names, gold, diag = read_lower_triangle(golden_file(metric))
        m = matrix_de_novo(small_genomes, METRICS[metric], 4)
        assert m.is_distance and m.nodes == names
        assert np.array_equal(m.to_ndarray(condensed=True), gold)
        assert all(m.get_weight(n, n) == 0.0 for n in names)
    # pairwise call of the same callable agrees with the matrix cell
    s, t = small_genomes[3], small_genomes[17]
    assert METRICS["peq"](s, t, as_distance=True) == m.get_weight(s.name, t.name)


def test_redundant_sequences_are_aligned_once(gpu_ctx, native_built):
    """Genomes that share identical proteins (clones, clones with a few genes changed, paralogs with the same
    sequence): the plan aligns each distinct (row sequence, column sequence) pair once and aliases the rest, and
    every value still equals the oracle, which aligns them all (metrics.py:211-217)."""
    from phamclust_amd.genome import Genome
    from phamclust_amd.pack import pack_genomes
    from phamclust_amd.synth import synth_genomes
    O = _oracle()
    base = synth_genomes(24, 300, seed=11)
    genomes = list(base)
    for i, g in enumerate(base[:12]):                       # exact clones and near-clones
        clone, near = Genome(f"clone_{i:02d}"), Genome(f"near_{i:02d}")
        for j, (pham, translations) in enumerate(g):
            for t in translations:
                clone.add(pham, t)
                near.add(pham, t if j % 5 else t[:-3] + "WWW")
            if j % 7 == 0:
                near.add(pham, translations[0])             # a paralog with an identical sequence
        genomes += [clone, near]
    genomes.sort(key=lambda g: g.name)
    packed = pack_genomes(genomes)
    for metric in ("aai", "peq"):
        got, st = gpu_ctx.upload(packed).fill(metric, as_distance=True, want_stats=True)
        assert np.array_equal(got, O.fill(packed, metric, as_distance=True))
        assert 0 < st["n_distinct_alignments"] < 0.7 * st["n_alignments"] and st["n_distinct_cells"] < st["n_cells"]
    clones = []                                             # 40 copies of one genome: one distinct alignment per gene
    for i in range(40):
        g = Genome(f"same_{i:02d}")
        for pham, translations in base[0]:
            for t in translations:
                g.add(pham, t)
        clones.append(g)
    packed = pack_genomes(clones)
    got, st = gpu_ctx.upload(packed).fill("peq", as_distance=False, want_stats=True)
    assert np.array_equal(got, O.fill(packed, "peq", as_distance=False)) and (got == 1.0).all()
    assert st["n_alignments"] >= 780 * st["n_distinct_alignments"] > 0
    got, st = gpu_ctx.upload(pack_genomes(base)).fill("peq", want_stats=True)      # the synthetic set itself: next to nothing aliased
    assert 0.99 * st["n_alignments"] < st["n_distinct_alignments"] <= st["n_alignments"] and st["n_distinct_cells"] <= st["n_cells"]


@pytest.mark.parametrize("metric", SET_METRICS)
def test_full_size_set_metrics_vs_oracle(gpu_ctx, native_built, metric):
    """BASELINE configs[1] at full size (synth(2000,5000): 1,999,000 pairs): every value against the oracle."""
    from phamclust_amd.synth import synth_packed
    O = _oracle()
    packed = synth_packed(2000, 5000)
    got = gpu_ctx.upload(packed).fill(metric, as_distance=True)
    want = O.fill(packed, metric, as_distance=True)
    assert got.shape == (1999000,) and np.array_equal(got, want)


def test_full_size_peq_properties(gpu_ctx, native_built):
    """The headline configuration at full size (synth(5000,5000) -m peq, 12,497,500 pairs, 33.5 M alignments), through
    properties that need no full CPU run: the fill is deterministic (bucket order comes from atomics); an 8-way
    shard assembles to the unsharded matrix; peq == round(af * aai, 6) over three independent fills
    (metrics.py:247-253); a value is zero exactly where no pham is shared; 20,000 random pairs equal the oracle."""
    import torch
    from phamclust_amd.synth import synth_packed
    O = _oracle()
    packed = synth_packed(5000, 5000)
    gpu_ctx.upload(packed)
    n = packed.n_genomes
    peq = gpu_ctx.fill("peq", as_distance=False)
    assert peq.shape == (n * (n - 1) // 2,) and peq.min() >= 0.0 and peq.max() <= 1.0
    assert np.array_equal(gpu_ctx.fill("peq", as_distance=False), peq)
    af, aai, jc = (gpu_ctx.fill(m, as_distance=False) for m in ("af", "aai", "jc"))
    product = af * aai                                   # both factors are the rounded similarities (metrics.py:247-252)
    py_round = _py_round6                                                     # CPython's round: the reference's semantics
    assert np.array_equal(peq, py_round(product))
    assert not peq[jc == 0.0].any() and not aai[jc == 0.0].any() and not af[jc == 0.0].any()
    dist = gpu_ctx.fill("peq", as_distance=True)
    assert np.array_equal(dist, py_round(1.0 - product))                      # metrics.py:250-253
    # 8-way shard, gathered by hand on the one GPU
    stream = torch.cuda.current_stream().cuda_stream
    parts = []
    for rank in range(8):
        gpu_ctx.set_shard(rank, 8)
        buf = torch.empty(gpu_ctx.shard_stride(), dtype=torch.float64, device="cuda:0")
        gpu_ctx.fill_shard_dev("peq", True, buf.data_ptr(), stream)
        parts.append(buf)
    out = torch.empty(packed.n_pairs, dtype=torch.float64, device="cuda:0")
    gathered = torch.cat(parts)                          # queued on torch's stream; the assembly below runs on the same one
    gpu_ctx.assemble_dev(gathered.data_ptr(), 8, out.data_ptr(), stream)
    torch.cuda.synchronize()
    gpu_ctx.set_shard(0, 1)
    assert np.array_equal(out.cpu().numpy(), dist)
    rng = np.random.default_rng(5)
    a, b = rng.integers(0, n, 20000), rng.integers(0, n, 20000)
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    keep = lo < hi
    lo, hi = lo[keep], hi[keep]
    idx = lo * n - lo * (lo + 1) // 2 + (hi - lo - 1)
    assert np.array_equal(dist[idx], O.pairs(packed, "peq", lo, hi, as_distance=True))


def test_config2_peq_2000_full_fill(gpu_ctx, native_built):
    """BASELINE configs[2]: synth(2000,5000) -m peq on one GPU (1,999,000 pairs, ~6.5 M alignments, ~3.9e11 DP cells).
    The whole matrix: determinism, range, peq == round(af * aai, 6) against three independent fills, zero exactly
    where nothing is shared, a 3-way cost-balanced shard assembling to the same matrix, and 24,000 random pairs
    (aai and peq, distance and similarity) equal to the oracle."""
    import torch
    from phamclust_amd.synth import synth_packed
    O = _oracle()
    packed = synth_packed(2000, 5000)
    gpu_ctx.upload(packed)
    n = packed.n_genomes
    peq, st = gpu_ctx.fill("peq", as_distance=False, want_stats=True)
    assert peq.shape == (1999000,) and st["n_pairs"] == 1999000
    assert 6.0e6 < st["n_alignments"] < 7.0e6 and 3.5e11 < st["n_cells"] < 4.3e11        # 6.52 M, 3.90e11
    assert np.array_equal(gpu_ctx.fill("peq", as_distance=False), peq)
    assert peq.min() >= 0.0 and peq.max() <= 1.0
    af, aai, jc = (gpu_ctx.fill(m, as_distance=False) for m in ("af", "aai", "jc"))
    py_round = _py_round6
    assert np.array_equal(peq, py_round(af * aai))
    assert not peq[jc == 0.0].any() and (aai[jc > 0.0] > 0.0).all()
    dist = gpu_ctx.fill("peq", as_distance=True)
    assert np.array_equal(dist, py_round(1.0 - af * aai))
    stream = torch.cuda.current_stream().cuda_stream
    parts = []
    for rank in range(3):
        gpu_ctx.set_shard(rank, 3, balanced=True)
        buf = torch.empty(gpu_ctx.shard_stride(), dtype=torch.float64, device="cuda:0")
        gpu_ctx.fill_shard_dev("peq", True, buf.data_ptr(), stream)
        parts.append(buf)
    out = torch.empty(packed.n_pairs, dtype=torch.float64, device="cuda:0")
    gathered = torch.cat(parts)
    gpu_ctx.assemble_dev(gathered.data_ptr(), 3, out.data_ptr(), stream)
    torch.cuda.synchronize()
    gpu_ctx.set_shard(0, 1)
    assert np.array_equal(out.cpu().numpy(), dist)
    rng = np.random.default_rng(2000)
    a, b = rng.integers(0, n, 25000), rng.integers(0, n, 25000)
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    keep = lo < hi
    lo, hi = lo[keep][:24000], hi[keep][:24000]
    assert lo.size == 24000
    idx = lo * n - lo * (lo + 1) // 2 + (hi - lo - 1)
    assert np.array_equal(dist[idx], O.pairs(packed, "peq", lo, hi, as_distance=True))
    assert np.array_equal(peq[idx], O.pairs(packed, "peq", lo, hi, as_distance=False))
    assert np.array_equal(aai[idx], O.pairs(packed, "aai", lo, hi, as_distance=False))


def test_balanced_deal_equals_host_mirror(gpu_ctx, native_built):
    """The device-side cost-balanced deal (pc_set_shard_balanced) and its host mirror (distributed.balanced_deal, what the
    CPU tests of the N>1 path use) are the same partition, and pc_shard_table describes the closed-form deal too."""
    from phamclust_amd import distributed as D
    from phamclust_amd.synth import synth_packed
    packed = synth_packed(173, 900, seed=31)
    gpu_ctx.upload(packed)
    for world in (2, 3, 8):
        gpu_ctx.set_shard(0, world, balanced=True)
        costs = gpu_ctx.target_costs()
        t_rank, t_lbase = gpu_ctx.shard_table()
        m_rank, m_lbase, m_stride = D.balanced_deal(costs, world)
        assert np.array_equal(t_rank, m_rank) and np.array_equal(t_lbase, m_lbase) and gpu_ctx.shard_stride() == m_stride
        gpu_ctx.set_shard(1, world, balanced=False)
        t_rank, t_lbase = gpu_ctx.shard_table()
        for r in range(world):
            owned, lbase = D.shard_layout(packed.n_genomes, r, world)
            assert (t_rank[owned] == r).all() and np.array_equal(t_lbase[owned], lbase[:-1])
    gpu_ctx.set_shard(0, 1)


def test_balanced_deal(gpu_ctx, native_built):
    """pc_set_shard_balanced: every target genome is owned once, the shards assemble to the unsharded matrix for
    a set metric and an alignment metric, and the alignment work per rank is level (genomes of very different size)."""
    import torch
    from phamclust_amd.genome import Genome
    from phamclust_amd.pack import pack_genomes
    from phamclust_amd.synth import synth_genomes
    base = synth_genomes(90, 400, seed=21)
    genomes = []
    for i, g in enumerate(base):                            # every third genome keeps only a fifth of its genes
        h = Genome(g.name)
        for j, (pham, translations) in enumerate(g):
            if i % 3 or j % 5 == 0:
                for t in translations:
                    h.add(pham, t)
        genomes.append(h)
    packed = pack_genomes(genomes)
    gpu_ctx.upload(packed)
    stream = torch.cuda.current_stream().cuda_stream
    for metric in ("jc", "peq"):
        gpu_ctx.set_shard(0, 1)
        want = gpu_ctx.fill(metric)
        for world in (2, 5):
            parts, pairs, cells = [], 0, []
            for rank in range(world):
                gpu_ctx.set_shard(rank, world, balanced=True)
                buf = torch.full((gpu_ctx.shard_stride(),), -1.0, dtype=torch.float64, device="cuda:0")
                st = gpu_ctx.fill_shard_dev(metric, True, buf.data_ptr(), stream)
                parts.append(buf); pairs += gpu_ctx.shard_pairs(); cells.append(st["n_cells"])
            assert pairs == packed.n_pairs
            gathered = torch.cat(parts)
            out = torch.empty(packed.n_pairs, dtype=torch.float64, device="cuda:0")
            gpu_ctx.assemble_dev(gathered.data_ptr(), world, out.data_ptr(), stream)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), want)
            if metric == "peq":
                assert max(cells) <= 1.03 * (sum(cells) / world)
    gpu_ctx.set_shard(0, 1)


@pytest.mark.parametrize("balanced", [False, True])
def test_more_ranks_than_genomes(gpu_ctx, native_built, balanced):
    """3 genomes dealt to 8 ranks: most ranks own nothing, their fills are no-ops, the assembly is still complete."""
    import torch
    from phamclust_amd.synth import synth_packed
    packed = synth_packed(3, 60, seed=4)
    gpu_ctx.upload(packed)
    stream = torch.cuda.current_stream().cuda_stream
    for metric in ("gcs", "pocp", "peq"):
        gpu_ctx.set_shard(0, 1)
        want = gpu_ctx.fill(metric)
        parts, pairs = [], 0
        for rank in range(8):
            gpu_ctx.set_shard(rank, 8, balanced=balanced)
            buf = torch.full((max(gpu_ctx.shard_stride(), 1),), -1.0, dtype=torch.float64, device="cuda:0")
            gpu_ctx.fill_shard_dev(metric, True, buf.data_ptr(), stream)
            parts.append(buf[:gpu_ctx.shard_stride()]); pairs += gpu_ctx.shard_pairs()
        assert pairs == 3
        gathered = torch.cat(parts)
        out = torch.empty(3, dtype=torch.float64, device="cuda:0")
        gpu_ctx.assemble_dev(gathered.data_ptr(), 8, out.data_ptr(), stream)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want)
    gpu_ctx.set_shard(0, 1)


def test_unsynchronised_fill_survives_reupload(gpu_ctx, native_built):
    """a fill called without stats returns while its kernels still run on the CALLER's stream and still use the
    context's work buffers; an upload (or re-shard, or test hook) issued right behind it must wait for them instead of
    overwriting those buffers.  Fill A asynchronously on a side stream, upload B at once, fill B: both results exact."""
    import torch
    from phamclust_amd.synth import synth_packed
    O = _oracle()
    a_pk, b_pk = synth_packed(170, 900, seed=41), synth_packed(90, 400, seed=42)
    want_a, want_b = O.fill(a_pk, "peq", as_distance=True), O.fill(b_pk, "peq", as_distance=True)
    side = torch.cuda.Stream()
    for _ in range(3):
        gpu_ctx.upload(a_pk)
        out_a = torch.full((a_pk.n_pairs,), -1.0, dtype=torch.float64, device="cuda:0")
        gpu_ctx.fill_dev("peq", True, out_a.data_ptr(), side.cuda_stream, want_stats=False)      # returns with kernels in flight
        gpu_ctx.upload(b_pk)                                                                     # must not disturb them
        got_b = gpu_ctx.fill("peq", as_distance=True)
        side.synchronize()
        assert np.array_equal(out_a.cpu().numpy(), want_a)
        assert np.array_equal(got_b, want_b)
    assert torch.cuda.current_device() == 0


def test_fill_borrow_is_the_same_matrix(gpu_ctx, native_built):
    """pc_fill_borrow: the context's pinned buffer holds exactly what pc_fill returns, is read-only for Python, and is
    re-used (not re-pinned) by the next call."""
    from phamclust_amd.synth import synth_packed
    packed = synth_packed(300, 900, seed=43)
    gpu_ctx.upload(packed)
    for metric in ("jc", "af", "peq"):
        owned = gpu_ctx.fill(metric, as_distance=True)
        lent, st = gpu_ctx.fill(metric, as_distance=True, want_stats=True, borrow=True)
        assert lent.shape == owned.shape and np.array_equal(lent, owned) and st["n_pairs"] == packed.n_pairs
        assert not lent.flags.writeable
        with pytest.raises(ValueError):
            lent[0] = 0.5
    first = gpu_ctx.fill("gcs", borrow=True).ctypes.data
    assert gpu_ctx.fill("jc", borrow=True).ctypes.data == first


def test_fill_distributed_single_rank(gpu_ctx, native_built):
    """The product's multi-GPU entry point with a 1-rank group: shard -> (no gather) -> device assembly."""
    import torch
    import torch.distributed as dist
    from phamclust_amd.distributed import fill_distributed
    from phamclust_amd.synth import synth_packed
    packed = synth_packed(70, 400, seed=8)
    gpu_ctx.upload(packed)
    want = gpu_ctx.fill("peq")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1)
    try:
        out, stats = fill_distributed(gpu_ctx, "peq", True)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want) and stats["n_pairs"] == packed.n_pairs
    finally:
        if created:
            dist.destroy_process_group()
        gpu_ctx.set_shard(0, 1)


def test_bench_json_contract(native_built):
    """bench.py prints ONE JSON line with the contract's keys (small workload, as a subprocess)."""
    import json
    import os
    import subprocess
    import sys
    from conftest import REPO
    proc = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--genomes", "300", "--steps", "1", "--warmup", "1",
                           "--cpu-seconds", "1", "--verify-pairs", "2000"], capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["config"]["workload"].startswith("synth(300,5000) -m peq") and d["data"] == "synthetic"
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"]) and d["cpu_baseline"]["kind"] == "port"
    assert d["verified"]["bit_exact"] is True and d["value"] > 0
    assert d["value_resident"] == d["value"] and 0 < d["value_wall"] <= d["value"] and "SURVEY 8(d)" in d["value_definition"]


def test_bench_two_ranks_rehearsal(native_built):
    """bench.py under the driver's launcher with 2 ranks.  This box has one GPU, so the ranks share it and the
    gather goes through gloo (PC_BENCH_BACKEND=gloo): everything but the RCCL transport itself is the N>1 path --
    rendezvous, build behind a barrier, shard fills, gather, device assembly, max-over-ranks timing, one JSON line
    whose sampled pairs equal the oracle."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from conftest import REPO
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PC_BENCH_BACKEND="gloo")
    proc = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--genomes", "301", "--steps", "2",
                           "--warmup", "1", "--verify-pairs", "3000"], capture_output=True, text=True, timeout=900, env=env)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["genome_pairs"] == 301 * 300 // 2 and "REHEARSAL" in d["config"]["parallelism"]
    assert d["verified"]["bit_exact"] is True and d["roofline"]["n_alignments"] > 0
    # a multi-GPU line explains itself: per-rank stages (max over ranks), the exchange and the root-only assembly, the shards
    assert d["config"]["dist_mode"] == "pairs"
    for key in ("plan_max_over_ranks", "align_max_over_ranks", "reduce_max_over_ranks", "exchange_rank0", "assemble_rank0", "exchange_bytes"):
        assert key in d["stage_ms"], key
    assert d["stage_ms"]["align_max_over_ranks"] > 0 and d["stage_ms"]["exchange_rank0"] > 0 and d["stage_ms"]["assemble_rank0"] > 0
    # the fixed cost of being two ranks travels with the line: start-up seconds and the rate one matrix gets with them
    assert d["init_s"] > 0.5 and d["upload_s"] > 0 and 0 < d["value_wall_incl_init"] < d["value"]
    lo, hi = d["shards"]["pairs_min_max"]
    assert 0 < lo <= hi and lo + hi == 301 * 300 // 2


@pytest.mark.parametrize("world", [1, 3, 8])
@pytest.mark.parametrize("redundant", [False, True])
def test_alignment_sliced_route_single_gpu(gpu_ctx, native_built, world, redundant):
    """The alignment-sliced multi-GPU route, all its ranks run back to back on one GPU: plan, every rank's slice into its
    own result array, the arrays summed as 64-bit integers (the reduce), the root's reduce stage == the unsharded fill.
    Slices are disjoint, together they hold every distinct alignment, and their DP work is balanced.  `redundant`:
    many byte-identical proteins, so slots alias distinct alignments across what would be different ranks' pairs."""
    import dataclasses
    import torch
    from phamclust_amd.synth import synth_packed
    pk = synth_packed(160, 900, seed=13)
    if redundant:                                  # overwrite 60 % of the genes with their (cluster, pham) group's first sequence
        G = pk.n_genes
        genome_of = np.repeat(np.arange(pk.n_genomes, dtype=np.int64), np.diff(pk.gene_off))
        group = (genome_of // 40) * (pk.n_phams + 1) + pk.gene_pham
        ug, first = np.unique(group, return_index=True)
        canon = first[np.searchsorted(ug, group)]
        src = np.where(np.random.default_rng(7).random(G) < 0.6, canon, np.arange(G))
        lens = np.diff(pk.seq_off)[src]
        seq_off = np.zeros(G + 1, dtype=np.int64); np.cumsum(lens, out=seq_off[1:])
        idx = np.repeat(pk.seq_off[:-1][src] - seq_off[:-1], lens) + np.arange(seq_off[-1])
        tlen = np.zeros(pk.n_genomes, dtype=np.int64); np.add.at(tlen, genome_of, lens)
        pk = dataclasses.replace(pk, seq_off=seq_off, residues=np.ascontiguousarray(pk.residues[idx]), tlen=tlen).validate()
    gpu_ctx.upload(pk)
    stream = torch.cuda.current_stream().cuda_stream
    for metric in ("peq", "aai"):
        want = torch.empty(pk.n_pairs, dtype=torch.float64, device="cuda")
        st_full = gpu_ctx.fill_dev(metric, True, want.data_ptr(), stream)
        plan = gpu_ctx.plan_dev(metric, stream)
        n = plan["n_distinct_alignments"]
        assert n == st_full["n_distinct_alignments"] and plan["n_alignments"] == st_full["n_alignments"]
        if redundant:
            assert n < 0.7 * plan["n_alignments"]
        parts = []
        for r in range(world):
            res = torch.full((max(n, 1),), -1, dtype=torch.int64, device="cuda")
            gpu_ctx.align_slice_dev(r, world, res.data_ptr(), stream)
            parts.append(res)
        torch.cuda.synchronize()
        stack = torch.stack(parts)[:, :n]
        assert int((stack != 0).sum(dim=0).max()) <= 1                      # no alignment computed by two ranks ...
        total = stack.sum(dim=0)
        assert int((total != 0).sum()) == n                                 # ... and every one by some rank (aln_len >= 1)
        if world > 1:
            per_rank = (stack != 0).sum(dim=1).double()
            assert float(per_rank.max() / per_rank.mean()) < 1.25
        out = torch.empty(pk.n_pairs, dtype=torch.float64, device="cuda")
        full = torch.zeros(max(n, 1), dtype=torch.int64, device="cuda"); full[:n] = total
        gpu_ctx.reduce_dev(metric, True, full.data_ptr(), out.data_ptr(), stream)
        torch.cuda.synchronize()
        assert torch.equal(out, want), metric
    gpu_ctx.set_shard(1, 2)                                                    # a plan belongs to the shard it was made for
    with pytest.raises(Exception):
        gpu_ctx.align_slice_dev(0, 1, parts[0].data_ptr(), stream)
    gpu_ctx.set_shard(0, 1)


@pytest.mark.parametrize("mode", ["alignments"])
def test_bench_two_ranks_rehearsal_alignment_slices(native_built, mode):
    """bench.py with 2 ranks sharing this box's GPU (gloo transport) on the alignment-sliced route
    (PHAMCLUST_DIST_MODE=alignments): plan on both ranks, a slice each, one reduce of the results, matrix on rank 0."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from conftest import REPO
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PC_BENCH_BACKEND="gloo", PHAMCLUST_DIST_MODE=mode)
    proc = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--genomes", "301", "--steps", "2",
                           "--warmup", "1", "--verify-pairs", "3000"], capture_output=True, text=True, timeout=900, env=env)
    assert proc.returncode == 0, proc.stderr[-3000:]
    d = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and "alignments sliced" in d["config"]["parallelism"]
    assert d["verified"]["bit_exact"] is True and d["roofline"]["n_alignments"] > 0
    assert d["roofline"]["n_distinct_alignments"] <= d["roofline"]["n_alignments"]
    assert d["config"]["dist_mode"] == "alignments" and "reduce" in d["stage_ms"]["exchange"]
    assert d["stage_ms"]["exchange_rank0"] > 0 and d["stage_ms"]["assemble_rank0"] > 0 and d["stage_ms"]["plan_max_over_ranks"] > 0


@pytest.mark.parametrize("mode", ["pairs", "alignments"])
def test_bench_two_gpus_rccl(native_built, mode):
    """bench.py --gpus 2 over RCCL (backend nccl, one GPU per rank), both ways of splitting an aai / peq fill: the
    line is bit-exact against the oracle and carries the stage breakdown.  Needs a box with at least two GPUs."""
    import json
    import os
    import socket
    import subprocess
    import sys
    import torch
    from conftest import REPO
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU here: the RCCL transport needs two")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PHAMCLUST_DIST_MODE=mode, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("PC_BENCH_BACKEND", None)
    proc = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--genomes", "1000", "--steps", "2",
                           "--warmup", "1", "--verify-pairs", "5000"], capture_output=True, text=True, timeout=900, env=env)
    assert proc.returncode == 0, proc.stderr[-3000:]
    d = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["verified"]["bit_exact"] is True and d["config"]["dist_mode"] == mode
    assert "REHEARSAL" not in d["config"]["parallelism"] and d["stage_ms"]["exchange_rank0"] > 0


def test_rccl_branches_with_one_rank(native_built):
    """The `nccl` (= RCCL) branches of the multi-GPU fill on a ONE-GPU box: a one-rank nccl process group and
    PHAMCLUST_DIST_FORCE_EXCHANGE=1, so that shard -> dist.gather into chunk views of one device buffer -> device assembly
    (pairs) and plan -> slice -> dist.reduce -> matrix on the root (alignments) run the very calls an 8-GPU job makes, on
    device tensors, through RCCL.  Every metric must equal the plain fill bit for bit.  (Two ranks cannot share a GPU
    under RCCL; the two-GPU test below needs a bigger box.)"""
    import os
    import socket
    import subprocess
    import sys
    from conftest import REPO
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    code = """
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from phamclust_amd import hip
from phamclust_amd.distributed import fill_distributed
from phamclust_amd.synth import synth_packed
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
packed = synth_packed(257, 3000, seed=5)
with hip.Context(0) as ctx:
    ctx.upload(packed)
    for metric in ("jc", "af", "aai", "peq"):
        want = np.array(ctx.fill(metric, True), copy=True)
        for mode in ("pairs", "alignments"):
            out, st = fill_distributed(ctx, metric, True, mode=mode)
            torch.cuda.synchronize()
            got = out.cpu().numpy()
            assert st["ms_exchange"] >= 0 and st["exchange_bytes"] > 0, st
            assert np.array_equal(got, want), (metric, mode)
            print(metric, mode, st["dist_mode"], "exchange ms", round(st["ms_exchange"], 3), flush=True)
        ctx.set_shard(0, 1)
dist.barrier()
dist.destroy_process_group()
print("ok")
""" % REPO
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               PHAMCLUST_DIST_FORCE_EXCHANGE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    proc = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert proc.returncode == 0 and proc.stdout.strip().endswith("ok"), (proc.stdout[-1500:], proc.stderr[-3000:])


@pytest.mark.parametrize("mode", ["pairs", "alignments"])
def test_bench_distributed_flow_with_one_rank_over_rccl(native_built, mode):
    """bench.py's N > 1 flow -- nccl process group, sharded fill, exchange, assembly, its all_reduce bookkeeping, the init_s /
    upload_s keys -- with ONE rank (PC_BENCH_FORCE_DIST=1): everything the driver's 8-GPU run executes except a second GPU."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from conftest import REPO
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PC_BENCH_FORCE_DIST="1", PHAMCLUST_DIST_MODE=mode, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("PC_BENCH_BACKEND", "RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    proc = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--genomes", "301", "--steps", "2", "--warmup", "1",
                           "--verify-pairs", "3000", "--cpu-seconds", "0"], capture_output=True, text=True, timeout=900, env=env)
    assert proc.returncode == 0, proc.stderr[-3000:]
    d = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["verified"]["bit_exact"] is True and d["config"]["dist_mode"] == mode
    assert "ONE rank" in d["config"]["parallelism"] and d["stage_ms"]["exchange_rank0"] > 0 and d["stage_ms"]["assemble_rank0"] > 0
    assert d["init_s"] > 0 and d["upload_s"] > 0 and d["value_wall_incl_init"] > 0
    assert "HIP events" in d["stage_ms"]["clock"]


def test_graft_entry_smoke(native_built):
    """The driver's smoke(): all six metrics on a small synthetic set against the oracle."""
    import importlib
    import sys
    from conftest import REPO
    sys.path.insert(0, REPO)
    importlib.import_module("__graft_entry__").smoke()


@pytest.mark.parametrize("dataset", ["small_input", "synth2000"])
def test_certified_alignments_equal_the_unique_statistics(gpu_ctx, native_built, small_packed, dataset):
    """Reference-independent pin of the aligner at real lengths (metrics.py:160-175, 216-217).  oracle/pc_cooptimal.c
    counts EVERY optimal global alignment of a sequence pair and the range of (n_ident, n_diag) over them, with a DP that
    shares nothing with the oracle's aligner and is itself pinned by brute force (tests/test_cooptimal.py).  Where the
    range is one point any correct Needleman-Wunsch -- parasail included -- must report it: the kernels must, on every
    such alignment, through the default variant chooser; everywhere else they must stay inside the range."""
    from phamclust_amd.synth import synth_packed
    from phamclust_amd import hip
    O = _oracle()
    if dataset == "small_input":
        packed = small_packed
        iu = np.triu_indices(packed.n_genomes, 1)
        s_idx, t_idx = iu[0], iu[1]
    else:
        packed = synth_packed(2000, 5000)
        rng = np.random.default_rng(20241218)                   # head of the sample behind tests/golden/unique_optimum.json
        s = rng.integers(0, packed.n_genomes, 100000)
        t = rng.integers(0, packed.n_genomes, 100000)
        keep = s != t
        s_idx, t_idx = np.minimum(s, t)[keep][:4000], np.maximum(s, t)[keep][:4000]
    a, b, _ = O.enumerate_alignments(packed, s_idx, t_idx)
    _, count, rng4 = O.cooptimal_batch(packed.residues, packed.seq_off, a, b)
    cert = (rng4[:, 0] == rng4[:, 1]) & (rng4[:, 2] == rng4[:, 3])
    gpu_ctx.upload(packed)
    ident, diag = gpu_ctx.align_pairs(a, b, variant=0)
    assert cert.sum() > 0.8 * a.shape[0] > 1000
    assert np.array_equal(ident[cert], rng4[cert, 0]) and np.array_equal(diag[cert], rng4[cert, 2])
    assert ((rng4[:, 0] <= ident) & (ident <= rng4[:, 1]) & (rng4[:, 2] <= diag) & (diag <= rng4[:, 3])).all()
    assert (count[~cert] > 1).all()
    # the certified set runs through many widths of the default chooser
    lens = (packed.seq_off[1:] - packed.seq_off[:-1])[b[cert]]
    widths = {hip.Context.variant_width(int(x)) for x in np.unique(lens)}
    assert len(widths) >= (8 if dataset == "synth2000" else 4), sorted(widths)


def test_chunked_fill_is_the_unchunked_matrix(gpu_ctx, native_built):
    """Memory-bounded batching (matrix.py:474-493: the reference never holds more than ~10,000 pairs per CPU in flight).
    With the plan budget forced tiny the synth(2000,5000) peq fill (BASELINE configs[2]) runs plan -> align -> reduce over
    >= 5 successive target ranges and reproduces the one-piece matrix bit for bit; so do a sharded fill and aai."""
    import torch
    from phamclust_amd.synth import synth_packed
    packed = synth_packed(2000, 5000)
    gpu_ctx.upload(packed)
    try:
        want, st1 = gpu_ctx.fill("peq", want_stats=True)
        assert st1["n_chunks"] == 1
        gpu_ctx.set_plan_budget(st1["n_alignments"] * 56 // 6)
        got, st = gpu_ctx.fill("peq", want_stats=True)
        assert st["n_chunks"] >= 5, st
        assert np.array_equal(got, want)
        assert st["n_alignments"] == st1["n_alignments"] and st["n_cells"] == st1["n_cells"]
        assert st["n_distinct_alignments"] >= st1["n_distinct_alignments"]      # duplicates merge inside a chunk only
        assert st["ms_align"] > 0 and st["ms_plan"] > 0
        # without stats (asynchronous w.r.t. the host except for the plan read-backs), into device memory
        out = torch.full((packed.n_pairs,), -1.0, dtype=torch.float64, device="cuda:0")
        gpu_ctx.fill_dev("peq", True, out.data_ptr(), torch.cuda.current_stream().cuda_stream, want_stats=False)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want)
        # a shard of a 3-rank job, chunked, against the same shard unchunked
        gpu_ctx.set_shard(1, 3, balanced=True)
        stride = gpu_ctx.shard_stride()
        a = torch.full((stride,), -1.0, dtype=torch.float64, device="cuda:0")
        b = torch.full((stride,), -2.0, dtype=torch.float64, device="cuda:0")
        sa = gpu_ctx.fill_shard_dev("peq", True, a.data_ptr(), torch.cuda.current_stream().cuda_stream)
        gpu_ctx.set_plan_budget(0)
        sb = gpu_ctx.fill_shard_dev("peq", True, b.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert sa["n_chunks"] >= 2 and sb["n_chunks"] == 1 and torch.equal(a, b)
        gpu_ctx.set_shard(0, 1)
        # a budget below one target genome's alignments: one target per chunk, still the same values
        small = synth_packed(120, 600, seed=9)
        gpu_ctx.upload(small)
        want_aai = gpu_ctx.fill("aai")
        gpu_ctx.set_plan_budget(56)
        got_aai, st = gpu_ctx.fill("aai", want_stats=True)
        assert st["n_chunks"] >= 100 and np.array_equal(got_aai, want_aai)
        # the alignment-sliced route keeps ONE whole plan: a chunk's plan is not accepted in its place
        with pytest.raises(Exception):
            gpu_ctx.align_slice_dev(0, 1, a.data_ptr(), torch.cuda.current_stream().cuda_stream)
    finally:
        gpu_ctx.set_plan_budget(0)
        gpu_ctx.set_shard(0, 1)


def test_out_of_memory_plan_becomes_smaller_chunks(native_built):
    """A device allocation that fails inside a fill is answered with smaller chunks, not with PC_ERR_HIP.  The failure is
    injected (PC_FAKE_OOM_ABOVE: allocations above that many bytes fail) -- a switch that exists only in the library's
    -DPC_TEST_HOOKS twin (libphamclust_hip_hooks.so, PHAMCLUST_NATIVE_VARIANT=hooks), so this runs in its own process."""
    import subprocess
    import sys
    import tempfile
    from phamclust_amd.synth import synth_packed
    code = r'''
import numpy as np, sys
sys.path.insert(0, %r)
from phamclust_amd import hip
from phamclust_amd.synth import synth_packed
pk = synth_packed(260, 1500, seed=3)
assert hip.load().pc_test_hooks() == 1, "not the hooks build"
with hip.Context(0) as ctx:
    ctx.upload(pk)
    got, st = ctx.fill("peq", want_stats=True)
    assert st["n_chunks"] >= 2, st
    assert np.array_equal(got, np.load(sys.argv[1])), "chunked-after-OOM fill differs from the oracle"
    print("chunks", st["n_chunks"])
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # the release library has no such switch: the same environment changes nothing there
    code2 = code.replace('== 1, "not the hooks build"', '== 0, "the release build carries test hooks"').replace('assert st["n_chunks"] >= 2, st', 'assert st["n_chunks"] == 1, st')
    env = dict(os.environ, PC_FAKE_OOM_ABOVE=str(1 << 19), PHAMCLUST_NATIVE_VARIANT="hooks", PHAMCLUST_NO_TORCH="1")       # ~0.1 M alignments: the one-piece plan's A * 8-byte buffers are ~0.9 MB
    env2 = dict(os.environ, PC_FAKE_OOM_ABOVE=str(1 << 19), PHAMCLUST_NO_TORCH="1")
    env2.pop("PHAMCLUST_NATIVE_VARIANT", None)
    with tempfile.TemporaryDirectory() as tmp:
        want = os.path.join(tmp, "want.npy")
        np.save(want, _oracle().fill(synth_packed(260, 1500, seed=3), "peq"))
        runs = [subprocess.Popen([sys.executable, "-c", c, want], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for c, e in ((code, env), (code2, env2))]
        outs = [r.communicate(timeout=600)[0] for r in runs]
    assert runs[0].returncode == 0 and "chunks" in outs[0], outs[0]
    assert runs[1].returncode == 0, outs[1]


def test_two_part_upload(gpu_ctx, native_built):
    """gcs / jc / pocp / af never read a residue (metrics.py:26-157): pc_upload_sets is enough for them; aai / peq at the
    C-ABI then answer PC_ERR_STATE until pc_upload_residues, which the Python Context calls by itself."""
    import ctypes
    from phamclust_amd import hip
    from phamclust_amd.synth import synth_packed
    O = _oracle()
    packed = synth_packed(150, 800, seed=21)
    gpu_ctx.upload(packed, residues=False)
    for metric in SET_METRICS:
        assert np.array_equal(gpu_ctx.fill(metric), O.fill(packed, metric))
    lib = hip.load()
    out = np.zeros(packed.n_pairs)
    rc = lib.pc_fill(gpu_ctx._h, hip.METRIC_IDS["peq"], 1, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), None)
    assert rc == -3 and b"residues" in lib.pc_last_error()
    gpu_ctx.set_shard(1, 2, balanced=True)                      # the cost-balanced deal needs lengths only
    assert gpu_ctx.shard_pairs() > 0
    gpu_ctx.set_shard(0, 1)
    assert np.array_equal(gpu_ctx.fill("peq"), O.fill(packed, "peq"))          # the Context uploads the residues on demand
    assert np.array_equal(gpu_ctx.fill("jc"), O.fill(packed, "jc"))
    # a different packed object of another size is refused by part 2
    other = synth_packed(40, 300, seed=2)
    s = gpu_ctx._struct(other)
    gpu_ctx.upload(packed, residues=False)
    assert lib.pc_upload_residues(gpu_ctx._h, ctypes.byref(s)) == -1


def test_unsynchronised_slice_survives_reupload(gpu_ctx, native_built):
    """pc_plan_dev / pc_align_slice_dev / pc_reduce_dev without stats leave work on the caller's stream; an
    upload, pc_align_pairs or a second slice on another stream right behind them must wait for it (one "last work" event)."""
    import torch
    from phamclust_amd.synth import synth_packed
    O = _oracle()
    a_pk, b_pk = synth_packed(150, 700, seed=51), synth_packed(60, 400, seed=52)
    want_a, want_b = O.fill(a_pk, "peq"), O.fill(b_pk, "peq")
    side = torch.cuda.Stream()
    for _ in range(2):
        gpu_ctx.upload(a_pk)
        plan = gpu_ctx.plan_dev("peq", side.cuda_stream)
        n = plan["n_distinct_alignments"]
        with torch.cuda.stream(side):
            res = torch.zeros(max(n, 1), dtype=torch.int64, device="cuda:0")
            out = torch.full((a_pk.n_pairs,), -1.0, dtype=torch.float64, device="cuda:0")
        side.synchronize()
        gpu_ctx.align_slice_dev(0, 1, res.data_ptr(), side.cuda_stream, want_stats=False)        # returns with K4 in flight
        gpu_ctx.reduce_dev("peq", True, res.data_ptr(), out.data_ptr(), side.cuda_stream)
        gpu_ctx.upload(b_pk)                                                                     # rewrites codes, tables
        got_b = gpu_ctx.fill("peq")
        side.synchronize()
        assert np.array_equal(out.cpu().numpy(), want_a)
        assert np.array_equal(got_b, want_b)
    # the test hook overwrites the plan's buffers: the plan must not survive it
    gpu_ctx.upload(a_pk)
    plan = gpu_ctx.plan_dev("peq")
    gpu_ctx.align_pairs(np.array([0, 1], np.int32), np.array([2, 3], np.int32))
    res = torch.zeros(max(plan["n_distinct_alignments"], 1), dtype=torch.int64, device="cuda:0")
    with pytest.raises(Exception, match="pc_plan_dev first"):
        gpu_ctx.align_slice_dev(0, 1, res.data_ptr())


def test_borrowed_result_loan(gpu_ctx, native_built):
    """fill(borrow=True) lends pinned memory the context recycles; the view refuses access once the loan ended."""
    from phamclust_amd import hip
    from phamclust_amd.synth import synth_packed
    packed = synth_packed(60, 300, seed=5)
    gpu_ctx.upload(packed)
    lent = gpu_ctx.fill("jc", borrow=True)
    kept = lent.copy()
    assert type(kept) is np.ndarray and kept.shape == (packed.n_pairs,)
    assert float(lent[0]) == kept[0] and np.asarray(lent).shape == kept.shape
    again = gpu_ctx.fill("gcs", borrow=True)                    # ends the first loan
    with pytest.raises(hip.HipLibraryError, match="loan has ended"):
        lent[0]
    with pytest.raises(hip.HipLibraryError, match="loan has ended"):
        lent.copy()
    with pytest.raises(hip.HipLibraryError, match="loan has ended"):      # ufuncs and reductions as well
        np.sum(lent)
    with pytest.raises(hip.HipLibraryError, match="loan has ended"):
        lent + 1.0
    assert again[0] >= 0.0 and float(np.sum(again)) == float(np.sum(again.copy()))
    gpu_ctx.upload(packed)
    with pytest.raises(hip.HipLibraryError):
        again.copy()


def _set_kernel_case(rng, n_genomes, n_phams, wide_rows=(), shared_by_all=(), empty_translation_in=None):
    from phamclust_amd.genome import Genome
    from phamclust_amd.pack import pack_genomes
    genomes = []
    for k in range(n_genomes):
        g = Genome(f"g{k:04d}")
        n = 420 if k in wide_rows else int(rng.integers(0 if k % 17 == 0 else 20, 110))
        phams = set(int(x) for x in rng.choice(n_phams, size=n, replace=False)) | set(shared_by_all)
        if k == 34:
            phams = set()                                                # a genome without genes
        elif k % 5 == 0:
            phams |= set(range(100 * (k % 3), 100 * (k % 3) + 60))       # blocks of genomes sharing dozens of phams: masks with many bits
        for p in sorted(phams):
            for _ in range(int(rng.integers(1, 4)) if p % 11 == 0 else 1):
                g.add(f"p{p:05d}", "M" * int(rng.integers(1, 40)))
        if k == empty_translation_in:
            g.add("p00003", "")
        genomes.append(g)
    return pack_genomes(genomes)


def test_every_pocp_af_kernel_agrees(gpu_ctx, native_built):
    """pocp / af have four kernels behind one selector (popcount tiles + paralog excess, the 32 x 32 and 64 x 64 sparse tile
    kernels, the shared-pham walker; pc_api.hip picks by size).  Each one, forced through PC_SET_KERNEL, must give the oracle's
    matrix (metrics.py:83-157) on a data set that reaches their corners: more phams than one mask chunk holds (64 x 64: three
    chunks), rows with more entries than a wave keeps in registers, phams shared by every genome and by blocks of genomes
    (broadcast adds), genomes without genes, similarity and distance, unsharded and as a shard."""
    import torch
    O = _oracle()
    rng = np.random.default_rng(11)
    packed = _set_kernel_case(rng, 150, 40000, wide_rows=(3, 77, 149), shared_by_all=(5, 4100, 39999))
    assert packed.words_per_row * 64 > 7680                                  # (the VOCABULARY; the 64 x 64 kernel chunks over the phams with two holders
    #                                                                          or more -- ~1,300 here: test_sparse64_chunked_instances_and_forced_split drives that count)
    want = {(m, dist): O.fill(packed, m, dist) for m in ("pocp", "af") for dist in (True, False)}
    stream = torch.cuda.current_stream().cuda_stream
    try:
        for kernel in ("popc", "sparse", "sparse64", "sparsecol", "walker"):
            os.environ["PC_SET_KERNEL"] = kernel                             # read per fill
            gpu_ctx.upload(packed, residues=False)
            for (m, dist), w in want.items():
                assert np.array_equal(gpu_ctx.fill(m, dist), w), (kernel, m, dist)
                assert gpu_ctx.last_set_kernel() == kernel or (kernel, m) == ("popc", "af"), (kernel, m, gpu_ctx.last_set_kernel())   # (af has no popcount form)
            if kernel in ("popc", "sparse64", "sparsecol"):                  # gcs / jc: popcount tiles, or the sparse tiles' counting mode
                for m in ("gcs", "jc"):
                    for dist in (True, False):
                        assert np.array_equal(gpu_ctx.fill(m, dist), O.fill(packed, m, dist)), (kernel, m, dist)
                        assert gpu_ctx.last_set_kernel() == kernel
            gpu_ctx.set_shard(1, 3)
            t_rank, t_lbase = gpu_ctx.shard_table()
            n = packed.n_genomes
            for m in ("pocp", "af") + (("jc",) if kernel in ("popc", "sparse64", "sparsecol") else ()):
                buf = torch.full((gpu_ctx.shard_stride(),), -1.0, dtype=torch.float64, device="cuda:0")
                gpu_ctx.fill_shard_dev(m, True, buf.data_ptr(), stream)
                torch.cuda.synchronize()
                got = buf.cpu().numpy()
                w = want[(m, True)] if (m, True) in want else O.fill(packed, m, True)
                for t in range(1, n):
                    if t_rank[t] != 1:
                        continue
                    col = np.array([w[s * n - s * (s + 1) // 2 + (t - s - 1)] for s in range(t)])
                    assert np.array_equal(got[t_lbase[t]:t_lbase[t] + t], col), (kernel, m, t)
            gpu_ctx.set_shard(0, 1)
        # one mask chunk of more than 64 KB of LDS (5,952 < phams <= 7,680)
        mid = _set_kernel_case(rng, 120, 25000, wide_rows=(9,))
        assert 5952 < mid.words_per_row * 64 <= 7680                          # (again the vocabulary, see above)
        for kernel in ("sparse64", "sparsecol"):
            os.environ["PC_SET_KERNEL"] = kernel
            gpu_ctx.upload(mid, residues=False)
            for m in ("pocp", "af", "gcs", "jc"):
                assert np.array_equal(gpu_ctx.fill(m), O.fill(mid, m)), (kernel, m)
        os.environ["PC_SET_KERNEL"] = "sparse64"
        # no pham with two holders: the kernel's lists (phams that can be shared at all) are empty
        from phamclust_amd.genome import Genome
        from phamclust_amd.pack import pack_genomes
        lonely = []
        for k in range(5):
            g = Genome(f"lonely{k}")
            for q in range(4):
                g.add(f"own{k}_{q}", "MKT" * (q + 1))
            lonely.append(g)
        lonely = pack_genomes(lonely)
        gpu_ctx.upload(lonely, residues=False)
        for m in ("pocp", "af", "gcs", "jc"):
            assert np.array_equal(gpu_ctx.fill(m), O.fill(lonely, m)), m
        # an empty translation makes "sum == 0" ambiguous: the 64 x 64 kernel must step aside, whatever was asked for
        odd = _set_kernel_case(rng, 40, 300, empty_translation_in=2)
        gpu_ctx.upload(odd, residues=False)
        for m in ("pocp", "af"):
            assert np.array_equal(gpu_ctx.fill(m), O.fill(odd, m)), m
    finally:
        os.environ.pop("PC_SET_KERNEL", None)
        gpu_ctx.set_shard(0, 1)


def _dense_phams(packed):
    """Phams with at least two holders: what k_sparse_tile64's lists and mask chunks cover (k_sp_build renumbers them densely)."""
    bits = np.unpackbits(packed.bitmap.reshape(packed.n_genomes, packed.words_per_row).view(np.uint8), axis=1, bitorder="little")
    return int((bits.sum(axis=0) >= 2).sum())


def _two_holder_case(rng, n_genomes, per_genome, wide=None, extra_holders=0):
    """Every pham of the pool is held by exactly two genomes (then `extra_holders` random ones get a third ... holder), so the dense
    pham count is the pool size n_genomes * per_genome / 2 -- whatever the vocabulary; `wide` = (genome, entries) gets a long row."""
    from phamclust_amd.genome import Genome
    from phamclust_amd.pack import pack_genomes
    pool = n_genomes * per_genome // 2
    slots = np.repeat(np.arange(pool), 2)
    rng.shuffle(slots)
    held = [set() for _ in range(n_genomes)]
    for k, p in enumerate(slots):                           # deal the shuffled (pham, pham) list round: a genome rarely gets both copies
        g = k % n_genomes
        if int(p) in held[g]:
            g = (g + 1) % n_genomes
        held[g].add(int(p))
    for p in rng.choice(pool, size=extra_holders, replace=False):
        for g in rng.choice(n_genomes, size=int(rng.integers(3, 40)), replace=False):
            held[int(g)].add(int(p))                        # phams with many holders: masks with many bits (broadcast adds)
    if wide is not None:
        held[wide[0]] |= set(int(x) for x in rng.choice(pool, size=wide[1], replace=False))
    genomes = []
    for k in range(n_genomes):
        g = Genome(f"g{k:04d}")
        for p in sorted(held[k]):
            for _ in range(int(rng.integers(2, 5)) if p % 7 == 0 else 1):      # paralogs: pocp's second direction
                g.add(f"p{p:05d}", "M" * int(rng.integers(1, 40)))
        g.add(f"own{k:04d}", "MK")                          # a pham nobody else holds: dropped from the dense numbering
        genomes.append(g)
    return pack_genomes(genomes)


def test_sparse64_chunked_instances_and_forced_split(gpu_ctx, native_built):
    """k_sparse_tile64<*, 1> -- the one-batch instances every gcs / jc / pocp / af fill takes once the masks are split, i.e.
    from ~4,000 genomes and for every collection of more than 7,680 shareable phams (nothing reached them at test
    size, because the kernel chunks over the phams with at least two holders, not over the vocabulary).  Here the dense count
    itself is driven past 7,680 (two equal chunks, ~100 entries per row and chunk: the overflow loop behind the 64-entry batch),
    a second collection sits between 5,952 and 7,680 (one chunk of more than 64 KB of dynamic LDS, two-batch instances, a
    300-entry row for their overflow loop), and PC_S64_CHUNKS forces the split -- 2 and 3 chunks -- on it.  All four metrics,
    similarity and distance, and a shard, against the oracle (metrics.py:26-157)."""
    import torch
    O = _oracle()
    rng = np.random.default_rng(23)
    big = _two_holder_case(rng, 150, 200, wide=(77, 900), extra_holders=60)
    mid = _two_holder_case(rng, 136, 100, wide=(9, 300), extra_holders=40)
    assert _dense_phams(big) > 7680 and 5952 < _dense_phams(mid) <= 7680, (_dense_phams(big), _dense_phams(mid))
    stream = torch.cuda.current_stream().cuda_stream
    try:
        os.environ["PC_SET_KERNEL"] = "sparse64"
        for label, packed, chunks in (("big", big, None), ("mid", mid, None), ("mid", mid, "2"), ("mid", mid, "3"), ("big", big, "5"), ("mid", mid, "col")):
            os.environ["PC_SET_KERNEL"] = "sparsecol" if chunks == "col" else "sparse64"    # (k_sparse_col: its largest mask array, 7,680 phams)
            if chunks in (None, "col"):
                os.environ.pop("PC_S64_CHUNKS", None)
            else:
                os.environ["PC_S64_CHUNKS"] = chunks                        # read per launch
            gpu_ctx.upload(packed, residues=False)
            for m in ("gcs", "jc", "pocp", "af"):
                for dist in (True, False):
                    assert np.array_equal(gpu_ctx.fill(m, dist), O.fill(packed, m, dist)), (label, chunks, m, dist)
            gpu_ctx.set_shard(2, 3)
            t_rank, t_lbase = gpu_ctx.shard_table()
            n = packed.n_genomes
            for m in ("jc", "pocp", "af"):
                w = O.fill(packed, m, True)
                buf = torch.full((gpu_ctx.shard_stride(),), -1.0, dtype=torch.float64, device="cuda:0")
                gpu_ctx.fill_shard_dev(m, True, buf.data_ptr(), stream)
                torch.cuda.synchronize()
                got = buf.cpu().numpy()
                for t in range(1, n):
                    if t_rank[t] != 2:
                        continue
                    col = np.array([w[s * n - s * (s + 1) // 2 + (t - s - 1)] for s in range(t)])
                    assert np.array_equal(got[t_lbase[t]:t_lbase[t] + t], col), (label, chunks, m, t)
            gpu_ctx.set_shard(0, 1)
    finally:
        os.environ.pop("PC_SET_KERNEL", None)
        os.environ.pop("PC_S64_CHUNKS", None)
        gpu_ctx.set_shard(0, 1)


def test_pocp_paralog_lists_spill_and_wide_entries(gpu_ctx, native_built):
    """pocp on the popcount kernels: |S n T| from the popcount, plus the excess gene counts of the shared PARALOG phams from
    per-genome paralog lists (metrics.py:102-110).  Genomes with none, with dozens of paralog phams, and with thousands of
    copies of one pham, against the oracle, through both tile kernels, unsharded and as a shard."""
    import torch
    from phamclust_amd.genome import Genome
    from phamclust_amd.pack import pack_genomes
    O = _oracle()
    rng = np.random.default_rng(3)
    genomes = []
    for k in range(70):
        g = Genome(f"g{k:03d}")
        n_para = int(rng.integers(0, 30)) if k % 3 else 0              # up to 29 paralog phams: well past the 12 staged
        phams = rng.choice(300, size=int(rng.integers(20, 60)), replace=False)
        for idx, p in enumerate(sorted(phams)):
            copies = int(rng.integers(2, 5)) if idx < n_para else 1
            if k == 7 and idx == 0:
                copies = 4200                                           # excess 4,199 does not fit the packed entry
            for _ in range(copies):
                g.add(f"p{p:04d}", "MK")
        genomes.append(g)
    packed = pack_genomes(genomes)
    want = O.fill(packed, "pocp")
    try:
        for tile in ("32", "64"):
            os.environ["PC_POPC_TILE"] = tile                           # the launcher reads the knob per launch
            gpu_ctx.upload(packed, residues=False)
            assert np.array_equal(gpu_ctx.fill("pocp"), want), tile
            assert np.array_equal(gpu_ctx.fill("jc"), O.fill(packed, "jc")), tile
    finally:
        os.environ.pop("PC_POPC_TILE", None)
    gpu_ctx.set_shard(1, 3)
    stride = gpu_ctx.shard_stride()
    buf = torch.full((stride,), -1.0, dtype=torch.float64, device="cuda:0")
    gpu_ctx.fill_shard_dev("pocp", True, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    t_rank, t_lbase = gpu_ctx.shard_table()
    got = buf.cpu().numpy()
    n = packed.n_genomes
    for t in range(1, n):
        if t_rank[t] != 1:
            continue
        s = np.arange(t)
        cond = s * n - s * (s + 1) // 2 + (t - s - 1)
        assert np.array_equal(got[t_lbase[t]:t_lbase[t] + t], want[cond])
    gpu_ctx.set_shard(0, 1)


def test_upload_without_page_locked_staging(native_built):
    """pc_upload stages the raw residues through page-locked memory up to 512 MB and sends larger inputs from the caller's
    pageable buffer; PC_RAW_STAGE_MAX=0 forces that second route (own process: the knob is read once)."""
    import subprocess
    import sys
    code = r'''
import numpy as np, sys
sys.path.insert(0, %r)
from phamclust_amd import hip
from phamclust_amd.synth import synth_packed
from oracle import oracle as O
pk = synth_packed(150, 800, seed=8)
with hip.Context(0) as ctx:
    ctx.upload(pk)
    assert np.array_equal(ctx.fill("peq"), O.fill(pk, "peq"))
    ctx.upload(pk, residues=False)
    assert np.array_equal(ctx.fill("aai"), O.fill(pk, "aai"))
print("ok")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PC_RAW_STAGE_MAX="0"), capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and "ok" in run.stdout, run.stdout + run.stderr


def test_real_collection_shape_full_matrix(gpu_ctx, native_built):
    """The second workload (phamclust_amd.synth.synth_real: power-law clusters, ~6 N phams most of them with 1-3 holders, > 60 %
    byte-identical proteins inside a cluster, paralog runs up to 8, 5-8 k-residue proteins, 2-column "M" genomes) at N = 1,000:
    the WHOLE matrix of every metric against the oracle, whichever kernels the selector and the planner pick for it, plus every
    set-metric family forced (metrics.py:26-253; the selector's thresholds were all tuned on synth(N, 5000))."""
    from phamclust_amd.synth import synth_real
    O = _oracle()
    pk = synth_real(1000)
    lens = np.diff(pk.seq_off)
    assert (lens > 4096).any() and pk.n_phams > 5000 and (pk.tlen == pk.ngen).any()      # long genes, many phams, an "M" genome
    gpu_ctx.upload(pk)
    picked = {}
    n = pk.n_genomes
    rng = np.random.default_rng(8)
    filled = {}
    for metric in ALL_METRICS:
        got, st = gpu_ctx.fill(metric, want_stats=True)
        got = filled[metric] = np.asarray(got).copy()
        if metric in ("aai", "peq"):                           # (the checker needs ~1 min of 16 cores per alignment metric for the whole
            lo = rng.integers(0, n - 1, 30000); hi = rng.integers(0, n, 30000)       # matrix at this size: 30,000 random pairs each, and
            lo, hi = np.minimum(lo, hi), np.maximum(lo, hi)                          # below the WHOLE peq matrix through its definition)
            keep = lo < hi; lo, hi = lo[keep], hi[keep]
            assert np.array_equal(got[lo * n - lo * (lo + 1) // 2 + (hi - lo - 1)], O.pairs(pk, metric, lo, hi)), metric
        else:
            assert np.array_equal(got, O.fill(pk, metric)), metric
        picked[metric] = gpu_ctx.last_set_kernel() if metric in SET_METRICS else (st["n_distinct_alignments"], st["n_alignments"])
    # every peq value: round(1 - round(af, 6) * round(aai, 6), 6) (metrics.py:247-253) from the af matrix (whole, against the oracle
    # above) and the aai matrix of an independent fill
    assert np.array_equal(filled["peq"], _py_round6(1.0 - np.asarray(gpu_ctx.fill("af", as_distance=False)) * np.asarray(gpu_ctx.fill("aai", as_distance=False))))
    assert picked["peq"][0] < 0.7 * picked["peq"][1]                                       # most alignments are repeats of identical proteins
    try:
        for kernel in ("popc", "sparse", "sparse64", "sparsecol", "walker"):
            os.environ["PC_SET_KERNEL"] = kernel
            for metric in SET_METRICS:
                assert np.array_equal(gpu_ctx.fill(metric), O.fill(pk, metric)), (kernel, metric)
    finally:
        os.environ.pop("PC_SET_KERNEL", None)


def test_launch_policy_switches_change_no_value(native_built):
    """The r04 launch machinery -- tier launches (PC_FUSE), one- / two-wave workgroups for small tasks (PC_SMALL_MODES, fold
    threshold PC_SMALL_LAUNCH_MIN), strip-mined passes (PC_STRIP), the strip-mined launches' slab regions and streams
    (PC_STRIP_STREAMS, PC_SLAB_BUDGET: side by side, in line on the caller's stream, or some of each) -- is policy: every setting must give the oracle's matrix
    (metrics.py:178-253) and differ only in how many launches it takes.  The switches are read once per process, so each
    setting runs in a process of its own, on a collection with small and large buckets and a few genes beyond 4,096 residues."""
    import json
    import subprocess
    import sys
    import tempfile
    from phamclust_amd.synth import synth_real
    code = r'''
import json, sys
import numpy as np
sys.path.insert(0, %r)
from phamclust_amd import hip
from phamclust_amd.synth import synth_real
pk = synth_real(120, seed=12)
assert (np.diff(pk.seq_off) > 4096).any()
with hip.Context(0) as ctx:
    ctx.upload(pk)
    got, st = ctx.fill("peq", want_stats=True)
    assert np.array_equal(got, np.load(sys.argv[1])), "peq differs from the oracle"
    print(json.dumps({"launches": st["n_align_launches"], "tasks": st["n_tasks"]}))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seen = {}
    settings = (("default", {}), ("per_class", {"PC_FUSE": "0", "PC_SMALL_LAUNCH_MIN": "1"}), ("no_modes_no_strip", {"PC_SMALL_MODES": "0", "PC_STRIP": "0"}),
                ("strips_in_line", {"PC_STRIP_STREAMS": "0"}), ("one_strip_region_fits", {"PC_SLAB_BUDGET": "300000"}))
    with tempfile.TemporaryDirectory() as tmp:
        want = os.path.join(tmp, "want.npy")
        np.save(want, _oracle().fill(synth_real(120, seed=12), "peq"))        # the oracle's matrix once, for all of them
        for wave in (settings[:3], settings[3:]):                              # side by side, at most three processes (the box allows six on its GPU)
            procs = [(name, subprocess.Popen([sys.executable, "-c", code, want], env=dict(os.environ, PHAMCLUST_NO_TORCH="1", **env),
                                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)) for name, env in wave]
            for name, proc in procs:
                out, err = proc.communicate(timeout=900)
                assert proc.returncode == 0, (name, out[-1500:], err[-3000:])
                seen[name] = json.loads(out.strip().splitlines()[-1])
    assert seen["default"]["tasks"] == seen["per_class"]["tasks"] and seen["default"]["launches"] < seen["per_class"]["launches"], seen


def test_multi_context_one_process_many_devices(native_built):
    """pc_multi_* (pc_ctx_create(out, device_ids, n_dev)): one process, a context and a host thread per device, the
    cost-balanced deal of the target genomes, ONE exchange (each device copies its shard to the root, device to device), device-side
    assembly.  Rehearsed with 1-4 contexts on this box's one GPU: every metric equals the single-context fill bit for bit (which
    the other tests hold to the oracle), the devices' shards partition the pairs, uneven genomes included."""
    from phamclust_amd import hip
    from phamclust_amd.synth import synth_packed, synth_real
    O = _oracle()
    for packed in (synth_packed(211, 700, seed=31), synth_real(150, seed=4)):
        with hip.Context(0) as one:
            one.upload(packed)
            want = {m: np.asarray(one.fill(m)).copy() for m in ALL_METRICS + ["aai_ppos"]}
        assert np.array_equal(want["peq"], O.fill(packed, "peq"))
        for ids in ([0], [0, 0], [0, 0, 0, 0]):
            with hip.MultiContext(ids) as multi:
                multi.upload(packed, residues=False)                  # the residues follow on demand, on every device
                for m in ALL_METRICS + ["aai_ppos"]:
                    got, st = multi.fill(m, want_stats=True)
                    assert np.array_equal(got, want[m]), (ids, m)
                    assert st["n_devices"] == len(ids) and len(st["per_device"]) == len(ids)
                    assert sum(p["n_pairs"] for p in st["per_device"]) == packed.n_pairs
                    if len(ids) > 1 and m == "peq":
                        cells = [p["n_cells"] for p in st["per_device"]]
                        assert max(cells) <= 1.2 * (sum(cells) / len(cells)) + 1e6          # the deal balances the alignment work
                assert np.array_equal(multi.fill("jc", as_distance=False), O.fill(packed, "jc", as_distance=False))
    with pytest.raises(hip.HipLibraryError):
        hip.MultiContext([0, 99])


def _bench_line(argv, env_extra, timeout=900):
    import json
    import os
    import subprocess
    import sys
    from conftest import REPO
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PC_BENCH_FORCE_DIST"):
        env.pop(k, None)
    proc = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *argv], capture_output=True, text=True, timeout=timeout, env=env)
    assert proc.returncode == 0, (proc.stdout[-1500:], proc.stderr[-3000:])
    lines = [json.loads(ln) for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout[-1500:]
    return lines[0]


def test_bench_gpus_2_by_itself_spawns_two_ranks(native_built):
    """`python bench.py --gpus 2` with no launcher around it (the way the driver runs `--gpus 1`): the parent starts
    torch.distributed.run as a child before touching the GPU, two ranks meet, fill their shards, ONE gather, assembly, and the
    parent relays rank 0's line.  This box has one GPU, so the ranks share it over the gloo rehearsal transport (RCCL refuses two
    ranks on one device); on a node with >= 2 GPUs the same command runs over RCCL.  Replaces the reference's joblib fan-out
    (matrix.py:471-472,488-491)."""
    d = _bench_line(["--gpus", "2", "--genomes", "301", "--steps", "2", "--warmup", "1", "--verify-pairs", "3000", "--cpu-seconds", "0"],
                    {"PC_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["route"] == "rank" and d["devices_visible"] >= 1
    assert d["verified"]["bit_exact"] is True and d["valid"] is True
    assert d["launched_by"]["parent_modules_touching_gpu"] == [] and d["launched_by"]["child_exit_status"] == 0
    assert d["shards"]["pairs_min_max"][0] > 0 and d["stage_ms"]["exchange_rank0"] > 0


@pytest.mark.parametrize("metric", ["peq", "jc"])
def test_bench_route_process_drives_the_devices_from_one_process(native_built, metric):
    """`bench.py --gpus 2 --route process`: hip.MultiContext (pc_multi_*) timed by the same harness, so that on a multi-GPU node the
    peer-copy exchange can be put next to the RCCL gather.  Here: two contexts on the one GPU; the line says so, and says how
    every shard travelled (pc_multi_peer_access)."""
    d = _bench_line(["--gpus", "2", "--route", "process", "--genomes", "301", "--steps", "2", "--warmup", "1", "--verify-pairs", "3000",
                     "--cpu-seconds", "0", "--metric", metric], {"PC_BENCH_DEVICE_IDS": "0,0"})
    assert d["route"] == "process" and d["n_gpus"] == 2 and d["device_ids"] == [0, 0]
    assert d["verified"]["bit_exact"] is True and "REHEARSAL" in d["config"]["parallelism"]
    peers = d["peer_access"]
    assert [p["access"] for p in peers["devices"]] == ["same-device", "same-device"] and peers["staged_through_host"] == 0
    assert sum(d["shards"]["pairs_min_max"]) == 301 * 300 // 2 and d["stage_ms"]["assemble_root"] > 0
    assert d["roofline"]["frac"] > 0


def test_multi_context_reports_peer_access(native_built):
    """pc_multi_create records, per device, whether its copies to the root are peer copies -- and the answer is part of every
    fill's stats.  With two or more GPUs on the box, the cross-device branch (hipMemcpyPeerAsync after hipDeviceEnablePeerAccess)
    runs for real and must give the single-GPU matrix bit for bit."""
    import torch
    from phamclust_amd import hip
    from phamclust_amd.synth import synth_packed
    packed = synth_packed(157, 600, seed=8)
    with hip.Context(0) as one:
        one.upload(packed)
        want = {m: np.asarray(one.fill(m)).copy() for m in ("jc", "af", "peq")}
    ids = [0, 1] if torch.cuda.device_count() >= 2 else [0, 0]
    with hip.MultiContext(ids) as multi:
        acc = multi.peer_access()
        assert [d["device"] for d in acc["devices"]] == ids and acc["devices"][0]["access"] == "same-device"
        if ids[1] != ids[0]:
            assert acc["devices"][1]["code"] in (1, 0, -1)
            assert (acc["staged_through_host"] == 0) == (acc["devices"][1]["code"] == 1) and bool(acc["note"]) == bool(acc["staged_through_host"])
        else:
            assert acc["staged_through_host"] == 0 and acc["note"] == ""
        multi.upload(packed)
        for m, w in want.items():
            got, st = multi.fill(m, want_stats=True)
            assert np.array_equal(got, w), m
            assert st["peer_access"] == acc


@pytest.mark.parametrize("metric", ALL_METRICS)
def test_reference_written_synth200(gpu_ctx, synth200_packed, metric):
    """SURVEY 8(d) config 1's live-reference subset, through the C-ABI: the first 200 genomes of synth(2000,5000), 19,900 pairs per
    metric, against files the reference itself wrote (tests/golden/synth200/, matrix.py:432-497 + scripts/phamclust.py:254-268).
    gcs / jc / pocp ``==``; af / aai / peq <= 1e-6 (north_star's tolerance, written here) -- and, beyond the contract, equal."""
    from conftest import read_lower_triangle, synth200_file
    names, gold, diag = read_lower_triangle(synth200_file(metric))
    assert names == synth200_packed.names and gold.shape == (19900,) and not diag.any()
    got = np.asarray(gpu_ctx.upload(synth200_packed).fill(metric, as_distance=True))
    if metric in ("gcs", "jc", "pocp"):
        assert np.array_equal(got, gold)
    else:
        assert np.max(np.abs(got - gold)) <= 1e-6
        assert np.array_equal(got, gold)
    # the pipeline reads the file back as its cache (scripts/phamclust.py:254-257): what matrix_de_novo returns must print to it
    from phamclust_amd.matrix import SymMatrix
    m = SymMatrix.from_condensed(names, got, is_distance=True)
    assert m.get_weight(names[3], names[150]) == gold[3 * 200 - 3 * 4 // 2 + (150 - 3 - 1)]


def test_column_kernel_value_modes(gpu_ctx, native_built):
    """pocp and af on the column kernel (k_sparse_col): a hit adds the source entry's value and the TARGET's, which is looked up in a
    table of the target block's values laid out pham by pham in LDS (16-bit values, as many as two workgroups per CU leave room for).
    Collections with few and with many paralogs (counts up to 5 per pham) run there -- unsharded and as a shard, every value against
    the oracle (metrics.py:83-157); one whose blocks hold more entries than the table, and one with an entry whose summed length
    passes 65,535 (af), must be sent to another kernel by the selector, not truncated -- while gcs / jc keep the column kernel."""
    import torch
    from phamclust_amd.genome import Genome
    from phamclust_amd.pack import pack_genomes
    O = _oracle()
    rng = np.random.default_rng(77)

    def collection(n_genomes, paralogs_per_genome, n_phams=900, per_genome=(30, 90), long_entry=False):
        genomes = []
        for k in range(n_genomes):
            g = Genome(f"g{k:04d}")
            phams = sorted(set(int(x) for x in rng.choice(n_phams, size=int(rng.integers(*per_genome)), replace=False)) | set(range(10 * (k % 7), 10 * (k % 7) + 25)))
            para = set(int(x) for x in rng.choice(phams, size=min(paralogs_per_genome, len(phams)), replace=False))
            for p in phams:
                for _ in range(int(rng.integers(2, 6)) if p in para else 1):
                    g.add(f"p{p:04d}", "MK" * int(rng.integers(1, 30)))
            if long_entry and k == 3:
                g.add("p0001", "A" * 40000); g.add("p0001", "C" * 40000)          # one (genome, pham) entry of > 80,000 residues
            genomes.append(g)
        return pack_genomes(genomes)

    stream = torch.cuda.current_stream().cuda_stream
    try:
        os.environ["PC_SET_KERNEL"] = "sparsecol"
        for packed in (collection(200, 6), collection(130, 40)):
            gpu_ctx.upload(packed, residues=False)
            for m in ("pocp", "af"):
                for dist in (True, False):
                    assert np.array_equal(gpu_ctx.fill(m, dist), O.fill(packed, m, dist)) and gpu_ctx.last_set_kernel() == "sparsecol", (m, dist)
            gpu_ctx.set_shard(1, 3)
            t_rank, t_lbase = gpu_ctx.shard_table()
            n = packed.n_genomes
            for m in ("pocp", "af"):
                want = O.fill(packed, m, True)
                buf = torch.full((gpu_ctx.shard_stride(),), -1.0, dtype=torch.float64, device="cuda:0")
                gpu_ctx.fill_shard_dev(m, True, buf.data_ptr(), stream)
                torch.cuda.synchronize()
                assert gpu_ctx.last_set_kernel() == "sparsecol"
                got = buf.cpu().numpy()
                for t in range(1, n):
                    if t_rank[t] == 1:
                        col = np.array([want[s * n - s * (s + 1) // 2 + (t - s - 1)] for s in range(t)])
                        assert np.array_equal(got[t_lbase[t]:t_lbase[t] + t], col), (m, t)
            gpu_ctx.set_shard(0, 1)
        crowded = collection(100, 5, n_phams=5000, per_genome=(380, 420))          # 64 x ~400 entries per block: more than the value table holds
        gpu_ctx.upload(crowded, residues=False)
        for m in ("pocp", "af"):
            assert np.array_equal(gpu_ctx.fill(m), O.fill(crowded, m)) and gpu_ctx.last_set_kernel() != "sparsecol", m
        assert np.array_equal(gpu_ctx.fill("jc"), O.fill(crowded, "jc")) and gpu_ctx.last_set_kernel() == "sparsecol"
        longish = collection(90, 6, long_entry=True)
        gpu_ctx.upload(longish, residues=False)
        assert np.array_equal(gpu_ctx.fill("af"), O.fill(longish, "af")) and gpu_ctx.last_set_kernel() != "sparsecol"
        assert np.array_equal(gpu_ctx.fill("pocp"), O.fill(longish, "pocp")) and gpu_ctx.last_set_kernel() == "sparsecol"
    finally:
        os.environ.pop("PC_SET_KERNEL", None)


@pytest.mark.parametrize("n_genomes", [2, 3, 63, 64, 65, 129, 200])
def test_column_kernel_edge_sizes(gpu_ctx, native_built, n_genomes):
    """k_sparse_col at the edges of its tiling (blocks and tiles of 64 genomes; 16 waves x 4 rows): fewer genomes than a tile, exactly one,
    one more, a ragged last block -- all four set metrics, similarity and distance, unsharded and as the shards of a 2-rank deal, every
    value against the oracle (metrics.py:26-157), with the kernel forced (the selector would not pick it at these sizes)."""
    import torch
    from phamclust_amd.synth import synth_packed
    O = _oracle()
    packed = synth_packed(n_genomes, 700, seed=100 + n_genomes)
    stream = torch.cuda.current_stream().cuda_stream
    try:
        os.environ["PC_SET_KERNEL"] = "sparsecol"
        gpu_ctx.upload(packed, residues=False)
        for m in SET_METRICS:
            for dist in (True, False):
                assert np.array_equal(gpu_ctx.fill(m, dist), O.fill(packed, m, dist)), (m, dist)
                assert gpu_ctx.last_set_kernel() == "sparsecol", m
        n = packed.n_genomes
        for rank in range(2):
            gpu_ctx.set_shard(rank, 2)
            t_rank, t_lbase = gpu_ctx.shard_table()
            for m in SET_METRICS:
                want = O.fill(packed, m, True)
                buf = torch.full((max(gpu_ctx.shard_stride(), 1),), -1.0, dtype=torch.float64, device="cuda:0")
                gpu_ctx.fill_shard_dev(m, True, buf.data_ptr(), stream)
                torch.cuda.synchronize()
                got = buf.cpu().numpy()
                for t in range(1, n):
                    if t_rank[t] == rank:
                        col = np.array([want[s * n - s * (s + 1) // 2 + (t - s - 1)] for s in range(t)])
                        assert np.array_equal(got[t_lbase[t]:t_lbase[t] + t], col), (m, rank, t)
        gpu_ctx.set_shard(0, 1)
    finally:
        os.environ.pop("PC_SET_KERNEL", None)


def test_column_kernel_at_20000_genomes(gpu_ctx, native_built):
    """The size the column kernel exists for: synth(20000, 5000), 199,990,000 pairs per metric.  Its matrix (the selector's own choice
    there) must equal, value for value, the matrix of an independent kernel family -- the 64 x 64 sparse tiles for all four metrics,
    the popcount tiles as well for jc -- and 100,000 random pairs per metric must equal the oracle (metrics.py:26-157)."""
    from phamclust_amd.synth import synth_packed
    O = _oracle()
    packed = synth_packed(20000, 5000)
    n = packed.n_genomes
    gpu_ctx.upload(packed, residues=False)
    rng = np.random.default_rng(2020)
    a, b = rng.integers(0, n, 100000), rng.integers(0, n, 100000)
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    keep = lo < hi
    lo, hi = lo[keep], hi[keep]
    idx = lo * n - lo * (lo + 1) // 2 + (hi - lo - 1)
    try:
        for m in SET_METRICS:
            os.environ.pop("PC_SET_KERNEL", None)
            got = np.asarray(gpu_ctx.fill(m, borrow=True)).copy()
            assert gpu_ctx.last_set_kernel() == "sparsecol" and got.shape == (n * (n - 1) // 2,)
            assert np.array_equal(got[idx], O.pairs(packed, m, lo, hi, as_distance=True)), m
            for other in ("sparse64",) + (("popc",) if m == "jc" else ()):
                os.environ["PC_SET_KERNEL"] = other
                ref = gpu_ctx.fill(m, borrow=True)
                assert gpu_ctx.last_set_kernel() == other
                assert np.array_equal(got, np.asarray(ref)), (m, other)
    finally:
        os.environ.pop("PC_SET_KERNEL", None)
