"""Python face of the CPU oracle (TEST INFRASTRUCTURE ONLY -- see pc_oracle.c header).

Two independent restatements of the reference's hot path live here:

* the C library ``libpc_oracle.so`` (closed forms on packed arrays + the aligner),
  reached through ctypes;
* ``py_*`` functions: pure-Python per-pair code over ``Genome``-like objects (anything
  with ``.phams: dict[str, list[str]]``), written from the reference's per-pair
  semantics (metrics.py:26-253), for small cases only.

Only tests/, ``__graft_entry__.smoke()`` and bench.py's cpu_baseline leg may import
this module.  aai/peq: PARITY UNPINNED vs parasail for co-optimal alignment ties.
"""

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpc_oracle.so")
METRIC_IDS = {"gcs": 0, "jc": 1, "pocp": 2, "af": 3, "aai": 4, "peq": 5, "aai_ppos": 6}

_u8p = ctypes.POINTER(ctypes.c_uint8)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_f64p = ctypes.POINTER(ctypes.c_double)


class _Packed(ctypes.Structure):
    _fields_ = [("n_genomes", ctypes.c_int32), ("n_phams", ctypes.c_int32),
                ("words_per_row", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("bitmap", _u64p), ("nph", _i32p), ("ngen", _i32p), ("tlen", _i64p),
                ("gene_off", _i64p), ("gene_pham", _i32p), ("seq_off", _i64p), ("residues", _u8p)]


def _lib_path():
    """libpc_oracle.so next to its sources, or the AddressSanitizer + UBSan twin under oracle/asan/ when
    PHAMCLUST_NATIVE_VARIANT=asan (built by `make -C oracle asan`; tests/test_sanitized.py)."""
    if os.environ.get("PHAMCLUST_NATIVE_VARIANT") == "asan":
        return os.path.join(_HERE, "asan", "libpc_oracle.so")
    return _LIB_PATH


def build(force=False):
    """Compile the oracle with gcc (a few seconds).  Building the checker is not using it."""
    if os.environ.get("PHAMCLUST_NATIVE_VARIANT") == "asan":
        subprocess.check_call(["make", "-C", _HERE, "asan"], stdout=subprocess.DEVNULL)
        return _lib_path()
    newest = max(os.path.getmtime(os.path.join(_HERE, name)) for name in ("pc_oracle.c", "pc_cooptimal.c", "Makefile"))
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < newest:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpc_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_lib_path())
        L.pco_nw_stats.argtypes = [_u8p, ctypes.c_int, _u8p, ctypes.c_int, _i32p, _i32p, _i32p]
        L.pco_nw_traceback.argtypes = [_u8p, ctypes.c_int, _u8p, ctypes.c_int, ctypes.c_char_p,
                                       ctypes.c_char_p, ctypes.c_char_p, _i32p]
        L.pco_nw_batch.argtypes = [_u8p, _i64p, _i32p, _i32p, ctypes.c_int64, _i32p, _i32p, _i32p, ctypes.c_int]
        L.pco_round6.argtypes = [ctypes.c_double]
        L.pco_div_million_mismatches.argtypes = [ctypes.c_int64]
        L.pco_div_million_mismatches.restype = ctypes.c_int64
        L.pco_round6.restype = ctypes.c_double
        L.pco_pair.argtypes = [ctypes.POINTER(_Packed), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.pco_pair.restype = ctypes.c_double
        L.pco_fill.argtypes = [ctypes.POINTER(_Packed), ctypes.c_int, ctypes.c_int, _f64p, ctypes.c_int]
        L.pco_fill_rows.argtypes = [ctypes.POINTER(_Packed), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    _f64p, ctypes.c_int, _i64p]
        L.pco_pairs.argtypes = [ctypes.POINTER(_Packed), ctypes.c_int, ctypes.c_int, _i32p, _i32p, ctypes.c_int64, _f64p, ctypes.c_int]
        L.pco_blosum62.argtypes = [ctypes.c_int, ctypes.c_int]
        L.pco_map.argtypes = [ctypes.c_int]
        L.pco_set_tie_rule.argtypes = [ctypes.c_int]
        L.pco_set_tie_rule.restype = None
        L.pco_tie_sensitivity.argtypes = [ctypes.POINTER(_Packed), _i32p, _i32p, ctypes.c_int64, _f64p, _f64p, _i64p, ctypes.c_int]
        L.pco_set_gap.argtypes = [ctypes.c_int, ctypes.c_int]
        L.pco_set_gap.restype = None
        L.pco_set_compat.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.pco_set_compat.restype = None
        L.pcc_cooptimal.argtypes = [_u8p, ctypes.c_int, _u8p, ctypes.c_int, ctypes.c_int, _i32p, _u64p, _i32p]
        L.pcc_batch.argtypes = [_u8p, _i64p, _i32p, _i32p, ctypes.c_int64, ctypes.c_int, _i32p, _u64p, _i32p, ctypes.c_int]
        L.pcc_enumerate.argtypes = [ctypes.POINTER(_Packed), _i32p, _i32p, ctypes.c_int64, _i32p, _i32p, _i64p, ctypes.c_int64]
        L.pcc_enumerate.restype = ctypes.c_int64
        _lib = L
    return _lib


def _ptr(arr, typ):
    return arr.ctypes.data_as(typ)


def _as_bytes(seq):
    return seq if isinstance(seq, (bytes, bytearray)) else seq.encode("latin-1")


def nw_stats(seq_a, seq_b):
    """One-pass formulation -> (score, n_identical, n_diagonal_columns)."""
    a, b = _as_bytes(seq_a), _as_bytes(seq_b)
    sc, ni, nd = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    ab = (ctypes.c_uint8 * len(a)).from_buffer_copy(a)
    bb = (ctypes.c_uint8 * len(b)).from_buffer_copy(b)
    rc = lib().pco_nw_stats(ab, len(a), bb, len(b), ctypes.byref(sc), ctypes.byref(ni), ctypes.byref(nd))
    if rc != 0:
        raise ValueError("pco_nw_stats failed (empty sequence?)")
    return sc.value, ni.value, nd.value


class Traceback:
    """What the reference reads from parasail's ``result.get_traceback(...)``
    (metrics.py:175, 216-220): ``.query``, ``.comp``, ``.ref``."""

    def __init__(self, query, comp, ref, score):
        self.query, self.comp, self.ref, self.score = query, comp, ref, score


def nw_traceback(seq_a, seq_b):
    """Table + traceback formulation -> :class:`Traceback`."""
    a, b = _as_bytes(seq_a), _as_bytes(seq_b)
    n = len(a) + len(b) + 1
    q, c, r = ctypes.create_string_buffer(n), ctypes.create_string_buffer(n), ctypes.create_string_buffer(n)
    sc = ctypes.c_int32()
    ab = (ctypes.c_uint8 * len(a)).from_buffer_copy(a)
    bb = (ctypes.c_uint8 * len(b)).from_buffer_copy(b)
    length = lib().pco_nw_traceback(ab, len(a), bb, len(b), q, c, r, ctypes.byref(sc))
    if length < 0:
        raise ValueError("pco_nw_traceback failed (empty sequence?)")
    return Traceback(q.raw[:length].decode("latin-1"), c.raw[:length].decode("latin-1"),
                     r.raw[:length].decode("latin-1"), sc.value)


def nw_batch(residues, seq_off, a_idx, b_idx, nthreads=0):
    """(score, n_ident, n_diag) int32 arrays for gene-index pairs into a packed residue buffer."""
    a_idx = np.ascontiguousarray(a_idx, dtype=np.int32)
    b_idx = np.ascontiguousarray(b_idx, dtype=np.int32)
    n = a_idx.shape[0]
    sc, ni, nd = (np.zeros(n, np.int32) for _ in range(3))
    rc = lib().pco_nw_batch(_ptr(residues, _u8p), _ptr(seq_off, _i64p), _ptr(a_idx, _i32p), _ptr(b_idx, _i32p),
                            n, _ptr(sc, _i32p), _ptr(ni, _i32p), _ptr(nd, _i32p), nthreads)
    if rc != 0:
        raise ValueError("pco_nw_batch failed")
    return sc, ni, nd


def div_million_mismatches(k_max):
    """Integers k in [0, k_max] for which the device's three-instruction k / 1e6 (pc_div_million) would differ from the division."""
    return int(lib().pco_div_million_mismatches(int(k_max)))


def round6(x):
    return lib().pco_round6(float(x))


def _struct(packed):
    s = _Packed(packed.n_genomes, packed.n_phams, packed.words_per_row, 0,
                _ptr(packed.bitmap, _u64p), _ptr(packed.nph, _i32p), _ptr(packed.ngen, _i32p),
                _ptr(packed.tlen, _i64p), _ptr(packed.gene_off, _i64p), _ptr(packed.gene_pham, _i32p),
                _ptr(packed.seq_off, _i64p), _ptr(packed.residues, _u8p))
    return s


def pair(packed, metric, s, t, as_distance=True):
    _require_default_switches("pair")
    return lib().pco_pair(ctypes.byref(_struct(packed)), METRIC_IDS[metric], int(as_distance), s, t)


def pairs(packed, metric, s_idx, t_idx, as_distance=True, nthreads=0):
    """Values of the listed pairs (s < t required, as in the matrix's upper triangle)."""
    _require_default_switches("pairs")
    s_idx = np.ascontiguousarray(s_idx, dtype=np.int32)
    t_idx = np.ascontiguousarray(t_idx, dtype=np.int32)
    assert (s_idx < t_idx).all()
    out = np.zeros(s_idx.shape[0], dtype=np.float64)
    lib().pco_pairs(ctypes.byref(_struct(packed)), METRIC_IDS[metric], int(as_distance), _ptr(s_idx, _i32p),
                    _ptr(t_idx, _i32p), s_idx.shape[0], _ptr(out, _f64p), nthreads)
    return out


def fill(packed, metric, as_distance=True, nthreads=0):
    """Condensed (scipy order) f64 vector of the N(N-1)/2 pair values."""
    _require_default_switches("fill")
    out = np.zeros(packed.n_genomes * (packed.n_genomes - 1) // 2, dtype=np.float64)
    rc = lib().pco_fill(ctypes.byref(_struct(packed)), METRIC_IDS[metric], int(as_distance), _ptr(out, _f64p), nthreads)
    if rc != 0:
        raise RuntimeError("pco_fill failed")
    return out


def fill_rows(packed, metric, row_begin, row_end, as_distance=True, nthreads=0):
    """Rows [row_begin,row_end) only (bounded CPU-baseline sample).  Returns (condensed, n_aln, n_cells)."""
    _require_default_switches("fill_rows")
    out = np.zeros(packed.n_genomes * (packed.n_genomes - 1) // 2, dtype=np.float64)
    stats = np.zeros(2, dtype=np.int64)
    rc = lib().pco_fill_rows(ctypes.byref(_struct(packed)), METRIC_IDS[metric], int(as_distance), row_begin, row_end,
                             _ptr(out, _f64p), nthreads, _ptr(stats, _i64p))
    if rc != 0:
        raise RuntimeError("pco_fill_rows failed")
    return out, int(stats[0]), int(stats[1])


def n_threads():
    return lib().pco_threads()


# ---------------------------------------------------------------------------
# Tie-break rule table of the aligner (pc_oracle.c, "Tie-break rule table").  Rule 0 is SURVEY 8c as recalled.
# ---------------------------------------------------------------------------
N_TIE_RULES = 16            # bit 3 (gap state before DIAG) is oracle-only; the HIP kernels implement rules 0..7
TIE_RULE_BITS = {1: "H pointer: INS (E) before DEL (F) when both tie", 2: "E pointer: open on open == extend",
                 4: "F pointer: open on open == extend", 8: "H pointer: a tying gap state before DIAG"}


def set_tie_rule(rule):
    """Process-wide; every aligner entry point of the C oracle (and py_aai through nw_traceback) follows it."""
    lib().pco_set_tie_rule(int(rule))


def get_tie_rule():
    return lib().pco_get_tie_rule()


class tie_rule:
    """``with tie_rule(r): ...`` -- restores rule 0 afterwards."""

    def __init__(self, rule):
        self.rule = rule

    def __enter__(self):
        set_tie_rule(self.rule)

    def __exit__(self, *exc):
        set_tie_rule(0)


def tie_sensitivity(packed, s_idx, t_idx, nthreads=0):
    """All 16 rule combinations in one pass over the listed genome pairs.
    Returns (aai_sim[n,16], peq_sim[n,16], counters[16,8]); counter columns: alignments, changed (ident or length),
    ident changed, length changed, max |d ident|, max |d length|, identity fraction changed, unused."""
    s_idx = np.ascontiguousarray(s_idx, dtype=np.int32)
    t_idx = np.ascontiguousarray(t_idx, dtype=np.int32)
    assert (s_idx < t_idx).all()
    n = s_idx.shape[0]
    aai = np.zeros((n, N_TIE_RULES), dtype=np.float64)
    peq = np.zeros((n, N_TIE_RULES), dtype=np.float64)
    counters = np.zeros((N_TIE_RULES, 8), dtype=np.int64)
    lib().pco_tie_sensitivity(ctypes.byref(_struct(packed)), _ptr(s_idx, _i32p), _ptr(t_idx, _i32p), n, _ptr(aai, _f64p),
                              _ptr(peq, _f64p), _ptr(counters, _i64p), nthreads)
    return aai, peq, counters


# ---------------------------------------------------------------------------
# Co-optimal certificate (pc_cooptimal.c): an independent three-state DP that counts ALL optimal alignments of a
# sequence pair and the range of (n_ident, n_diag) over them.  min == max for both => the statistics are the same for
# every optimal alignment, so any correct Needleman-Wunsch (parasail included) must report them.
# ---------------------------------------------------------------------------
COUNT_SATURATED = (1 << 64) - 1


def cooptimal(seq_a, seq_b, ppos=False):
    """-> (score, n_optimal_alignments, (id_lo, id_hi), (dg_lo, dg_hi))"""
    a, b = _as_bytes(seq_a), _as_bytes(seq_b)
    sc, cnt = ctypes.c_int32(), ctypes.c_uint64()
    rng = (ctypes.c_int32 * 4)()
    ab = (ctypes.c_uint8 * len(a)).from_buffer_copy(a)
    bb = (ctypes.c_uint8 * len(b)).from_buffer_copy(b)
    if lib().pcc_cooptimal(ab, len(a), bb, len(b), int(bool(ppos)), ctypes.byref(sc), ctypes.byref(cnt), rng) != 0:
        raise ValueError("pcc_cooptimal failed (empty sequence?)")
    return sc.value, cnt.value, (rng[0], rng[1]), (rng[2], rng[3])


def cooptimal_batch(residues, seq_off, a_idx, b_idx, ppos=False, nthreads=0):
    """(score int32[n], count uint64[n], range int32[n,4] = id_lo, id_hi, dg_lo, dg_hi) for gene-index pairs."""
    a_idx = np.ascontiguousarray(a_idx, dtype=np.int32)
    b_idx = np.ascontiguousarray(b_idx, dtype=np.int32)
    n = a_idx.shape[0]
    sc, cnt, rng = np.zeros(n, np.int32), np.zeros(n, np.uint64), np.zeros((n, 4), np.int32)
    if lib().pcc_batch(_ptr(residues, _u8p), _ptr(seq_off, _i64p), _ptr(a_idx, _i32p), _ptr(b_idx, _i32p), n, int(bool(ppos)),
                       _ptr(sc, _i32p), _ptr(cnt, _u64p), _ptr(rng, _i32p), nthreads) != 0:
        raise ValueError("pcc_batch failed")
    return sc, cnt, rng


def enumerate_alignments(packed, s_idx, t_idx):
    """The alignments the reference runs for the listed genome pairs, in its loop order (metrics.py:203-214).
    -> (a_gene int32[A] = seq_a / rows, b_gene int32[A] = seq_b / columns, pair_of int64[A] = index into the pair list)."""
    s_idx = np.ascontiguousarray(s_idx, dtype=np.int32)
    t_idx = np.ascontiguousarray(t_idx, dtype=np.int32)
    st = _struct(packed)
    n = s_idx.shape[0]
    total = lib().pcc_enumerate(ctypes.byref(st), _ptr(s_idx, _i32p), _ptr(t_idx, _i32p), n, None, None, None, 0)
    a, b, q = np.zeros(total, np.int32), np.zeros(total, np.int32), np.zeros(total, np.int64)
    if total:
        lib().pcc_enumerate(ctypes.byref(st), _ptr(s_idx, _i32p), _ptr(t_idx, _i32p), n, _ptr(a, _i32p), _ptr(b, _i32p), _ptr(q, _i64p), total)
    return a, b, q


_DEFAULT_SWITCHES = {"gap": (11, 1), "compat": (False, 23, False)}
_switches = dict(_DEFAULT_SWITCHES)       # what the C library was last told (its switches are process-global)
_scoped = 0                               # > 0 inside `with gap(...)` / `with compat(...)`: non-default on purpose


def set_gap(open_=11, extend=1):
    """Gap costs of every aligner entry point (process-wide); the reference's are 11 / 1 (metrics.py:160).  Prefer
    ``with gap(12, 1): ...``, which restores the default; a bare call leaves the checker switched for the rest of the process,
    and ``fill`` / ``pairs`` / ``pair`` / ``fill_rows`` then refuse to answer (see ``_require_default_switches``)."""
    lib().pco_set_gap(int(open_), int(extend))
    _switches["gap"] = (int(open_), int(extend))


def set_compat(case_sensitive=False, unknown_row=23, lower_unknown=False):
    """Recalled parasail behaviours as switches; defaults restore them.  Prefer ``with compat(...)``."""
    lib().pco_set_compat(int(bool(case_sensitive)), int(unknown_row), int(bool(lower_unknown)))
    _switches["compat"] = (bool(case_sensitive), int(unknown_row), bool(lower_unknown))


class _scoped_switch:
    def __init__(self, setter, args, kwargs):
        self._setter, self._args, self._kwargs = setter, args, kwargs

    def __enter__(self):
        global _scoped
        _scoped += 1
        self._setter(*self._args, **self._kwargs)

    def __exit__(self, *exc):
        global _scoped
        self._setter()                                        # the defaults
        _scoped -= 1


def gap(open_=11, extend=1):
    """``with gap(12, 1): ...`` -- the other affine convention for the block, 11 / 1 again afterwards."""
    return _scoped_switch(set_gap, (open_, extend), {})


def compat(**kwargs):
    """``with compat(case_sensitive=True): ...`` -- one recalled convention flipped for the block, defaults afterwards."""
    return _scoped_switch(set_compat, (), kwargs)


def _require_default_switches(what):
    """The matrix entry points are what the GPU path is compared with: they answer only with the reference's conventions in
    force (a stray set_gap / set_compat would silently turn every later comparison of the process into one against
    another aligner), unless the caller is inside one of the context managers above."""
    if _scoped == 0 and _switches != _DEFAULT_SWITCHES:
        raise RuntimeError(f"oracle.{what}: the checker's conventions were left switched ({_switches}); "
                           f"call set_gap() / set_compat() to restore the defaults, or use `with gap(...)` / `with compat(...)`")


# ---------------------------------------------------------------------------
# Pure-Python per-pair restatement (small cases).  Each function follows the
# reference function of the same role; nothing is imported from the reference.
# ---------------------------------------------------------------------------
def _finish(similarity, as_distance):
    return round(1.0 - similarity, 6) if as_distance else round(similarity, 6)


def _shared(source, target):
    return set(source.phams) & set(target.phams)


def _n_genes(genome):
    return sum(len(v) for v in genome.phams.values())


def py_gcs(source, target, as_distance=False):          # metrics.py:26-53
    shared = _shared(source, target)
    sim = 0.0 if not shared else 2.0 * len(shared) / (len(source.phams) + len(target.phams))
    return _finish(sim, as_distance)


def py_jc(source, target, as_distance=False):           # metrics.py:56-80
    shared = _shared(source, target)
    sim = 0.0 if not shared else len(shared) / len(set(source.phams) | set(target.phams))
    return _finish(sim, as_distance)


def py_pocp(source, target, as_distance=False):         # metrics.py:83-115
    shared = _shared(source, target)
    if not shared:
        return _finish(0.0, as_distance)
    conserved = sum(len(source.phams[p]) for p in shared) + sum(len(target.phams[p]) for p in shared)
    return _finish(conserved / (_n_genes(source) + _n_genes(target)), as_distance)


def py_af(source, target, as_distance=False):           # metrics.py:118-157
    shared = _shared(source, target)
    if not shared:
        return _finish(0.0, as_distance)
    conserved = total = 0
    for genome in (source, target):
        for pham, translations in genome.phams.items():
            for translation in translations:
                if pham in shared:
                    conserved += len(translation)
                total += len(translation)
    return _finish(conserved / total, as_distance)


def py_aai(source, target, ppos=False, as_distance=False):   # metrics.py:178-232
    shared = _shared(source, target)
    if not shared:
        return _finish(0.0, as_distance)
    identities, lengths = [], []
    for pham in sorted(shared):
        anchors, others = source.phams[pham], target.phams[pham]
        if len(anchors) > len(others):
            anchors, others = others, anchors
        for anchor in anchors:
            candidates = []
            for other in others:
                tb = nw_traceback(anchor, other)
                hits = float(tb.comp.count("|")) + (float(tb.comp.count("+")) if ppos else 0.0)
                candidates.append((hits / len(tb.query), len(tb.query)))
            best = sorted(candidates, key=lambda c: c[0])[-1]
            identities.append(best[0])
            lengths.append(best[1])
    sim = float(sum([x * w for x, w in zip(identities, lengths)])) / sum(lengths)
    return _finish(sim, as_distance)


def py_peq(source, target, as_distance=False):          # metrics.py:235-253
    sim = py_af(source, target) * py_aai(source, target)
    return _finish(sim, as_distance)


PY_METRICS = {"gcs": py_gcs, "jc": py_jc, "pocp": py_pocp, "af": py_af, "aai": py_aai, "peq": py_peq}
