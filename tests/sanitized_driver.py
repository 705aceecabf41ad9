"""Run under AddressSanitizer + UBSan (tests/test_sanitized.py starts this file as a subprocess with LD_PRELOAD=libasan and
PHAMCLUST_NATIVE_VARIANT=asan): the C code that reads user files and writes into caller-sized buffers -- the TSV loader
/ packer, the "%.6f" row formatter and parser (csrc/pc_pack.c) --, the synthetic-data generator (csrc/pc_synth.c) and the
checker (oracle/pc_oracle.c, oracle/pc_cooptimal.c).  Every section must end without a sanitizer report; malformed input
must come back as a Python exception, never as a crash.  Prints one "ok <section>" line per section."""

import os
import random
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
assert os.environ.get("PHAMCLUST_NATIVE_VARIANT") == "asan"

from phamclust_amd import matrix as M                                         # noqa: E402
from phamclust_amd import pack                                                # noqa: E402
from phamclust_amd.build import native_path                                   # noqa: E402
from phamclust_amd.synth import synth_packed, write_tsv_packed                # noqa: E402

assert "asan" in native_path("libpc_pack.so")
GOLDEN = os.path.join(REPO, "tests", "golden")
tmp = tempfile.mkdtemp(prefix="pc_asan_")


def same_packed(a, b):
    for name in ("bitmap", "nph", "ngen", "tlen", "gene_off", "gene_pham", "seq_off", "residues"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    assert a.names == b.names and a.pham_names == b.pham_names


# ---- 1. loader / packer on well-formed input: equals the Python packer; lazy genomes; FASTA text
sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_tsv_genomes                                         # noqa: E402
small = os.path.join(GOLDEN, "small_input.tsv")
py_genomes = load_tsv_genomes(small)
same_packed(pack.load_tsv_packed(small), pack.pack_genomes(py_genomes))
lazy = pack.load_tsv_genomes(small)
assert [g.name for g in lazy] == [g.name for g in py_genomes]
for g, ref in zip(lazy[:6], py_genomes[:6]):
    assert g.fasta_bytes().decode() == str(ref) and len(g) == len(ref) and g.phams == ref.phams
pk = synth_packed(60, 400, seed=3)
tsv = os.path.join(tmp, "synth.tsv")
write_tsv_packed(pk, tsv)
again = pack.load_tsv_packed(tsv)
assert np.array_equal(again.residues, pk.residues) and np.array_equal(again.bitmap, pk.bitmap)
print("ok loader", flush=True)

# ---- 2. formatter / parser round trips ("%.6f", caller-sized buffers)
rng = np.random.default_rng(1)
for n in (1, 2, 3, 17, 130):
    names = [("g%d" % i) * (1 + i % 7) for i in range(n)]
    vals = np.round(rng.random(n * (n - 1) // 2), 6)
    vals[rng.random(vals.shape[0]) < 0.2] = 0.0
    vals[rng.random(vals.shape[0]) < 0.05] = 1.0
    m = M.SymMatrix.from_condensed(names, vals, is_distance=True)
    for lower in (True, False):
        path = os.path.join(tmp, f"sq_{n}_{lower}.tsv")
        M.matrix_to_squareform(m, path, lower_triangle=lower)
        back = M.matrix_from_squareform(path)
        assert back.nodes == names and np.array_equal(back.to_ndarray(), m.to_ndarray())
    for skip in (False, True):
        path = os.path.join(tmp, f"adj_{n}_{skip}.tsv")
        M.matrix_to_adjacency(m, path, skip_zero=skip)
        if not skip and n > 1:
            back = M.matrix_from_adjacency(path)
            assert np.array_equal(back.extract_submatrix(names).to_ndarray(), m.to_ndarray())
lib = M._text_lib()
import ctypes                                                                 # noqa: E402
# the formatters never write past `cap`, whatever the doubles (a value outside [0, 1] can take 317 bytes instead of 8)
wild = np.array([0.0, 1.0, -0.0, 0.123456789, 1e300, -1e300, 1e-300, np.inf, -np.inf, np.nan, 5e-7, 123456789.5, 2 ** 63, -2.5], dtype=np.float64)
for cap in (1, 8, 100, 321, 322, 323, 400, 1000, 5000, 10000):
    buf = ctypes.create_string_buffer(cap)
    for n in (0, 1, 2, 7, 14):
        got = lib.pcp_format_row(wild.ctypes.data, n, buf, cap)
        assert got == -1 or (0 < got <= cap and buf.raw[got - 1:got] == b"\n")
        if got > 0:
            assert buf.raw[:got].decode() == "\t".join("%.6f" % x for x in wild[:n]) + "\n"
    names_blob = b"alpha" + b"b" * 300 + b"c"
    offs = np.array([0, 5, 305, 306] + [306] * 12, dtype=np.int64)
    got = lib.pcp_format_adjacency(b"src", 3, names_blob, offs.ctypes.data, wild.ctypes.data, 0, 3, 0, buf, cap)
    assert got == -1 or 0 < got <= cap
for text in (b"", b"0.5", b"0.5\t", b"\t", b"abc", b"1e309\t-0.0\tnan", b"0.1\t0.2\t0.3\t0.4", b"1" * 5000, b"0.25\t" * 999 + b"0.5",
             b"\x00\x01\x02", b"0.5\x000.7", b" 0.5 \t 0.7 "):
    for cap in (0, 1, 3, 1000):
        row = np.empty(max(cap, 1), dtype=np.float64)
        got = lib.pcp_parse_row(text, len(text), row.ctypes.data, cap)
        assert -1 <= got <= cap
print("ok text", flush=True)

# ---- 3. fuzz of the loader: malformed files give ValueError (or load), never a crash
good = open(small, "rb").read()[:6000]
good = good[:good.rfind(b"\n") + 1]
cases = {
    "empty": b"", "newline": b"\n", "newlines": b"\n\n\n", "no_final_newline": good.rstrip(b"\n"), "truncated_mid_line": good[:len(good) // 2 - 3],
    "one_column": b"genome_only\n", "four_columns": b"a\tb\tc\td\n", "only_tabs": b"\t\t\n", "tabs_two": b"\t\n",
    "crlf": good.replace(b"\n", b"\r\n"), "cr_only": good.replace(b"\n", b"\r"),
    "nul_in_fields": b"g\x001\tp\x00h\tMK\x00V\nG2\tph\tMKV\n", "nul_line": b"\x00\n", "high_bytes": bytes(range(1, 256)).replace(b"\n", b"") + b"\n",
    "huge_name": b"N" * (1 << 20) + b"\tp1\tMKV\n", "huge_translation": b"g1\tp1\t" + b"A" * (5 << 20) + b"\n",
    "huge_pham": b"g1\t" + b"p" * (1 << 20) + b"\tMKV\n",
    "many_columns": b"\t".join([b"x"] * 100000) + b"\n", "many_empty_columns": b"\t" * 100000 + b"\n",
    "two_col_default_M": b"g1\tp1\ng2\tp1\ng2\tp2\n", "trailing_space": b"g1\tp1\tMKV   \n g2 \t p1 \t MKV\n",
    "empty_translation": b"g1\tp1\t\ng2\tp1\tMK\n", "duplicate_lines": b"g1\tp1\tMKV\n" * 5000, "one_genome": b"g1\tp1\tMKV\n",
    "utf8_names": "gén\tφ1\tMKV\ngén2\tφ1\tMKV\n".encode(), "bad_utf8_names": b"g\xff\xfe\tp\xc3\tMKV\n",
    "long_no_newline": b"g1\tp1\t" + b"W" * 70000,
}
pyrng = random.Random(9)
for k in range(300):                                   # random damage to a valid file
    blob = bytearray(good)
    for _ in range(pyrng.randint(1, 8)):
        kind = pyrng.randrange(5)
        pos = pyrng.randrange(len(blob)) if blob else 0
        if kind == 0 and blob:
            blob[pos] = pyrng.randrange(256)
        elif kind == 1:
            blob[pos:pos] = bytes(pyrng.choice(b"\t\n\r\x00 A") for _ in range(pyrng.randint(1, 40)))
        elif kind == 2 and blob:
            del blob[pos:pos + pyrng.randint(1, 200)]
        elif kind == 3 and blob:
            blob = blob[:pos]
        else:
            blob += bytes(pyrng.randrange(256) for _ in range(pyrng.randint(1, 64)))
    cases[f"mutation_{k}"] = bytes(blob)
loaded = refused = 0
for name, blob in cases.items():
    path = os.path.join(tmp, "fuzz.tsv")
    with open(path, "wb") as fh:
        fh.write(blob)
    try:
        got = pack.load_tsv_packed(path)
        got.validate()
        assert got.n_genomes >= 0 and int(got.seq_off[-1]) == got.residues.shape[0]
        genomes = pack.load_tsv_genomes(path)
        if genomes:
            genomes[0].fasta_bytes(); genomes[-1].fasta_bytes()
        loaded += 1
    except (ValueError, UnicodeDecodeError, MemoryError):
        refused += 1
try:
    pack.load_tsv_packed(os.path.join(tmp, "does_not_exist.tsv"))
    raise SystemExit("a missing file loaded")
except ValueError:
    pass
assert loaded > 50 and refused > 20, (loaded, refused)
print(f"ok fuzz ({loaded} loaded, {refused} refused)", flush=True)

# ---- 4. generator
for n, p, seed in ((1, 10, 1), (2, 64, 2), (41, 65, 3), (300, 5000, None)):
    synth_packed(n, p, seed).validate()
print("ok synth", flush=True)

# ---- 5. the checker: both aligner formulations, the fills, the certificate, the rule sweep
from oracle import oracle as O                                                # noqa: E402
assert "asan" in O._lib_path()
aa = "ACDEFGHIKLMNPQRSTVWY"
for it in range(400):
    alpha = aa[:3] if it % 3 == 0 else aa + "xbz*JU"
    a = "".join(pyrng.choice(alpha) for _ in range(pyrng.randint(1, 70)))
    b = "".join(pyrng.choice(alpha) for _ in range(pyrng.randint(1, 70)))
    tb = O.nw_traceback(a, b)
    sc, ident, diag = O.nw_stats(a, b)
    assert sc == tb.score and ident == tb.comp.count("|") and len(tb.query) == len(a) + len(b) - diag
    score, count, idr, dgr = O.cooptimal(a, b)
    assert score == sc and idr[0] <= ident <= idr[1] and dgr[0] <= diag <= dgr[1] and count >= 1
assert O.nw_stats("M", "M") == (5, 1, 1)
for bad in (("", "A"), ("A", "")):
    try:
        O.nw_stats(*bad)
        raise SystemExit("an empty sequence aligned")
    except ValueError:
        pass
pk = synth_packed(10, 150, seed=5)
for metric in ("gcs", "jc", "pocp", "af", "aai", "peq", "aai_ppos"):
    full = O.fill(pk, metric)
    n = pk.n_genomes
    want = np.array([O.pair(pk, metric, s, t) for s in range(n) for t in range(s + 1, n)])
    assert np.array_equal(full, want), metric
iu = np.triu_indices(pk.n_genomes, 1)
O.tie_sensitivity(pk, iu[0], iu[1])
a_idx, b_idx, _ = O.enumerate_alignments(pk, iu[0], iu[1])
O.cooptimal_batch(pk.residues, pk.seq_off, a_idx, b_idx)
O.nw_batch(pk.residues, pk.seq_off, a_idx, b_idx)
for x in (0.0, 1.0, 0.5, 1 / 640, 5e-7, 0.9999995, 1e-300, 123456.789):
    assert O.round6(x) == round(x, 6)
print("ok oracle", flush=True)
print("SANITIZED RUN COMPLETE", flush=True)
