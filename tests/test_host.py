"""CPU tests of the host side: Genome / SymMatrix surface, packing, synthetic data,
TSV I/O, the C-ABI export list, and loud failure without the HIP library."""

import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO, golden_file, read_lower_triangle


# ---- Genome ---------------------------------------------------------------------------
def test_genome_surface():
    from phamclust_amd.genome import Genome
    g = Genome("g1")
    g.add("p1", "MKV"); g.add("p2"); g.add("p1", "MKL")
    assert g.phams == {"p1": ["MKV", "MKL"], "p2": ["M"]}
    assert len(g) == 3 and len(g.phams) == 2 and "p1" in g and "zz" not in g
    assert g["p1"] == ["MKV", "MKL"]
    with pytest.raises(KeyError):
        g["zz"]
    with pytest.raises(TypeError):
        g.add(1, "M")
    with pytest.raises(TypeError):
        g.add("p", 5)
    with pytest.raises(TypeError):
        1 in g
    h = Genome("g2"); h.add("p2", "MA"); h.add("p3", "MC")
    assert g & h == {"p2"} == g.intersection(h)
    assert g | h == {"p1", "p2", "p3"} == g.union(h)
    assert g - h == {"p1"} == g.difference(h)
    assert g ^ h == {"p1", "p3"} == g.symmetric_difference(h)
    with pytest.raises(TypeError):
        g & "x"
    assert str(g) == ">name=g1|pham=p1|n=1\nMKV\n>name=g1|pham=p1|n=2\nMKL\n>name=g1|pham=p2|n=1\nM\n"
    assert h < g and g.pop("p2") == ["M"] and g.pop("p2") is None


def test_genome_fasta_roundtrip(tmp_path):
    from phamclust_amd.genome import Genome, GenomeLoadError
    g = Genome("g1"); g.add("p1", "MKV"); g.add("p1", "MKL"); g.add("p9", "MW")
    path = g.save(tmp_path / "g1.fasta")
    h = Genome("g1"); h.load(path)
    assert h.phams == g.phams
    bad = tmp_path / "bad.fasta"
    bad.write_text(">name=x|n=1\nMK\n")
    with pytest.raises(GenomeLoadError):
        Genome("x").load(bad)


# ---- packing --------------------------------------------------------------------------
def test_pack_layout(small_genomes, small_packed):
    pk = small_packed
    assert pk.names == [g.name for g in small_genomes] == sorted(pk.names)
    assert pk.n_phams == len({p for g in small_genomes for p in g.phams})
    for i, g in enumerate(small_genomes):
        assert pk.nph[i] == len(g.phams) and pk.ngen[i] == len(g)
        assert pk.tlen[i] == sum(len(t) for ts in g.phams.values() for t in ts)
        ids = pk.gene_pham[pk.gene_off[i]:pk.gene_off[i + 1]]
        assert (np.diff(ids) >= 0).all()
        row = pk.bitmap[i * pk.words_per_row:(i + 1) * pk.words_per_row]
        assert sum(bin(int(w)).count("1") for w in row) == len(g.phams)
    from phamclust_amd.pack import unpack_genomes
    back = unpack_genomes(pk)
    for g, h in zip(small_genomes, back):
        assert {p: ts for p, ts in g.phams.items()} == {p: ts for p, ts in h.phams.items()}


def test_pack_rejects_empty_and_wide_chars():
    from phamclust_amd.genome import Genome
    from phamclust_amd.pack import pack_genomes
    with pytest.raises(ValueError):
        pack_genomes([])
    g = Genome("g"); g.add("p", "MK中")
    with pytest.raises(ValueError):
        pack_genomes([g])


def test_synth_is_deterministic_and_consistent(native_built):
    from phamclust_amd.pack import pack_genomes, unpack_genomes
    from phamclust_amd.synth import synth_packed
    a, b = synth_packed(50, 700), synth_packed(50, 700)
    for f in ("bitmap", "nph", "ngen", "tlen", "gene_off", "gene_pham", "seq_off", "residues"):
        assert np.array_equal(getattr(a, f), getattr(b, f))
    c = synth_packed(50, 700, seed=1)
    assert not np.array_equal(a.residues[:1000], c.residues[:1000])
    again = pack_genomes(unpack_genomes(a))          # the generator's packing == pack_genomes of its genomes
    for f in ("bitmap", "nph", "ngen", "tlen", "gene_off", "gene_pham", "seq_off", "residues"):
        assert np.array_equal(getattr(a, f), getattr(again, f))
    assert a.names[0] == "synth_000000" and 60 <= a.nph.min() and a.nph.max() <= 140
    lens = np.diff(a.seq_off)
    assert lens.min() >= 1 and lens.max() <= 1500 and 150 < lens.mean() < 280
    assert set(np.unique(a.residues).tolist()) <= set(map(ord, "ACDEFGHIKLMNPQRSTVWY"))


# ---- SymMatrix ------------------------------------------------------------------------
def _toy():
    from phamclust_amd.matrix import SymMatrix
    m = SymMatrix(["b", "a", "c"], is_distance=True)
    for n in "abc":
        m.set_weight(n, n, 0.0)
    m.set_weight("a", "b", 0.1234564)
    m.set_weight("c", "a", 0.5)
    m.set_weight("b", "c", 0.9)
    return m


def test_symmatrix_basic_surface():
    m = _toy()
    assert len(m) == 3 and m.nodes == ["b", "a", "c"] and m.is_distance
    assert m.get_weight("b", "a") == m.get_weight("a", "b") == 0.123456          # rounded on write
    assert list(m) == [("b", "b", 0.0), ("b", "a", 0.123456), ("b", "c", 0.9),
                       ("a", "a", 0.0), ("a", "c", 0.5), ("c", "c", 0.0)]
    assert [r for _, r in m.iterrows()] == [[0.0, 0.123456, 0.9], [0.123456, 0.0, 0.5], [0.9, 0.5, 0.0]]
    assert m["a"] == {"a": 0.0, "b": 0.123456, "c": 0.5} and m["c"] == {"c": 0.0}  # string-ordered keys
    with pytest.raises(ValueError):
        m.set_weight("a", "b", 1.5)
    with pytest.raises(KeyError):
        m.set_weight("a", "zz", 0.5)
    with pytest.raises(KeyError):
        m.get_weight("zz", "a")
    with pytest.raises(TypeError):
        5 in m
    m.lock()
    with pytest.raises(AttributeError):
        m.set_weight("a", "b", 0.2)
    assert m.is_locked()
    m.unlock()
    assert m.diameter == 0.9
    assert np.array_equal(m.to_ndarray(condensed=True), np.array([0.123456, 0.9, 0.5]))
    assert str(m).splitlines()[0] == "3" and str(m).splitlines()[1].startswith("b" + " " * 23 + "\t0.000000\t0.123456")
    from phamclust_amd.matrix import SymMatrix
    assert SymMatrix(["x"]) < m
    unset = SymMatrix(["x", "y"])
    assert unset.get_weight("x", "y") is None


def test_symmatrix_medoid_invert_extract_reorder():
    from phamclust_amd.matrix import SymMatrix
    m = _toy()
    # medoid: mean over N+1 incident weights with the diagonal counted twice, ties by node order
    central = {"b": (0.0 + 0.0 + 0.123456 + 0.9) / 4, "a": (0.123456 + 0.0 + 0.0 + 0.5) / 4, "c": (0.9 + 0.5 + 0.0 + 0.0) / 4}
    assert m.medoid[0] == "a" and m.medoid[1] == pytest.approx(central["a"])
    assert m.anti_medoid[0] == "c"
    sub = m.extract_submatrix(["c", "a"])
    assert sub.nodes == ["c", "a"] and sub.get_weight("a", "c") == 0.5 and sub.is_distance
    with pytest.raises(KeyError):
        m.extract_submatrix(["a", "zz"])
    m.invert()
    assert not m.is_distance and m.get_weight("a", "b") == 0.876544 and m.get_weight("a", "a") == 1.0
    assert m.medoid[0] == "a"                       # similarity: largest mean
    with pytest.raises(ValueError):
        m.diameter
    m.invert()
    m.reorder(["a", "b", "c"])
    assert m.nodes == ["a", "b", "c"] and list(m)[1] == ("a", "b", 0.123456)
    with pytest.raises(ValueError):
        m.reorder(["a", "b"])
    m.reorder()                                     # single-linkage tree order
    assert sorted(m.nodes) == ["a", "b", "c"]
    assert m.nearest_neighbors("a", 0.6) == ["b", "c"]
    m.append_node("d", {"a": 0.2, "b": 0.3, "c": 0.4, "d": 0.0})
    assert len(m) == 4 and m.get_weight("d", "b") == 0.3
    with pytest.raises(KeyError):
        m.append_node("d", {"a": 0.2, "b": 0.3, "c": 0.4, "d": 0.0})
    with pytest.raises(ValueError):
        m.append_node("e", {"a": 0.2, "b": 0.3, "c": 0.4, "d": 0.1, "e": 1.0})
    mean, sd, skew = m.statistics
    assert mean == pytest.approx(np.mean([0.123456, 0.5, 0.9, 0.2, 0.3, 0.4]))
    one = SymMatrix(["x"], True); one.set_weight("x", "x", 0.0)
    assert one.statistics == (0.0, 0.0, 0.0)


def test_from_condensed_and_tsv_roundtrip(tmp_path):
    from phamclust_amd.matrix import (SymMatrix, matrix_from_adjacency, matrix_from_squareform,
                                      matrix_to_adjacency, matrix_to_squareform)
    names, gold, diag = read_lower_triangle(golden_file("jc"))
    m = SymMatrix.from_condensed(names, gold, is_distance=True)
    assert np.array_equal(m.to_ndarray(condensed=True), gold)
    # lower-triangle writer is byte-identical to the file the reference wrote
    out = matrix_to_squareform(m, tmp_path / "lt.tsv", lower_triangle=True)
    assert open(out).read() == open(golden_file("jc")).read()
    back = matrix_from_squareform(out)
    assert back.is_distance and back.nodes == names and np.array_equal(back.to_ndarray(condensed=True), gold)
    full = matrix_to_squareform(m, tmp_path / "sq.tsv")
    assert open(full).readline().rstrip().split("\t") == [str(len(names))] + names
    assert np.array_equal(matrix_from_squareform(full).to_ndarray(condensed=True), gold)
    m.invert()
    adj = matrix_to_adjacency(m, tmp_path / "adj.tsv")
    assert open(adj).read() == open(golden_file("jc", "similarity")).read()
    again = matrix_from_adjacency(adj)
    assert not again.is_distance and np.array_equal(again.to_ndarray(), m.to_ndarray())
    with pytest.raises(ValueError):
        SymMatrix.from_condensed(names, gold[:-1])


def test_matrix_de_novo_generic_callable_and_errors():
    from phamclust_amd.genome import Genome
    from phamclust_amd.matrix import _outside_in_index_iterator, matrix_de_novo
    assert list(_outside_in_index_iterator(5)) == [0, 4, 1, 3, 2]          # reference matrix.py:415-416
    assert list(_outside_in_index_iterator(4)) == [0, 3, 1, 2] and list(_outside_in_index_iterator(1)) == [0]
    with pytest.raises(ValueError):
        matrix_de_novo([], lambda s, t, as_distance=False: 0.0, 1)
    gs = []
    for k in range(4):
        g = Genome(f"g{k}")
        for p in range(k + 1):
            g.add(f"p{p}")
        gs.append(g)

    def overlap(source, target, as_distance=False):       # any user callable: generic per-pair path
        sim = len(source & target) / 4.0
        return round(1.0 - sim, 6) if as_distance else round(sim, 6)

    m = matrix_de_novo(gs, overlap, 2)
    assert m.is_distance and m.get_weight("g0", "g3") == 0.75 and m.get_weight("g2", "g3") == 0.25
    assert all(m.get_weight(g.name, g.name) == 0.0 for g in gs)
    sim = matrix_de_novo(gs, overlap, 1, as_distance=False)
    assert not sim.is_distance and sim.get_weight("g1", "g1") == 1.0 and sim.get_weight("g1", "g2") == 0.5


def _toy_shared_fraction(source, target, as_distance=False):
    """A metric that is none of the six METRICS: matrix_de_novo's generic path (module level: joblib pickles it by name)."""
    union = len(source | target)
    sim = len(source & target) / union if union else 0.0
    return round(1.0 - sim, 6) if as_distance else round(sim, 6)


def test_generic_callable_over_cpus_equals_one_process(native_built):
    """matrix.py:471-493: ANY callable is spread over `cpus` workers.  cpus=2 (joblib, unordered results) must give the
    matrix cpus=1 gives, for plain genomes and for the C loader's lazy genomes (which pickle as plain ones and stay lazy)."""
    import random
    from phamclust_amd.genome import Genome
    from phamclust_amd.matrix import matrix_de_novo
    from phamclust_amd.pack import load_tsv_genomes, packed_behind
    rng = random.Random(5)
    gs = []
    for k in range(37):
        g = Genome(f"g{k:02d}")
        for p in rng.sample(range(60), rng.randint(1, 25)):
            g.add(f"p{p}", "MK" * rng.randint(1, 4))
        gs.append(g)
    one = matrix_de_novo(gs, _toy_shared_fraction, 1)
    two = matrix_de_novo(gs, _toy_shared_fraction, 2)
    assert one.nodes == two.nodes and np.array_equal(one.to_ndarray(), two.to_ndarray())
    assert len({one.get_weight(a.name, b.name) for a in gs for b in gs}) > 20        # not a constant matrix
    lazy = load_tsv_genomes(os.path.join(REPO, "tests", "golden", "small_input.tsv"))
    m1 = matrix_de_novo(lazy, _toy_shared_fraction, 1, as_distance=False)
    assert packed_behind(lazy) is None                          # cpus=1 read .phams in this process: detached
    lazy = load_tsv_genomes(os.path.join(REPO, "tests", "golden", "small_input.tsv"))
    m3 = matrix_de_novo(lazy, _toy_shared_fraction, 3, as_distance=False)
    assert packed_behind(lazy) is not None                      # the workers got plain copies; the originals stayed lazy
    assert np.array_equal(m1.to_ndarray(), m3.to_ndarray())
    # more workers than pairs: clamped like the reference (matrix.py:460-462)
    tiny = matrix_de_novo(gs[:2], _toy_shared_fraction, 64)
    assert tiny.get_weight("g00", "g01") == one.get_weight("g00", "g01")


def test_cli_surface():
    from phamclust_amd import cli
    assert list(cli.METRICS) == ["gcs", "jc", "pocp", "af", "aai", "peq"]
    args = cli.parse_args(["in.tsv", "out"])
    assert args.metric == "peq" and args.nr_thresh == 0.75 and args.clu_thresh == 0.25 and args.sub_thresh == 0.6
    assert args.nr_linkage == "complete" and args.clu_linkage == "average" and args.sub_linkage == "single" and args.k_min == 6
    assert cli.parse_args(["in.tsv", "out", "-m", "jc", "-t", "3"]).metric == "jc"
    assert args.device is None and args.gpus == 1 and cli.parse_args(["in.tsv", "out", "--device", "3"]).device == 3
    with pytest.raises(SystemExit):
        cli.parse_args(["in.tsv", "out", "-m", "nope"])


# ---- C-ABI ------------------------------------------------------------------------------
def test_abi_exports_match_header(native_built):
    """The library loads and exports exactly the functions include/phamclust_hip.h declares
    (no compute calls here: there is no GPU on the CPU box)."""
    from phamclust_amd import hip
    header = open(os.path.join(REPO, "include", "phamclust_hip.h")).read()
    declared = set(re.findall(r"\b(pc_[a-z0-9_]+)\s*\(", header)) - {"pc_ctx"}
    assert declared == set(hip.EXPORTS)
    lib = ctypes.CDLL(hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert hip.load().pc_version() == int(re.search(r"#define\s+PC_VERSION\s+(\d+)", header).group(1)) >= 110
    # the release library carries no fault injection; its -DPC_TEST_HOOKS twin exports the same ABI and says what it is
    assert hip.load().pc_test_hooks() == 0
    twin = ctypes.CDLL(os.path.join(os.path.dirname(hip.LIB_PATH), "libphamclust_hip_hooks.so"))
    for name in declared:
        assert hasattr(twin, name), name
    assert twin.pc_test_hooks() == 1 and twin.pc_version() == hip.load().pc_version()
    assert b"PC_FAKE_OOM_ABOVE" not in open(hip.LIB_PATH, "rb").read()
    assert ctypes.sizeof(hip.PcPacked) == 16 + 8 * 8 and ctypes.sizeof(hip.PcStats) == 5 * 8 + 2 * 4 + 4 * 4 + 2 * 8
    assert "n_chunks" in hip.PcStats().as_dict()


def test_borrowed_array_guards_every_numpy_route():
    """a lent result must refuse ufuncs, reductions and numpy functions too once its loan has ended, not only
    indexing and conversion (host logic: the view class itself, over ordinary memory)."""
    from phamclust_amd import hip

    class Owner:
        pass
    base = np.arange(12, dtype=np.float64)
    lent = hip.BorrowedArray(base.view(), Owner())
    part = lent[2:5]                                          # a slice shares the loan
    assert lent.shape == (12,) and len(part) == 3 and lent.dtype == np.float64 and float(lent[3]) == 3.0
    assert type(np.asarray(lent)) is np.ndarray and np.array_equal(lent, base) and list(part) == [2.0, 3.0, 4.0]
    with pytest.raises(ValueError):
        lent[0] = 0.5
    assert float(np.sum(lent)) == 66.0 and type(lent + 1) is np.ndarray and (lent + 1)[0] == 1.0
    assert type(np.ascontiguousarray(lent)) is np.ndarray and np.concatenate([lent, part]).shape == (15,)
    assert float(part.max()) == 4.0 and float(np.dot(lent, lent)) == float(np.dot(base, base))
    into = np.zeros(12)
    np.add(lent, 1.0, out=into)
    assert into[3] == 4.0
    lent._end_loan()
    for use in (lambda: lent[0], lambda: lent.copy(), lambda: np.asarray(lent), lambda: np.sum(lent), lambda: lent + 1, lambda: lent.max(),
                lambda: np.ascontiguousarray(lent), lambda: np.concatenate([base, lent]), lambda: np.add(base, lent, out=into),
                lambda: part[0], lambda: part * 2.0, lambda: np.mean(part)):
        with pytest.raises(hip.HipLibraryError, match="loan has ended"):
            use()


def test_product_fails_loudly_without_library(monkeypatch, tmp_path):
    from phamclust_amd import hip
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "LIB_PATH", str(tmp_path / "missing.so"))
    with pytest.raises(hip.HipLibraryError):
        hip.load()
    with pytest.raises(hip.HipLibraryError):
        hip.Context(0)


def test_native_variant_selects_a_library_next_to_the_product_one():
    """PHAMCLUST_NATIVE_VARIANT=<name> (tools/build_variant.py: A/B of two builds in one GPU call) loads
    csrc/libphamclust_hip_<name>.so; unset -- or "asan", which names the HOST libraries' sanitized twins -- the product library;
    a variant that was never built fails loudly instead of falling back."""
    code = ("import os, sys; sys.path.insert(0, %r); from phamclust_amd import hip; print(os.path.basename(hip.LIB_PATH));\n"
            "try:\n    hip.load(); print('loaded')\nexcept hip.HipLibraryError: print('refused')") % REPO
    def run(variant):
        env = dict(os.environ)
        env.pop("PHAMCLUST_NATIVE_VARIANT", None)
        if variant is not None:
            env["PHAMCLUST_NATIVE_VARIANT"] = variant
        return subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split()
    assert run(None) == ["libphamclust_hip.so", "loaded"]
    assert run("asan")[0] == "libphamclust_hip.so"
    assert run("hooks")[0] == "libphamclust_hip_hooks.so"
    assert run("no_such_build") == ["libphamclust_hip_no_such_build.so", "refused"]


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under phamclust_amd/ may reference it."""
    pkg = os.path.join(REPO, "phamclust_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".c")):
                text = open(os.path.join(root, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "pc_oracle" not in text, f


def test_graft_entry_build(native_built):
    """The driver's build check (`__graft_entry__.build()`): compiles what is stale, imports the package, and
    agrees with the header about the ABI version."""
    import importlib
    import sys
    sys.path.insert(0, REPO)
    entry = importlib.import_module("__graft_entry__")
    entry.build()


def test_text_io_c_path_is_byte_identical(native_built, tmp_path):
    """The C row formatter / parser behind matrix_to_squareform, matrix_to_adjacency and read_squareform produce the
    bytes (and values) of the Python "%.6f" path, on rounded values and on arbitrary doubles (ties, tiny, 1.0)."""
    import ctypes
    from phamclust_amd import matrix as M
    lib = M._text_lib()
    assert lib is not None
    rng = np.random.default_rng(5)
    special = np.array([0.0, 1.0, 0.5, 1e-7, 4.9999999e-7, 5e-7, 5.0000001e-7, 0.9999995, 0.99999949, 0.1234565, 0.1234575,
                        2.5e-6, 3.5e-6, 123456.7890125, -0.0, -1e-9, -0.25, 1e8, 0.000001, 0.999999])
    values = np.concatenate([special, rng.random(5000), np.round(rng.random(5000), 6), rng.random(200) * 1e-6])
    cap = values.size * 24 + 1024
    buf = ctypes.create_string_buffer(cap)
    size = lib.pcp_format_row(values.ctypes.data, values.size, buf, cap)
    assert lib.pcp_format_row(values.ctypes.data, values.size, buf, 4000) == -1          # a buffer that cannot hold the row is refused, not overrun
    assert buf.raw[:size] == ("\t".join(f"{x:.6f}" for x in values.tolist()) + "\n").encode()
    back = np.empty(values.size)
    assert lib.pcp_parse_row(buf.raw[:size - 1], size - 1, back.ctypes.data, values.size) == values.size
    assert back.tolist() == [float(f"{x:.6f}") for x in values.tolist()]
    # whole files, both code paths
    n = 37
    names = [f"g{'x' * (i % 5)}{i:03d}" for i in range(n)]
    m = M.SymMatrix(names, is_distance=True)
    cond = np.round(rng.random(n * (n - 1) // 2), 6)
    cond[::7] = 0.0
    m = M.SymMatrix.from_condensed(names, cond, is_distance=True)
    outputs = {}
    for label, forced in (("c", lib), ("py", False)):
        M._TEXT_LIB = forced
        try:
            paths = [M.matrix_to_squareform(m, tmp_path / f"{label}_sq.tsv"), M.matrix_to_squareform(m, tmp_path / f"{label}_lt.tsv", lower_triangle=True),
                     M.matrix_to_adjacency(m, tmp_path / f"{label}_adj.tsv"), M.matrix_to_adjacency(m, tmp_path / f"{label}_adj0.tsv", skip_zero=True)]
            outputs[label] = [open(p, "rb").read() for p in paths]
            rows = [(name, row.tolist()) for name, row in M._read_rows(paths[1])]
            assert [r[0] for r in rows] == names and all(len(r[1]) == i + 1 for i, r in enumerate(rows))
            back = M.matrix_from_squareform(paths[1])
            outputs[label] += [rows, back.nodes, back.to_ndarray().tolist() if hasattr(back.to_ndarray(), "tolist") else back.to_ndarray()]
        finally:
            M._TEXT_LIB = lib
    assert outputs["c"] == outputs["py"]


def test_chunk_plan_arithmetic(native_built):
    """The rule by which a fill that exceeds its memory budget is cut into target ranges (pc_chunk_plan: host arithmetic in
    the library, no GPU).  Includes a plan of more than 2^31 alignments -- the size at which r02 refused ("shard the job"):
    every chunk stays below what one plan can index, the chunks tile the targets, and none is empty."""
    from phamclust_amd import hip
    rng = np.random.default_rng(5)
    assert hip.Context.chunk_plan([], 10).tolist() == [0]
    assert hip.Context.chunk_plan([5, 5, 5, 20, 1, 1, 1], 10).tolist() == [0, 2, 3, 4, 7]
    assert hip.Context.chunk_plan([0, 0, 0], 1).tolist() == [0, 3]
    assert hip.Context.chunk_plan([7], 1).tolist() == [0, 1]                   # one target above the budget: its own chunk
    # synth-like ramp: target t has ~2.7 alignments per pair (s, t), N = 60,000 genomes -> 4.9e9 alignments
    n = 60000
    counts = (np.arange(n, dtype=np.float64) * rng.uniform(1.5, 4.0, n)).astype(np.uint64)
    total = int(counts.sum())
    assert total > 2 ** 32
    limit = 2 ** 31 - 2
    cuts = hip.Context.chunk_plan(counts, limit)
    assert cuts[0] == 0 and cuts[-1] == n and (np.diff(cuts) > 0).all()
    sums = np.add.reduceat(counts, cuts[:-1])
    assert int(sums.sum()) == total and int(sums.max()) <= limit
    assert len(sums) <= total // limit + 2 + 1                                 # greedy: at most one chunk more than needed, +1 slack
    # a memory budget: 16 GiB of plan buffers at 56 B per alignment
    per_chunk = (16 << 30) // 56
    cuts = hip.Context.chunk_plan(counts, per_chunk)
    sums = np.add.reduceat(counts, cuts[:-1])
    assert int(sums.max()) <= per_chunk and len(sums) >= total // per_chunk
    # uint64 overflow of the running sum is a cut, not a wrap
    big = np.array([2 ** 63, 2 ** 63, 5], dtype=np.uint64)
    assert hip.Context.chunk_plan(big, 2 ** 64 - 1).tolist() == [0, 1, 3]


def test_gpus_request_is_clamped_to_the_work(native_built, tmp_path, monkeypatch):
    """`--gpus N` starts N ranks only when they can win back what starting them costs (the reference clamps its workers to
    the work as well, matrix.py:460-462).  The estimate is host arithmetic on the loaded genomes, before any GPU call; the
    log line says why the run stayed on one GPU."""
    from phamclust_amd import startup
    from phamclust_amd.synth import synth_packed
    small = synth_packed(120, 1500, seed=4)
    cells = startup.alignment_cells(small)
    from oracle import oracle as O
    _, n_aln, n_cells = O.fill_rows(small, "peq", 0, small.n_genomes)
    assert n_cells <= cells <= 1.2 * n_cells                       # an upper estimate, close (paralog-vs-paralog repeats only)
    n, why = startup.choose_gpus(8, small, "peq")
    assert n == 1 and "running on ONE GPU" in why and "8 GPUs" in why and "process group" in why
    assert startup.choose_gpus(1, small, "peq")[0] == 1
    assert startup.choose_gpus(4, small, "jc")[0] == 1             # set metrics: microseconds of kernel, never worth a launch
    monkeypatch.setenv("PHAMCLUST_LAUNCH_COST_S", "0.000001")
    n, why = startup.choose_gpus(8, small, "peq")
    assert n == 8 and "save up to" in why
    monkeypatch.delenv("PHAMCLUST_LAUNCH_COST_S")
    monkeypatch.setenv("PHAMCLUST_FORCE_GPUS", "1")                # what the launcher sets for the ranks it starts
    assert startup.choose_gpus(8, small, "peq")[0] == 8
    # a fill that does outweigh the start-up: cells scaled up by pretending every gene is 40 x longer
    monkeypatch.delenv("PHAMCLUST_FORCE_GPUS")
    import copy
    big = copy.copy(small)
    big.seq_off = small.seq_off * 40
    assert startup.estimate_fill_seconds(big, "peq") > 1600 * 0.9 * startup.estimate_fill_seconds(small, "peq") - 1.0
    # the in-process route (pc_multi_*) has almost nothing to start: the same request goes through once the fill is worth 0.35 s
    assert startup.multi_gpu_route() == "process"
    monkeypatch.setenv("PHAMCLUST_MULTI", "launcher")
    assert startup.multi_gpu_route() == "launcher"
    monkeypatch.delenv("PHAMCLUST_MULTI")
    assert startup.launch_cost_seconds(8, "process") < 0.1 * startup.launch_cost_seconds(8, "launcher") + 0.1
    n, why = startup.choose_gpus(8, small, "peq", route="process")
    assert n == 1 and "a context and an upload per device" in why          # 120 genomes: milliseconds of fill
    assert startup.choose_gpus(8, big, "peq", route="process")[0] == 8
    line = startup.Timeline()
    line.mark("imports")
    assert line.line().startswith("timing: {") and "imports" in line.as_dict()


def test_pipelined_strip_kernels_publish_after_their_stores():
    """the pipelined form of k_nw_strip hands boundary lines from wave to wave through HBM; the progress word that
    announces them must be stored after a vmcnt(0) wait (and, r05, with release / acquire atomics).  tools/check_pipe_publication.py
    compiles the alignment units to gfx950 assembly and checks every such kernel; a compiler update that reorders or drops the wait
    fails here, on the build machine, not as wrong scores on a GPU."""
    import subprocess
    import sys
    from conftest import REPO
    run = subprocess.run([sys.executable, os.path.join(REPO, "tools", "check_pipe_publication.py")], capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "8 pipelined strip kernels checked: ok" in run.stdout
