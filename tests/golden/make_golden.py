#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the LIVE reference.

Run in the build container only (the reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What runs here is the reference's own code, unmodified, imported from
/root/reference/src: ``phamclust.matrix.matrix_de_novo`` over ``phamclust.metrics``
functions, written out with the reference's own ``matrix_to_squareform`` /
``matrix_to_adjacency``.

``phamclust.metrics`` imports the third-party ``parasail`` (setup.cfg: parasail~=1.3.0),
which is absent from this image and cannot be installed offline.  A stand-in module is
registered so that the import succeeds:

* gcs / jc / pocp / af never touch it -> those fixtures are pure reference output;
* aai / peq call ``nw_trace_diag_16(...).get_traceback(...)``: the stand-in answers with
  the ORACLE's aligner (oracle/pc_oracle.c).  Those two fixtures therefore pin the
  reference's loop structure, anchor rule, best-match rule, weighted mean and rounding
  chain (metrics.py:178-253) -- NOT parasail's co-optimal tie-breaking, which stays
  unpinned.  Files carry the suffix ``.oracle_nw`` to keep the two classes apart.

Outputs: input TSV, per-metric lower-triangle distance matrices (the pipeline's cache
format, scripts/phamclust.py:262), similarity adjacency lists, and a manifest.
"""

import json
import os
import pathlib
import sys
import types

HERE = pathlib.Path(__file__).resolve().parent
REPO = HERE.parent.parent
REFERENCE_SRC = "/root/reference/src"

sys.path.insert(0, str(REPO))
from oracle import oracle as O                      # noqa: E402
from phamclust_amd.build import build_synth          # noqa: E402
from phamclust_amd.genome import Genome as OurGenome  # noqa: E402
from phamclust_amd.synth import synth_genomes, write_tsv  # noqa: E402


def install_parasail_stand_in():
    mod = types.ModuleType("parasail")
    mod.blosum62 = "blosum62"

    class _Result:
        def __init__(self, a, b):
            self._a, self._b = a, b

        def get_traceback(self, mch="|", sim="+", neg=" "):
            tb = O.nw_traceback(self._a, self._b)
            comp = tb.comp.replace("|", "\x00").replace("+", "\x01").replace(" ", neg)
            tb.comp = comp.replace("\x00", mch).replace("\x01", sim)
            return tb

    def nw_trace_diag_16(seq_a, seq_b, gap_open, gap_extend, matrix):
        assert (gap_open, gap_extend, matrix) == (11, 1, "blosum62")
        return _Result(seq_a, seq_b)

    mod.nw_trace_diag_16 = nw_trace_diag_16
    sys.modules["parasail"] = mod


def handmade():
    """Edge cases the reference's semantics exercise (names sort after synth_*)."""
    g = []
    a = OurGenome("zz_a_paralogs")                       # 2 copies vs 1 copy vs 3 copies of pham_x
    a.add("pham_x", "MKTAYIAKQRQISFVKSHFSRQLEERLGLIEVQ")
    a.add("pham_x", "MKTAYIAKQRQISFVKSHFSRQ")
    a.add("pham_y", "MSDNELLKKA")
    g.append(a)
    b = OurGenome("zz_b_single")
    b.add("pham_x", "MKTAYIAKQRQLSFVKSHFSRQLEERLGLIEVQ")
    b.add("pham_z", "MAAAAKKKK")
    g.append(b)
    c = OurGenome("zz_c_three")
    c.add("pham_x", "MKTAYIAKQRQISFVKSHFSRQLEERLGLIEVQ")
    c.add("pham_x", "MKSAYIAKQRQ")
    c.add("pham_x", "MKTAYIAKQRQISFVKSHFSRQLEERLGLIEVQGGG")
    c.add("pham_y", "MSDNELLKKAW")
    g.append(c)
    d = OurGenome("zz_d_disjoint")                       # shares nothing with anyone
    d.add("pham_only_d", "MWWWWW")
    g.append(d)
    e = OurGenome("zz_e_two_column")                     # 2-column input rows -> translation "M"
    e.add("pham_x")
    e.add("pham_y")
    e.add("pham_q")
    g.append(e)
    f = OurGenome("zz_f_odd_residues")                   # lower case, X, B, Z, *, U (non-alphabet)
    f.add("pham_x", "mktayiakqrqisfvkshfsrqleerlglievq")
    f.add("pham_y", "MSDXELLKBZ*U")
    f.add("pham_q", "M")
    g.append(f)
    h = OurGenome("zz_g_tie_equal_best")                 # two identical candidates: ties -> last
    h.add("pham_y", "MSDNELLKKA")
    h.add("pham_y", "MSDNELLKKA")
    g.append(h)
    return g


def main():
    if not os.path.isdir(REFERENCE_SRC):
        sys.exit("the reference is not mounted here; fixtures are generated in the build container only")
    build_synth()
    O.build()
    genomes = synth_genomes(16, 400, seed=7) + handmade()
    genomes.sort(key=lambda x: x.name)                   # scripts/phamclust.py:221
    tsv = HERE / "small_input.tsv"
    write_tsv(genomes, tsv)
    # 2-column rows for the genome whose translations are all "M"
    lines = tsv.read_text().splitlines()
    lines = [("\t".join(ln.split("\t")[:2]) if ln.startswith("zz_e_two_column\t") else ln) for ln in lines]
    tsv.write_text("\n".join(lines) + "\n")

    install_parasail_stand_in()
    sys.path.insert(0, REFERENCE_SRC)
    sys.dont_write_bytecode = True
    from phamclust.cli import METRICS                    # the reference's selector
    from phamclust.matrix import matrix_de_novo, matrix_to_adjacency, matrix_to_squareform
    from phamclust.scripts.phamclust import load_genomes_from_tsv

    ref_genomes = load_genomes_from_tsv(tsv)             # the reference's own loader
    ref_genomes.sort(key=lambda x: x.name)
    manifest = {"generator": "tests/golden/make_golden.py", "reference": "chg60/phamclust @ 2024-12-18 (v1.3.3)",
                "python": sys.version.split()[0], "n_genomes": len(ref_genomes), "input": tsv.name, "files": {}}
    for metric, func in METRICS.items():
        tag = "" if metric in ("gcs", "jc", "pocp", "af") else ".oracle_nw"
        dist = matrix_de_novo(ref_genomes, func, 1)                              # as the pipeline calls it
        f_dist = HERE / f"{metric}_distance_matrix{tag}.tsv"
        matrix_to_squareform(dist, f_dist, lower_triangle=True)                  # scripts/phamclust.py:262
        sim = matrix_de_novo(ref_genomes, func, 1, as_distance=False)
        f_sim = HERE / f"pairwise_{metric}_similarities{tag}.tsv"
        matrix_to_adjacency(sim, f_sim, skip_zero=False)
        manifest["files"][metric] = {"distance_lower_triangle": f_dist.name, "similarity_adjacency": f_sim.name,
                                     "aligner": "none" if not tag else "oracle (parasail absent)"}
        print(metric, "written")
    # ---- the reference's whole pipeline (stage 3-6: clustering, sub-clustering, outputs) for one set metric ----
    # Heatmap rendering needs kaleido (absent here), so the one name `draw_heatmap` is replaced in the imported
    # pipeline module's namespace for this run; heatmaps are not part of the fixture.  Everything else
    # (SymMatrix, clustering.py, file writers, the pipeline function itself) is the reference's own code.
    import phamclust.scripts.phamclust as ref_pipeline_module
    ref_pipeline_module.draw_heatmap = lambda *a, **k: None
    import shutil, tempfile
    from phamclust.scripts.phamclust import phamclust as ref_pipeline
    pipe_dir = HERE / "pipeline_jc"
    if pipe_dir.exists():
        shutil.rmtree(pipe_dir)
    pipe_dir.mkdir()
    with tempfile.TemporaryDirectory() as tmp:
        out = pathlib.Path(tmp) / "out"; out.mkdir()
        ref_pipeline(infile=tsv, outdir=out, is_genome_dir=False, metric="jc", nr_distance=round(1.0 - 0.75, 6),
                     nr_linkage="complete", clu_distance=round(1.0 - 0.25, 6), clu_linkage="average",
                     sub_distance=round(1.0 - 0.6, 6), sub_linkage="single", k_min=3, no_sub=False,
                     colors=["red", "yellow", "green"], midpoint=0.5, cpus=1, rm_tmp=False, debug=False)
        tree = {}
        for path in sorted(out.rglob("*")):
            rel = path.relative_to(out).as_posix()
            if path.is_file() and ".tmp/01_genomes" not in rel and not rel.endswith((".svg", ".html", ".log")):
                tree[rel] = path.read_text() if path.suffix == ".tsv" else None
        md5 = [p.name for p in out.iterdir() if p.name.endswith(".tmp")][0]
        (pipe_dir / "tree.json").write_text(json.dumps({"md5_tmp_dir": md5, "k_min": 3, "files": tree}, indent=0) + "\n")
    manifest["pipeline"] = {"metric": "jc", "fixture": "pipeline_jc/tree.json", "note": "reference scripts/phamclust.py:phamclust() run "
                            "with draw_heatmap disabled (kaleido absent); .tsv contents and the file tree are recorded"}
    (HERE / "manifest.json").write_text(json.dumps(manifest, indent=1) + "\n")


if __name__ == "__main__":
    main()
