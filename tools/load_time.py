#!/usr/bin/env python3
"""Input stage at N genomes: the reference-shaped Python path (parse TSV -> Genome objects -> sort -> md5 of the FASTA
text -> pack_genomes) against the C loader path the pipeline uses (pc_pack.c -> lazy genomes -> md5 from the loader's
FASTA text -> the loader's own packed arrays).  Both must agree on the cache key and on every packed array."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from phamclust_amd import build
from phamclust_amd.pack import pack_genomes, packed_behind
from phamclust_amd.scripts.phamclust import _hash_genomes, load_genomes, load_genomes_from_tsv
from phamclust_amd.synth import synth_genomes, write_tsv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
build.build_all()
with tempfile.TemporaryDirectory() as tmp:
    tsv = os.path.join(tmp, "in.tsv")
    write_tsv(synth_genomes(n, 5000), tsv)
    print(f"{n} genomes, {os.path.getsize(tsv) / 1e6:.0f} MB TSV", flush=True)
    t0 = time.perf_counter(); old = sorted(load_genomes_from_tsv(tsv), key=lambda g: g.name); t1 = time.perf_counter()
    md5_old = _hash_genomes(old); t2 = time.perf_counter()
    pk_old = pack_genomes(old); t3 = time.perf_counter()
    print(f"python path : parse+sort {t1 - t0:.2f} s, md5 {t2 - t1:.2f} s, pack {t3 - t2:.2f} s, total {t3 - t0:.2f} s", flush=True)
    t0 = time.perf_counter(); new = load_genomes(tsv); t1 = time.perf_counter()
    md5_new = _hash_genomes(new); t2 = time.perf_counter()
    pk_new = packed_behind(new); t3 = time.perf_counter()
    print(f"C loader    : parse+sort+pack {t1 - t0:.2f} s, md5 {t2 - t1:.2f} s, packed view {t3 - t2:.4f} s, total {t3 - t0:.2f} s", flush=True)
    assert md5_old == md5_new and pk_new is not None
    for f in ("bitmap", "nph", "ngen", "tlen", "gene_off", "gene_pham", "seq_off", "residues"):
        assert np.array_equal(getattr(pk_old, f), getattr(pk_new, f)), f
    print("same cache key, same packed arrays")
