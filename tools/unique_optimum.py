#!/usr/bin/env python3
"""Which part of aai / peq is pinned by mathematics rather than by recalled parasail tie rules?

For every alignment the reference would run (metrics.py:203-217) on

  * tests/golden/small_input.tsv   (all genome pairs), and
  * synth(2000, 5000)              (the seeded 50,000-pair sample tools/tie_sensitivity.py uses),

oracle/pc_cooptimal.c counts ALL optimal global alignments (three-state DP, independent of the oracle's aligner) and the
range of (n_ident, n_diag) over them.  An alignment is CERTIFIED when that range is one point: every correct
Needleman-Wunsch, parasail included, must then report exactly these statistics.  A genome pair is fully certified when
all its alignments are; its aai / peq then follow from the reference's own Python (pinned by tests/golden fixtures) and
nothing recalled.  For the rest a rule-independent interval of aai is derived from the ranges.

    python tools/unique_optimum.py [--pairs 50000] [--out tests/golden/unique_optimum.json]

Test infrastructure: imports oracle/, never shipped product code.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def aai_interval(la_lb, rng, anchor_start):
    """Rule-independent interval of one genome pair's aai.  la_lb[k] = la + lb, rng[k] = (id_lo, id_hi, dg_lo, dg_hi) per
    alignment in the reference's loop order; anchor_start: indices where a new anchor gene's candidate list begins.
    Per anchor gene the reference keeps the candidate with the best ident/len (metrics.py:221-223).  Any candidate whose
    best possible ratio reaches the largest guaranteed ratio may be the one kept; numerator and denominator are bounded
    over those."""
    num_lo = num_hi = den_lo = den_hi = 0
    bounds = list(anchor_start) + [len(la_lb)]
    for x, y in zip(bounds[:-1], bounds[1:]):
        id_lo, id_hi = rng[x:y, 0].astype(np.int64), rng[x:y, 1].astype(np.int64)
        len_hi, len_lo = la_lb[x:y] - rng[x:y, 2], la_lb[x:y] - rng[x:y, 3]
        r_lo, r_hi = id_lo / len_hi, id_hi / len_lo
        cand = r_hi >= r_lo.max()
        num_lo += id_lo[cand].min(); num_hi += id_hi[cand].max()
        den_lo += len_lo[cand].min(); den_hi += len_hi[cand].max()
    return num_lo / den_hi, min(1.0, num_hi / den_lo)


def certify(O, packed, s_idx, t_idx, name, pairs_desc):
    t0 = time.time()
    a, b, q = O.enumerate_alignments(packed, s_idx, t_idx)
    score, count, rng = O.cooptimal_batch(packed.residues, packed.seq_off, a, b)
    osc, oid, odg = O.nw_batch(packed.residues, packed.seq_off, a, b)
    assert np.array_equal(score, osc), "the oracle's aligner and the certificate disagree on an optimal score"
    assert ((rng[:, 0] <= oid) & (oid <= rng[:, 1]) & (rng[:, 2] <= odg) & (odg <= rng[:, 3])).all(), \
        "the oracle's aligner returned statistics no optimal alignment has"
    cert = (rng[:, 0] == rng[:, 1]) & (rng[:, 2] == rng[:, 3])
    unique = count == 1
    n_pairs = int(np.asarray(s_idx).shape[0])
    with_aln = np.unique(q)
    bad_pairs = np.unique(q[~cert])
    lens = (packed.seq_off[1:] - packed.seq_off[:-1]).astype(np.int64)
    la_lb = lens[a] + lens[b]
    # rule-independent aai interval of the pairs that hold an uncertified alignment
    order_start = np.flatnonzero(np.r_[True, q[1:] != q[:-1]])
    pair_begin = dict(zip(q[order_start].tolist(), order_start.tolist()))
    pair_end = dict(zip(q[order_start].tolist(), np.r_[order_start[1:], q.shape[0]].tolist()))
    width, width_peq = [], []
    for p in bad_pairs.tolist():
        x, y = pair_begin[p], pair_end[p]
        anchors = np.flatnonzero(np.r_[True, a[x + 1:y] != a[x:y - 1]])
        lo, hi = aai_interval(la_lb[x:y], rng[x:y], anchors)
        af = 1.0 - O.pair(packed, "af", int(s_idx[p]), int(t_idx[p]), as_distance=True)
        width.append(hi - lo); width_peq.append(af * (hi - lo))
    width = np.asarray(width) if width else np.zeros(1)
    width_peq = np.asarray(width_peq) if width_peq else np.zeros(1)
    cells = lens[a] * lens[b]
    sat = int((count == np.uint64(O.COUNT_SATURATED)).sum())
    out = {
        "name": name, "genomes": packed.n_genomes, "pairs": pairs_desc, "genome_pairs": n_pairs,
        "genome_pairs_with_alignments": int(with_aln.shape[0]),
        "alignments": int(a.shape[0]), "dp_cells": int(cells.sum()),
        "max_sequence_length": int(max(lens[a].max(), lens[b].max())) if a.shape[0] else 0,
        "alignments_unique_optimum": int(unique.sum()),
        "alignments_unique_optimum_frac": float(unique.mean()) if a.shape[0] else 1.0,
        "alignments_certified": int(cert.sum()),
        "alignments_certified_frac": float(cert.mean()) if a.shape[0] else 1.0,
        "alignments_co_optimal_count_saturated_u64": sat,
        "genome_pairs_fully_certified": int(with_aln.shape[0] - bad_pairs.shape[0]),
        "genome_pairs_fully_certified_frac_of_pairs_with_alignments": float(1.0 - bad_pairs.shape[0] / max(with_aln.shape[0], 1)),
        "genome_pairs_pinned_frac_of_all_pairs": float(1.0 - bad_pairs.shape[0] / max(n_pairs, 1)),
        "uncertified_alignments": {
            "count": int((~cert).sum()),
            "max_n_ident_spread": int((rng[:, 1] - rng[:, 0]).max()) if a.shape[0] else 0,
            "max_n_diag_spread": int((rng[:, 3] - rng[:, 2]).max()) if a.shape[0] else 0,
        },
        "residual_genome_pairs": {
            "count": int(bad_pairs.shape[0]),
            "rule_independent_aai_interval_width": {"max": float(width.max()), "mean": float(width.mean()),
                                                    "p99": float(np.quantile(width, 0.99))},
            "rule_independent_peq_interval_width": {"max": float(width_peq.max()), "mean": float(width_peq.mean()),
                                                    "p99": float(np.quantile(width_peq, 0.99))},
        },
        "seconds": round(time.time() - t0, 1),
    }
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=50000, help="genome pairs sampled from synth(2000,5000)")
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden", "unique_optimum.json"))
    a = ap.parse_args()
    from oracle import oracle as O
    from phamclust_amd.pack import pack_genomes
    from phamclust_amd.scripts.phamclust import load_genomes_from_tsv
    from phamclust_amd.synth import synth_packed

    report = {"generated_by": "tools/unique_optimum.py",
              "what": "oracle/pc_cooptimal.c: number of optimal global alignments (BLOSUM62, gap of k residues = 11 + (k-1)) and the range of "
                      "(n_ident, n_diag) over ALL of them, per alignment the reference runs (metrics.py:203-217).  Certified = the range "
                      "is one point: every correct Needleman-Wunsch reports these statistics, whatever its tie-breaking.",
              "assumes": "only what 'parasail.nw_trace_diag_16(a, b, 11, 1, blosum62)' means by definition: an OPTIMAL global alignment under "
                         "BLOSUM62 with gap cost open + (k-1)*extend, '|' = identical residues.  Nothing about tie-breaking.",
              "datasets": []}
    genomes = sorted(load_genomes_from_tsv(os.path.join(REPO, "tests", "golden", "small_input.tsv")), key=lambda g: g.name)
    small = pack_genomes(genomes)
    iu = np.triu_indices(small.n_genomes, 1)
    report["datasets"].append(certify(O, small, iu[0], iu[1], "tests/golden/small_input.tsv", "all"))
    print(json.dumps(report["datasets"][-1], indent=1), flush=True)

    big = synth_packed(2000, 5000)
    rng = np.random.default_rng(20241218)                      # the sample tools/tie_sensitivity.py draws
    s = rng.integers(0, big.n_genomes, a.pairs * 2)
    t = rng.integers(0, big.n_genomes, a.pairs * 2)
    keep = s != t
    lo, hi = np.minimum(s, t)[keep][:a.pairs], np.maximum(s, t)[keep][:a.pairs]
    report["datasets"].append(certify(O, big, lo, hi, "synth(2000,5000)", f"{lo.size} random pairs, numpy default_rng(20241218)"))
    print(json.dumps(report["datasets"][-1], indent=1), flush=True)
    with open(a.out, "w") as fh:
        json.dump(report, fh, indent=1)
        fh.write("\n")


if __name__ == "__main__":
    main()
