#!/usr/bin/env python3
"""How much of the aai / peq output depends on the aligner's co-optimal tie-breaks?

parasail (``nw_trace_diag_16`` + ``get_traceback``, reference metrics.py:160-175) is absent from the reference tree and
from this image, so the rules by which it resolves ties between equally good alignments are recalled,
not pinned.  The optimal SCORE is unaffected by them; the traced path -- hence ``comp.count("|")`` and
``len(query)`` (metrics.py:216-217), hence aai and peq -- is not.  This script measures the exposure: every
combination of the three binary rules (plus, oracle-only, "gap state before DIAG") is evaluated by the oracle over

  * tests/golden/small_input.tsv   (all genome pairs), and
  * synth(2000, 5000)              (a seeded random sample of genome pairs, BASELINE.json configs[2]),

in ONE forward pass per alignment (oracle/pc_oracle.c: pco_tie_sensitivity), and compared with rule 0.

    python tools/tie_sensitivity.py [--pairs 50000] [--out tests/golden/tie_sensitivity.json]

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


def rule_name(rule):
    parts = ["H ties: " + ("INS(E) before DEL(F)" if rule & 1 else "DEL(F) before INS(E)"),
             "E open==extend: " + ("open" if rule & 2 else "extend"),
             "F open==extend: " + ("open" if rule & 4 else "extend")]
    if rule & 8:
        parts.append("gap state before DIAG")
    return "; ".join(parts)


def summarise(aai, peq, counters):
    rows = []
    has_aln = aai[:, 0] > 0
    for r in range(aai.shape[1]):
        d_aai = np.abs(aai[:, r] - aai[:, 0])
        d_peq = np.abs(peq[:, r] - peq[:, 0])
        c = counters[r]
        rows.append({
            "rule": r, "meaning": rule_name(r), "kernel_supports": r < 8,
            "alignments": int(c[0]),
            "alignments_changed_frac": float(c[1]) / max(int(c[0]), 1),
            "n_ident_changed_frac": float(c[2]) / max(int(c[0]), 1),
            "aln_len_changed_frac": float(c[3]) / max(int(c[0]), 1),
            "identity_fraction_changed_frac": float(c[6]) / max(int(c[0]), 1),
            "max_abs_d_n_ident": int(c[4]), "max_abs_d_aln_len": int(c[5]),
            "genome_pairs": int(aai.shape[0]), "genome_pairs_with_alignments": int(has_aln.sum()),
            "aai_moved_gt_1e-6_frac": float((d_aai > 1e-6).mean()),
            "peq_moved_gt_1e-6_frac": float((d_peq > 1e-6).mean()),
            "aai_moved_gt_1e-3_frac": float((d_aai > 1e-3).mean()),
            "peq_moved_gt_1e-3_frac": float((d_peq > 1e-3).mean()),
            "max_abs_d_aai": float(d_aai.max()), "max_abs_d_peq": float(d_peq.max()),
            "mean_abs_d_aai": float(d_aai.mean()), "mean_abs_d_peq": float(d_peq.mean()),
        })
    return rows


ALPHABET = b"ARNDCQEGHILKMFPSTWYVBZX*"


def conventions(O, packed, s_idx, t_idx):
    """The other recalled parasail behaviours, each flipped alone against the defaults: how many
    residues of the data set it can touch at all, and what it moves.  Similarities rounded to 6 places, as above."""
    res = np.asarray(packed.residues)
    upper = np.zeros(256, bool); upper[list(ALPHABET)] = True
    lower = np.zeros(256, bool); lower[[c + 32 for c in ALPHABET if 65 <= c <= 90]] = True
    n_lower = int(lower[res].sum())
    n_other = int((~upper[res] & ~lower[res]).sum())
    base = {m: O.pairs(packed, m, s_idx, t_idx, as_distance=False) for m in ("aai", "peq")}
    out = {"residues": int(res.size), "lower_case_residues": n_lower, "residues_outside_the_24_letters": n_other, "switches": []}

    def measure(name, touches, switched, what):
        row = {"switch": name, "what": what, "residues_it_can_touch": touches}
        if touches == 0:
            row.update({"aai_moved_gt_1e-6_frac": 0.0, "peq_moved_gt_1e-6_frac": 0.0, "max_abs_d_aai": 0.0, "max_abs_d_peq": 0.0,
                        "note": "no residue of this data set is affected: exposure is zero by construction"})
        else:
            with switched:                                   # the checker's defaults are back when the block ends
                for m in ("aai", "peq"):
                    d = np.abs(O.pairs(packed, m, s_idx, t_idx, as_distance=False) - base[m])
                    row[f"{m}_moved_gt_1e-6_frac"] = float((d > 1e-6).mean())
                    row[f"max_abs_d_{m}"] = float(d.max())
        out["switches"].append(row)

    measure("case_sensitive_identity", n_lower, O.compat(case_sensitive=True),
            "item 6: '|' only for byte-equal residues (default: case-insensitive)")
    measure("lower_case_outside_alphabet", n_lower, O.compat(lower_unknown=True),
            "item 7: lower-case letters score as an unknown byte (default: like upper case)")
    measure("unknown_byte_scores_as_X", n_other, O.compat(unknown_row=22),
            "item 7: a byte outside the alphabet takes the X row (default: the * row)")
    measure("gap_of_k_costs_open_plus_k_extend", int(res.size), O.gap(12, 1),
            "item 2: the other affine convention (BLAST's), i.e. the same recurrence with open = 12 (default: 11 + (k-1))")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=50000, help="genome pairs sampled from synth(2000,5000)")
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden", "tie_sensitivity.json"))
    a = ap.parse_args()
    from oracle import oracle as O
    from phamclust_amd.pack import pack_genomes
    from phamclust_amd.scripts.phamclust import load_genomes_from_tsv
    from phamclust_amd.synth import synth_packed

    report = {"generated_by": "tools/tie_sensitivity.py", "baseline_rule": 0,
              "rule_bits": {str(k): v for k, v in O.TIE_RULE_BITS.items()},
              "note": "Scores never change; only the traced path does.  Values are SIMILARITIES rounded to 6 places "
                      "(metrics.py:232, 253 with as_distance=False), compared with rule 0 (SURVEY 8c as recalled).  "
                      "Rules 8-15 put a tying gap state before DIAG, which no parasail kernel is believed to do; they "
                      "are listed to show how much larger the exposure would be.",
              "datasets": []}

    genomes = sorted(load_genomes_from_tsv(os.path.join(REPO, "tests", "golden", "small_input.tsv")), key=lambda g: g.name)
    small = pack_genomes(genomes)
    iu = np.triu_indices(small.n_genomes, 1)
    t0 = time.time()
    aai, peq, counters = O.tie_sensitivity(small, iu[0], iu[1])
    report["datasets"].append({"name": "tests/golden/small_input.tsv", "genomes": small.n_genomes, "pairs": "all",
                               "seconds": round(time.time() - t0, 1), "rules": summarise(aai, peq, counters),
                               "other_recalled_conventions": conventions(O, small, iu[0], iu[1])})
    print(f"small_input: {time.time() - t0:.1f} s", flush=True)

    big = synth_packed(2000, 5000)
    rng = np.random.default_rng(20241218)
    s = rng.integers(0, big.n_genomes, a.pairs * 2)
    t = rng.integers(0, big.n_genomes, a.pairs * 2)
    keep = s != t
    lo, hi = np.minimum(s, t)[keep][:a.pairs], np.maximum(s, t)[keep][:a.pairs]
    t0 = time.time()
    aai, peq, counters = O.tie_sensitivity(big, lo, hi)
    report["datasets"].append({"name": "synth(2000,5000)", "genomes": big.n_genomes,
                               "pairs": f"{lo.size} random pairs, numpy default_rng(20241218)",
                               "seconds": round(time.time() - t0, 1), "rules": summarise(aai, peq, counters),
                               "other_recalled_conventions": conventions(O, big, lo, hi)})
    print(f"synth(2000,5000) sample: {time.time() - t0:.1f} s", flush=True)

    worst = {}
    for ds in report["datasets"]:
        for row in ds["rules"]:
            if 1 <= row["rule"] < 8:
                for key in ("alignments_changed_frac", "aai_moved_gt_1e-6_frac", "peq_moved_gt_1e-6_frac", "max_abs_d_aai", "max_abs_d_peq"):
                    worst[key] = max(worst.get(key, 0.0), row[key])
    report["worst_case_over_rules_1_to_7"] = worst
    with open(a.out, "w") as fh:
        json.dump(report, fh, indent=1)
        fh.write("\n")
    print(json.dumps(worst, indent=1))


if __name__ == "__main__":
    main()
