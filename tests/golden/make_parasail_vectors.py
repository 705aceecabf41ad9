#!/usr/bin/env python3
"""Known-answer vectors from REAL parasail, for whoever has it installed (this image does not: no network, no wheel).

    pip install parasail==1.3.4        # the reference pins parasail~=1.3.0 (setup.cfg:38)
    python tests/golden/make_parasail_vectors.py            # writes tests/golden/parasail_vectors.json
    python -m pytest tests/test_oracle.py -k parasail       # which tie rule reproduces every vector?

It calls parasail exactly as the reference does (metrics.py:174-175, 216-217):
``parasail.nw_trace_diag_16(a, b, 11, 1, parasail.blosum62).get_traceback(mch="|", sim="+", neg=" ")`` and keeps what
the reference reads from it -- ``len(traceback.query)`` and ``traceback.comp.count("|")`` (plus the score and the '+'
count).  The sequence set is built to expose every rule the oracle had to recall: tie-heavy 3-letter
alphabets, B / Z / X / * / U and lower case, long indels, very short and very unequal lengths, and a few pairs scoring
beyond the int16 range (the reference ignores parasail's saturation flag).

This script needs nothing from the reference and nothing from this repository: only parasail.
"""
import json
import os
import random

import parasail


def sequences():
    rng = random.Random(20241218)
    aa = "ACDEFGHIKLMNPQRSTVWY"
    pairs = []

    def rnd(alpha, lo, hi):
        return "".join(rng.choice(alpha) for _ in range(rng.randint(lo, hi)))

    for _ in range(300):                                   # co-optimal ties everywhere
        alpha = rng.choice(["AGS", "LIV", "DE", "KR", aa[:4]])
        pairs.append((rnd(alpha, 1, 60), rnd(alpha, 1, 60)))
    for _ in range(200):                                   # homologs: substitutions + indels of a common ancestor
        anc = rnd(aa, 20, 400)
        def mutate(s):
            out = []
            for ch in s:
                r = rng.random()
                if r < 0.03:
                    continue
                if r < 0.06:
                    out.append(rnd(aa, 1, 8))
                out.append(rng.choice(aa) if rng.random() < 0.25 else ch)
            return "".join(out) or "M"
        pairs.append((mutate(anc), mutate(anc)))
    for _ in range(60):                                    # ambiguity codes, stop, selenocysteine, lower case, junk bytes
        alpha = aa + "BZX*U" + "acdefghik" + "J0-"
        pairs.append((rnd(alpha, 1, 80), rnd(alpha, 1, 80)))
    for _ in range(40):                                    # very unequal lengths, long terminal gaps
        pairs.append((rnd(aa, 1, 6), rnd(aa, 80, 300)))
        pairs.append((rnd(aa, 80, 300), rnd(aa, 1, 6)))
    w = "W" * 3200                                         # 3200 x 11 = 35,200 > 32,767: int16 saturation
    pairs += [(w, w), (w, w[:-5] + "AAAAA"), ("C" * 4000, "C" * 3900)]
    pairs += [("M", "M"), ("M", "A"), ("MKV", "MKV"), ("A", "AAAA"), ("AAAA", "A")]
    return pairs


def main():
    out = []
    for a, b in sequences():
        res = parasail.nw_trace_diag_16(a, b, 11, 1, parasail.blosum62)
        tb = res.get_traceback(mch="|", sim="+", neg=" ")
        out.append({"a": a, "b": b, "score": int(res.score), "aln_len": len(tb.query), "n_ident": tb.comp.count("|"),
                    "n_pos": tb.comp.count("+"), "saturated": bool(getattr(res, "saturated", False))})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "parasail_vectors.json")
    with open(path, "w") as fh:
        json.dump({"parasail_version": getattr(parasail, "__version__", "?"), "call": "nw_trace_diag_16(a, b, 11, 1, blosum62)",
                   "vectors": out}, fh)
    print(f"wrote {len(out)} vectors to {path}")


if __name__ == "__main__":
    main()
