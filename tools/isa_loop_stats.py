#!/usr/bin/env python3
"""Instruction mix of the largest loop of one kernel in a hipcc -save-temps .s file.

usage: isa_loop_stats.py file.s kernel_substring
"""
import re
import sys
from collections import Counter


def main(path, needle):
    text = open(path).read().split("\n")
    start = next(i for i, l in enumerate(text) if needle in l and l.rstrip().endswith(needle.split()[0]) is False
                 and re.match(r"^_Z\S+:", l) and needle in l)
    end = next(i for i in range(start, len(text)) if text[i].strip().startswith(".Lfunc_end"))
    lines = []
    for l in text[start + 1:end]:
        l = l.split(";")[0].strip()
        if not l or (l.startswith(".") and not l.startswith(".LBB")):
            continue
        lines.append(l)
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    best = None
    for i, l in enumerate(lines):
        m = re.match(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            span = (labels[m.group(1)], i)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    loop = [l for l in lines[best[0]:best[1] + 1] if not l.endswith(":")]
    c = Counter(l.split()[0] for l in loop)
    print(f"function lines {len(lines)}, largest loop {len(loop)} instructions")
    for k, v in c.most_common(60):
        print(f"{v:5d} {k}")
    print("VALU", sum(v for k, v in c.items() if k.startswith("v_")), "SALU", sum(v for k, v in c.items() if k.startswith("s_")),
          "DS", sum(v for k, v in c.items() if k.startswith("ds_")), "VMEM", sum(v for k, v in c.items() if k.startswith(("global_", "buffer_", "flat_"))))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
