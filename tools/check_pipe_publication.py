#!/usr/bin/env python3
"""Build-time check of the pipelined strip kernel's hand-off (k_nw_strip<..., PIPE = true>, csrc/pc_nw_systolic.h).

A wave publishes how far it is by storing a progress word (LDS) AFTER the boundary entries it has written to HBM have left the
wave: the source says so with an `s_waitcnt vmcnt(0)` (tagged `pc_publish` in the assembly) followed by a workgroup-scope release
store.  The memory model orders the two at workgroup scope; that the entries have reached the L2 the reader's agent-scope loads are
served from rests on the explicit wait.  This script compiles the alignment units to gfx950 assembly (no GPU needed) and holds every
PIPE kernel to:  (1) each publication site is there: `s_waitcnt vmcnt(0)` carrying the tag;  (2) between a tagged wait and the
progress store (the next ds_write_b32) no vector-memory STORE is issued;  (3) the reader's polling loop loads the word with
ds_read_b32 inside a loop that contains an s_sleep.  Exit status 1 on any violation.  tests/test_host.py runs it.

usage: check_pipe_publication.py [--keep DIR]
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from phamclust_amd import build as b


def device_asm_start(src, extra, out):
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + b.HIPCC_FLAGS + ["-cuid=check"] + extra + ["--cuda-device-only", "-S", "-o", out, os.path.join(b.CSRC, src)]
    return subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def functions(text):
    """{mangled name: [instruction lines]} of every k_nw_strip instance whose last template argument (PIPE) is true."""
    out = {}
    for m in re.finditer(r"^(_Z10k_nw_stripILi\d+ELi\d+ELb[01]ELb1EE\w*):.*?^\.Lfunc_end\d+:", text, re.S | re.M):
        out[m.group(1)] = [l.strip() for l in m.group(0).split("\n")]
    return out


def check(name, lines):
    problems = []
    sites = [i for i, l in enumerate(lines) if l.startswith("s_waitcnt vmcnt(0)") and "pc_publish" in l]
    if len(sites) < 2:
        problems.append(f"{name}: {len(sites)} tagged publication waits (expected the refill's and the end-of-pass one)")
    for i in sites:
        for j in range(i + 1, min(i + 40, len(lines))):
            op = lines[j].split(" ")[0] if lines[j] else ""
            if op.startswith(("global_store", "buffer_store", "flat_store", "global_atomic", "scratch_store")):
                problems.append(f"{name}: a vector-memory store ({op}) between the tagged wait (line {i}) and the progress store")
                break
            if op == "ds_write_b32":
                break
        else:
            problems.append(f"{name}: no ds_write_b32 within 40 lines of the tagged wait at line {i}")
    if not any(l.startswith("s_sleep") for l in lines):
        problems.append(f"{name}: the polling loop's s_sleep is gone")
    return problems, len(sites)


def main():
    keep = sys.argv[sys.argv.index("--keep") + 1] if "--keep" in sys.argv else None
    tmp = keep or tempfile.mkdtemp(prefix="pc_pipe_check_")
    os.makedirs(tmp, exist_ok=True)
    problems, seen = [], 0
    jobs = [(os.path.join(tmp, obj.replace(".o", ".s")), device_asm_start(src, extra, os.path.join(tmp, obj.replace(".o", ".s"))))
            for src, obj, extra in b.HIP_UNITS if src.startswith("pc_nw")]                 # the units compile side by side
    for out, proc in jobs:
        if proc.wait() != 0:
            print(f"hipcc failed for {out}")
            return 1
        text = open(out).read()
        for name, lines in functions(text).items():
            p, n = check(name, lines)
            problems += p
            seen += 1
    print(f"{seen} pipelined strip kernels checked: {'ok' if not problems else 'PROBLEMS'}")
    for p in problems:
        print("  " + p)
    if seen == 0:
        print("  no k_nw_strip<..., PIPE> instance found in the assembly")
    return 1 if problems or seen == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
