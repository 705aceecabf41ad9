#!/usr/bin/env python3
"""Build csrc/libphamclust_hip_<name>.so from the current sources with extra compiler flags, for A/B runs in one GPU call
(PHAMCLUST_NATIVE_VARIANT=<name> loads it: phamclust_amd/hip.py).

usage: build_variant.py NAME [--only pc_pairs.hip[,pc_api.hip]] [-DFLAG=VALUE ...]
--only: compile just these units with the flags and link them with the release build's objects of the others (seconds, not a minute).
"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phamclust_amd import build as b


def main(name, flags):
    out_dir = os.path.join(b.CSRC, "variants", name)
    os.makedirs(out_dir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    jobs, objs = [], []
    only = None
    if "--only" in flags:
        i = flags.index("--only")
        only = set(flags[i + 1].split(","))
        flags = flags[:i] + flags[i + 2:]
        b.build_all()                                        # the release objects the variant links against
    for src, obj_name, extra in b.HIP_UNITS:
        if only is not None and src not in only:
            objs.append(os.path.join(b.CSRC, obj_name))
            continue
        obj = os.path.join(out_dir, obj_name)
        cmd = [hipcc] + b.HIPCC_FLAGS + ["-cuid=" + os.path.splitext(obj_name)[0]] + extra + flags + ["-c", os.path.join(b.CSRC, src), "-o", obj]
        jobs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, proc in jobs:
        if proc.wait() != 0:
            raise SystemExit("failed: " + " ".join(cmd))
    lib = os.path.join(b.CSRC, f"libphamclust_hip_{name}.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"])
    print(lib)


if __name__ == "__main__":
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    main(sys.argv[1], sys.argv[2:])
