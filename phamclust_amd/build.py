"""Build the native pieces in-tree: ``python -m phamclust_amd.build``.

* ``csrc/libphamclust_hip.so``  hipcc, gfx950 only (cross-compiles without a GPU)
* ``csrc/libpc_synth.so``       gcc, the synthetic-data generator
* ``csrc/libpc_pack.so``        gcc, the TSV loader/packer
The built files stay next to their sources so that they travel with the tree.
"""

import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
HIP_SOURCES = ["pc_api.hip", "pc_pairs.hip", "pc_plan.hip", "pc_nw.hip"]
HIP_LIB = os.path.join(CSRC, "libphamclust_hip.so")
SYNTH_LIB = os.path.join(CSRC, "libpc_synth.so")
PACK_LIB = os.path.join(CSRC, "libpc_pack.so")
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_hip(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, "pc_common.h"),
               os.path.join(CSRC, "..", "..", "include", "phamclust_hip.h")]
    objs, jobs = [], []
    for src in HIP_SOURCES:
        src_path = os.path.join(CSRC, src)
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        if force or _stale(obj, [src_path] + headers):
            cmd = [hipcc] + HIPCC_FLAGS + ["-c", src_path, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            jobs.append((cmd, subprocess.Popen(cmd)))          # the translation units compile side by side
        objs.append(obj)                                       # (pc_nw.hip: 24 widths x 8 tie rules, ~70 s)
    failed = [cmd for cmd, proc in jobs if proc.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    if force or _stale(HIP_LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", HIP_LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return HIP_LIB


def build_synth(force=False, verbose=False):
    src = os.path.join(CSRC, "pc_synth.c")
    if force or _stale(SYNTH_LIB, [src]):
        cmd = ["gcc", "-O2", "-fPIC", "-shared", "-o", SYNTH_LIB, src, "-lm"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return SYNTH_LIB


def build_pack(force=False, verbose=False):
    src = os.path.join(CSRC, "pc_pack.c")
    if force or _stale(PACK_LIB, [src]):
        cmd = ["gcc", "-O2", "-fPIC", "-shared", "-o", PACK_LIB, src, "-lm"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return PACK_LIB


def build_all(force=False, verbose=False):
    return build_hip(force, verbose), build_synth(force, verbose), build_pack(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
