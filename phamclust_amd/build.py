"""Build the native pieces in-tree: ``python -m phamclust_amd.build``.

* ``csrc/libphamclust_hip.so``  hipcc, gfx950 only (cross-compiles without a GPU)
* ``csrc/libpc_synth.so``       gcc, the synthetic-data generator
* ``csrc/libpc_pack.so``        gcc, the TSV loader/packer
The built files stay next to their sources so that they travel with the tree.

``python -m phamclust_amd.build --asan`` builds the HOST libraries a second time, under ``csrc/asan/``, with
``-fsanitize=address,undefined``; ``PHAMCLUST_NATIVE_VARIANT=asan`` makes the package load those instead.  GPU
AddressSanitizer is not available on the pool: sanitizers cover the CPU code -- the TSV loader / formatter / parser that
read user files and write into caller buffers, and the generator.  tests/test_sanitized.py runs them (LD_PRELOAD=libasan)
over the loader tests and a small fuzz (the test checker has its own sanitized build, outside this package).
"""

import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# (source, object, extra flags): the systolic alignment kernel's tie rules are spread over four translation units
# tools/build_variant.py adds -ffinite-math-only to these units for the builds that use __builtin_fmax on the cell's 64-bit words
# (PC_MAX_BUILTIN / PC_CELL_ORDER: experiment switches of pc_nw_systolic.h); the product build uses asm maxima and needs no flag.
NW_FLAGS = []
HIP_UNITS = [("pc_api.hip", "pc_api.o", []), ("pc_pairs.hip", "pc_pairs.o", []), ("pc_plan.hip", "pc_plan.o", []),
             ("pc_nw.hip", "pc_nw.o", NW_FLAGS),
             ("pc_nw_rules.hip", "pc_nw_r23.o", NW_FLAGS + ["-DPC_RULE_A=2", "-DPC_RULE_B=3"]),
             ("pc_nw_rules.hip", "pc_nw_r45.o", NW_FLAGS + ["-DPC_RULE_A=4", "-DPC_RULE_B=5"]),
             ("pc_nw_rules.hip", "pc_nw_r67.o", NW_FLAGS + ["-DPC_RULE_A=6", "-DPC_RULE_B=7"])]
# the same library once more with pc_api.hip compiled under -DPC_TEST_HOOKS (fault injection: PC_FAKE_OOM_ABOVE); every other
# object is shared.  Loaded only by the test that needs it (PHAMCLUST_NATIVE_VARIANT=hooks).
HOOKS_UNIT = ("pc_api.hip", "pc_api_hooks.o", ["-DPC_TEST_HOOKS"])
HIP_SOURCES = sorted({u[0] for u in HIP_UNITS})
HIP_LIB = os.path.join(CSRC, "libphamclust_hip.so")
HIP_HOOKS_LIB = os.path.join(CSRC, "libphamclust_hip_hooks.so")
SYNTH_LIB = os.path.join(CSRC, "libpc_synth.so")
PACK_LIB = os.path.join(CSRC, "libpc_pack.so")
# Every unit is compiled with -cuid=<its object's name>: hipcc otherwise derives the compilation-unit id from the source file's
# ABSOLUTE path and puts it into the device code object, so the same sources built in another directory (a scratch checkout on
# the GPU box) give other bytes -- and bench.py keys the PMC traffic record (profiles/traffic.json) by the hash of exactly
# those bytes.  (-fuse-cuid=none would do the same but gives every unit the one symbol __hip_cuid_: they no longer link.)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_hip(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, "pc_common.h"), os.path.join(CSRC, "pc_nw_systolic.h"),
               os.path.join(CSRC, "..", "..", "include", "phamclust_hip.h")]
    objs, jobs = [], []
    for src, obj_name, extra in HIP_UNITS + [HOOKS_UNIT]:
        src_path = os.path.join(CSRC, src)
        obj = os.path.join(CSRC, obj_name)
        if force or _stale(obj, [src_path] + headers):
            cmd = [hipcc] + HIPCC_FLAGS + ["-cuid=" + os.path.splitext(obj_name)[0]] + extra + ["-c", src_path, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            jobs.append((cmd, subprocess.Popen(cmd)))          # the translation units compile side by side
        objs.append(obj)                                       # (the systolic kernel: 8 tie rules x 45 kernels, ~40 s per unit)
    failed = [cmd for cmd, proc in jobs if proc.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    hooks_obj = objs.pop()
    for lib, lib_objs in ((HIP_LIB, objs), (HIP_HOOKS_LIB, [hooks_obj] + objs[1:])):
        if force or _stale(lib, lib_objs):
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + lib_objs + ["-ldl"]
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


ASAN_FLAGS = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]


def native_path(name):
    """Where csrc/<name> is loaded from: next to the sources, or the sanitized twin under csrc/asan/ when
    PHAMCLUST_NATIVE_VARIANT=asan (tests/test_sanitized.py)."""
    if os.environ.get("PHAMCLUST_NATIVE_VARIANT") == "asan":
        return os.path.join(CSRC, "asan", name)
    return os.path.join(CSRC, name)


def build_asan(force=False, verbose=False):
    """libpc_pack.so and libpc_synth.so (csrc/asan/) under AddressSanitizer + UBSan."""
    out_dir = os.path.join(CSRC, "asan")
    os.makedirs(out_dir, exist_ok=True)
    jobs = [(os.path.join(out_dir, "libpc_pack.so"), [os.path.join(CSRC, "pc_pack.c")], []),
            (os.path.join(out_dir, "libpc_synth.so"), [os.path.join(CSRC, "pc_synth.c")], [])]
    built = []
    for target, sources, extra in jobs:
        if force or _stale(target, sources):
            cmd = ["gcc"] + ASAN_FLAGS + extra + ["-fPIC", "-shared", "-o", target] + sources + ["-lm"]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        built.append(target)
    return built


def build_all(force=False, verbose=False):
    return build_hip(force, verbose), build_synth(force, verbose), build_pack(force, verbose)


if __name__ == "__main__":
    if "--asan" in sys.argv:
        build_asan(force="--force" in sys.argv, verbose=True)
    else:
        build_all(force="--force" in sys.argv, verbose=True)
