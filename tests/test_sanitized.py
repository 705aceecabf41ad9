"""CPU sanitizer job: the host-side C code -- csrc/pc_pack.c (parses user TSVs, formats rows into caller
buffers), csrc/pc_synth.c, oracle/pc_oracle.c, oracle/pc_cooptimal.c -- built with -fsanitize=address,undefined
(`python -m phamclust_amd.build --asan`, `make -C oracle asan`) and driven by tests/sanitized_driver.py in a subprocess with libasan preloaded.
GPU AddressSanitizer does not exist on the pool; the device side is covered by the parity tests."""

import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_c_code_under_asan_and_ubsan():
    from phamclust_amd import build
    build.build_asan()
    subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("gcc has no libasan.so here")
    env = dict(os.environ, PHAMCLUST_NATIVE_VARIANT="asan", LD_PRELOAD=os.path.realpath(libasan), OMP_NUM_THREADS="4",
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:allocator_may_return_null=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    run = subprocess.run([sys.executable, os.path.join(REPO, "tests", "sanitized_driver.py")], env=env, capture_output=True, text=True, timeout=1500)
    report = run.stdout + run.stderr
    out = os.environ.get("PC_ASAN_REPORT")
    if out:
        with open(out, "w") as fh:
            fh.write(report)
    assert "AddressSanitizer" not in report and "runtime error" not in report, report[-6000:]
    assert run.returncode == 0 and "SANITIZED RUN COMPLETE" in run.stdout, report[-6000:]
    for section in ("loader", "text", "fuzz", "synth", "oracle"):
        assert f"ok {section}" in run.stdout
