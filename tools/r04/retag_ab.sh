#!/bin/bash
# r04 (second session): one-instruction re-tags in the DP cell (v_or_b32 / v_and_b32 instead of v_and_or_b32), A/B against the
# previous build kept as libphamclust_hip_base.so (PHAMCLUST_NATIVE_VARIANT=base), interleaved in one call (boxes differ by ~1 %)
set -u
OUT=gpurun_out/r04_retag; mkdir -p $OUT
timeout -k 10 700 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "systolic or tie_rule or segment_counts or long_and_ragged or bytes_outside or both_cells or percent_positives or certified or golden or strip or config2" > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc $rc" | tee -a $OUT/ab.txt; tail -3 $OUT/pytest.log | tee -a $OUT/ab.txt
[ $rc -ne 0 ] && exit $rc
run() {  # label, env...
  local label=$1; shift
  echo "== $label" | tee -a $OUT/ab.txt
  env "$@" python3 tools/quick_bench.py -n 2000 --steps 4 --check 2000 2>&1 | grep -E "step 3|oracle" | tee -a $OUT/ab.txt
  env "$@" python3 tools/quick_bench.py -n 5000 --steps 4 2>&1 | grep -E "step [23]" | tee -a $OUT/ab.txt
}
run "base" PHAMCLUST_NATIVE_VARIANT=base
run "one-op retag" PC_DUMMY=1
run "base" PHAMCLUST_NATIVE_VARIANT=base
run "one-op retag" PC_DUMMY=1
