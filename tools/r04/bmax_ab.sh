#!/bin/bash
# r04 (second session): the cell's maxima as compiler v_max_f64 (PC_MAX_BUILTIN: no s_nop after inline asm) against asm maxima (libphamclust_hip_asmmax.so)
set -u
OUT=gpurun_out/r04_bmax; mkdir -p $OUT
timeout -k 10 700 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "systolic or tie_rule or segment_counts or long_and_ragged or bytes_outside or both_cells or percent_positives or certified or golden or strip or config2" > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc $rc" | tee -a $OUT/ab.txt; tail -3 $OUT/pytest.log | tee -a $OUT/ab.txt
[ $rc -ne 0 ] && exit $rc
run() {  # label, env...
  local label=$1; shift
  echo "== $label" | tee -a $OUT/ab.txt
  env "$@" python3 tools/quick_bench.py -n 2000 --steps 4 --check 2000 2>&1 | grep -E "step 3|oracle" | tee -a $OUT/ab.txt
  env "$@" python3 tools/quick_bench.py -n 5000 --steps 4 2>&1 | grep -E "step [23]" | tee -a $OUT/ab.txt
}
run "asm max" PHAMCLUST_NATIVE_VARIANT=asmmax
run "builtin max" PC_DUMMY=1
run "asm max" PHAMCLUST_NATIVE_VARIANT=asmmax
run "builtin max" PC_DUMMY=1
for v in asmmax ""; do
  echo "== bucket bench variant '$v'" | tee -a $OUT/ab.txt
  PHAMCLUST_NATIVE_VARIANT=$v python3 tools/bucket_size_bench.py --lens 100,207,420,800 --rows 1,2,4,8,16,64 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
done
