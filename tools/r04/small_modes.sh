#!/bin/bash
# r04: one- / two-wave launch modes for small tasks (pc_nw_task_mode), folded into their neighbours below PC_SMALL_LAUNCH_MIN tasks
set -u
OUT=gpurun_out/r04_modes; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "systolic or tie_rule or segment_counts or long_and_ragged or bytes_outside or both_cells or percent_positives or certified or chunked_fill or golden or sparse64_chunked or borrowed or config2" > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/ab.txt; tail -3 $OUT/pytest.log | tee -a $OUT/ab.txt
run() {  # label, env...
  local label=$1; shift
  echo "== $label" | tee -a $OUT/ab.txt
  env "$@" python3 tools/quick_bench.py -n 2000 --steps 4 --check 2000 2>&1 | grep -E "step 3|oracle" | tee -a $OUT/ab.txt
  env "$@" python3 tools/quick_bench.py -n 5000 --steps 3 2>&1 | grep -E "step 2" | tee -a $OUT/ab.txt
  env "$@" python3 tools/shard_balance.py 5000 8 peq balanced 2>&1 | grep -E "cells max|assembled" | tee -a $OUT/ab.txt
}
run "modes off" PC_SMALL_MODES=0
run "modes on, min 192 (default)"
run "modes on, min 32" PC_SMALL_LAUNCH_MIN=32
run "modes on, min 1024" PC_SMALL_LAUNCH_MIN=1024
for sm in 0 1; do
  echo "== bucket bench PC_SMALL_MODES=$sm" | tee -a $OUT/ab.txt
  PC_SMALL_MODES=$sm python3 tools/bucket_size_bench.py --lens 100,207,420,800 --rows 1,2,3,4,6,8,12,16,24,32,64 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
done
