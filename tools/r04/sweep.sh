#!/bin/bash
# r04: re-sweep of the K4 launch policy now that a fill is ~35 launches instead of ~150
set -u
OUT=gpurun_out/r04_sweep; mkdir -p $OUT
run() {  # label, env...
  local label=$1; shift
  echo "== $label" | tee -a $OUT/ab.txt
  env "$@" python3 -u tools/quick_bench.py -n 2000 --steps 4 2>&1 | grep --line-buffered -E "step 3" | cut -c1-60,230- | tee -a $OUT/ab.txt
  env "$@" python3 -u tools/quick_bench.py -n 5000 --steps 3 2>&1 | grep --line-buffered -E "step 2" | cut -c1-60,230- | tee -a $OUT/ab.txt
  env "$@" python3 -u tools/shard_balance.py 5000 8 peq balanced 2>&1 | grep --line-buffered -E "cells max" | tee -a $OUT/ab.txt
}
run "default"
run "budget 32768" PC_TASK_BUDGET=32768
run "budget 73728" PC_TASK_BUDGET=73728
run "budget 98304" PC_TASK_BUDGET=98304
run "streams 4" PC_ALIGN_STREAMS=4
run "streams 2" PC_ALIGN_STREAMS=2
run "small min 64" PC_SMALL_LAUNCH_MIN=64
run "small min 512" PC_SMALL_LAUNCH_MIN=512
run "order size" PC_ALIGN_ORDER=size
