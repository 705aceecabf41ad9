#!/bin/bash
# r04 experiment: K4 launches without the AQL barrier bit (PC_ANY_ORDER=1) against today's ordered launches, by stream count.
set -u
OUT=gpurun_out/r04_anyorder; mkdir -p $OUT
tests/hw/anyorder_probe > $OUT/probe.txt 2>&1; cat $OUT/probe.txt
run() {  # label, env...
  local label=$1; shift
  echo "== $label" | tee -a $OUT/ab.txt
  env "$@" python3 tools/quick_bench.py -n 2000 --steps 4 --check 2000 2>&1 | grep -E "step 3|oracle" | tee -a $OUT/ab.txt
  env "$@" python3 tools/quick_bench.py -n 5000 --steps 3 2>&1 | grep -E "step 2" | tee -a $OUT/ab.txt
  env "$@" python3 tools/shard_balance.py 5000 8 peq balanced 2>&1 | grep -E "cells max|assembled" | tee -a $OUT/ab.txt
}
run "ordered, 8 streams" PC_ANY_ORDER=0
run "any-order, 8 streams" PC_ANY_ORDER=1
run "any-order, 4 streams" PC_ANY_ORDER=1 PC_ALIGN_STREAMS=4
run "any-order, 2 streams" PC_ANY_ORDER=1 PC_ALIGN_STREAMS=2
run "any-order, 1 stream" PC_ANY_ORDER=1 PC_ALIGN_STREAMS=1
echo "== slice scaling, ordered" | tee -a $OUT/ab.txt
PC_ANY_ORDER=0 python3 tools/slice_scaling.py 5000 2>&1 | tee -a $OUT/ab.txt
echo "== slice scaling, any-order 4 streams" | tee -a $OUT/ab.txt
PC_ANY_ORDER=1 PC_ALIGN_STREAMS=4 python3 tools/slice_scaling.py 5000 2>&1 | tee -a $OUT/ab.txt
