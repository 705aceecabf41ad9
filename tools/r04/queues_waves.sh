#!/bin/bash
# r04 experiments: (1) more hardware queues / streams by fill size; (2) 1- and 2-wave workgroups on small buckets
set -u
OUT=gpurun_out/r04_qw; mkdir -p $OUT
run() {  # label, env...
  local label=$1; shift
  echo "== $label" | tee -a $OUT/queues.txt
  env "$@" python3 tools/quick_bench.py -n 2000 --steps 4 2>&1 | grep -E "step 3" | tee -a $OUT/queues.txt
  env "$@" python3 tools/quick_bench.py -n 5000 --steps 3 2>&1 | grep -E "step 2" | tee -a $OUT/queues.txt
  env "$@" python3 tools/shard_balance.py 5000 8 peq balanced 2>&1 | grep -E "cells max" | tee -a $OUT/queues.txt
}
run "4 queues (default), 8 streams"
run "16 queues, 4 streams" GPU_MAX_HW_QUEUES=16 PC_ALIGN_STREAMS=4
run "16 queues, 8 streams" GPU_MAX_HW_QUEUES=16 PC_ALIGN_STREAMS=8
run "16 queues, 16 streams" GPU_MAX_HW_QUEUES=16 PC_ALIGN_STREAMS=16
run "8 queues, 8 streams" GPU_MAX_HW_QUEUES=8 PC_ALIGN_STREAMS=8
for fw in 0 1 2; do for inc in -1 0; do
  echo "== bucket bench PC_FORCE_WAVES=$fw PC_INC16=$inc" | tee -a $OUT/waves.txt
  if [ $inc = -1 ]; then PC_FORCE_WAVES=$fw python3 tools/bucket_size_bench.py --lens 100,207,420 --rows 1,2,4,8,16,32,64 2>&1 | grep -v amdgpu.ids | tee -a $OUT/waves.txt
  else PC_FORCE_WAVES=$fw PC_INC16=$inc python3 tools/bucket_size_bench.py --lens 100,207,420 --rows 1,2,4,8,16,32,64 2>&1 | grep -v amdgpu.ids | tee -a $OUT/waves.txt; fi
done; done
