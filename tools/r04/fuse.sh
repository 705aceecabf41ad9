#!/bin/bash
# r04: launch classes of a register tier in ONE launch (k_nw_systolic_tier) against one launch per class
set -u
OUT=gpurun_out/r04_fuse; mkdir -p $OUT
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "systolic or tie_rule or segment_counts or long_and_ragged or bytes_outside or both_cells or percent_positives or certified or chunked_fill or golden or config2 or alignment_sliced" > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee $OUT/ab.txt; tail -3 $OUT/pytest.log | tee -a $OUT/ab.txt
run() {  # label, env...
  local label=$1; shift
  echo "== $label" | tee -a $OUT/ab.txt
  env "$@" python3 -u tools/quick_bench.py -n 2000 --steps 4 --check 2000 2>&1 | grep --line-buffered -E "step 3|oracle" | tee -a $OUT/ab.txt
  env "$@" python3 -u tools/quick_bench.py -n 5000 --steps 3 2>&1 | grep --line-buffered -E "step 2" | tee -a $OUT/ab.txt
  env "$@" python3 -u tools/shard_balance.py 5000 8 peq balanced 2>&1 | grep --line-buffered -E "cells max|assembled" | tee -a $OUT/ab.txt
}
run "one launch per class (PC_FUSE=0), small modes min 192" PC_FUSE=0
run "tier launches, small modes min 192"
run "tier launches, small modes min 32" PC_SMALL_LAUNCH_MIN=32
run "tier launches, small modes off" PC_SMALL_MODES=0
run "tier launches, 4 streams" PC_ALIGN_STREAMS=4
echo "== slice scaling, tier launches" | tee -a $OUT/ab.txt
python3 -u tools/slice_scaling.py 5000 2>&1 | grep --line-buffered -v amdgpu | tee -a $OUT/ab.txt
echo "== long genes: non-strip (2000, 4000) against strip-mined (4500)" | tee -a $OUT/ab.txt
timeout -k 10 200 python3 -u tools/long_gene_bench.py --lens 2000,4000,4500 --variants 0,32,48,64 --check 0 2>&1 | grep --line-buffered -v amdgpu | tee -a $OUT/ab.txt
