#!/bin/bash
set -u
OUT=gpurun_out/r04_win64; mkdir -p $OUT
for v in base win64 base win64; do
  echo "== variant '$v'" | tee -a $OUT/ab.txt
  PHAMCLUST_NATIVE_VARIANT=${v#base} python3 tools/bucket_size_bench.py --lens 100,207,420 --rows 16,64 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
  PHAMCLUST_NATIVE_VARIANT=${v#base} python3 tools/quick_bench.py -n 2000 --steps 4 --check 1000 2>&1 | grep -E "step 3|oracle" | cut -c1-120 | tee -a $OUT/ab.txt
done
