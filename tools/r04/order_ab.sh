#!/bin/bash
# r04 (second session): instruction ORDER and s_nop of the profile cell (PC_CELL_ORDER 1-4, libphamclust_hip_ord<k>.so) against the
# hand-placed asm blocks (the default build); kernel rate on uniform genes, then the N = 2,000 fill
set -u
OUT=gpurun_out/r04_order; mkdir -p $OUT
for v in ${ORDERS:-"" ord1 ord2 ord3 ord4 ""}; do
  echo "== variant '$v'" | tee -a $OUT/ab.txt
  PHAMCLUST_NATIVE_VARIANT=${v#base} python3 tools/bucket_size_bench.py --lens 207,420 --rows 16,64,208 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
  PHAMCLUST_NATIVE_VARIANT=${v#base} python3 tools/quick_bench.py -n 2000 --steps 4 --check 1000 2>&1 | grep -E "step 3|oracle" | cut -c1-120 | tee -a $OUT/ab.txt
done
