#!/bin/bash
# The r05 set-metric measurements behind profiles/r05/experiments/sparse_col.txt, in one GPU-box call:
#   gpurun --timeout 900 -- 'bash tools/r05/set_kernels.sh'
# (1) parity of every kernel family, forced; (2) jc by source tiles per unit (PC_COL_SEG, an environment knob of the launcher read per
# launch); (3) popcount tiles against the column kernel over N and over the number of phams.
set -u
OUT=gpurun_out/r05_col; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "every_pocp_af_kernel or sparse64_chunked or full_size_set_metrics or real_collection" > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt
: > $OUT/seg.jsonl; : > $OUT/cross.jsonl
for SEG in 1 2 4 8 16 32; do
  timeout -k 10 300 python3 tools/set_time.py --sizes 2000,3000,5000,20000 --metrics jc --check 3000 --variants=base --env PC_SET_KERNEL=sparsecol --env PC_COL_SEG=$SEG >> $OUT/seg.jsonl 2>> $OUT/err.txt
done
for K in popc sparsecol; do
  timeout -k 10 300 python3 tools/set_time.py --sizes 1000,1500,2000,2500,3000,4000,6000,8000,20000 --metrics jc --check 3000 --variants=base --env PC_SET_KERNEL=$K >> $OUT/cross.jsonl 2>> $OUT/err.txt
  for P in 600 1200 2500; do
    timeout -k 10 300 python3 tools/set_time.py --sizes 2000,5000,10000 --phams $P --metrics jc --check 3000 --variants=base --env PC_SET_KERNEL=$K >> $OUT/cross.jsonl 2>> $OUT/err.txt
  done
done
python3 -c "
import json
for f in ('seg', 'cross'):
    for l in open('$OUT/' + f + '.jsonl'):
        r = json.loads(l); print(f, r.get('env'), r.get('metric'), r.get('n'), r.get('kernel'), r.get('ms_min'), r.get('ms_median'), r.get('oracle_sample_equal'), r.get('failed'))"
