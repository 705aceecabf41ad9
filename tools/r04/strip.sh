#!/bin/bash
set -u
OUT=gpurun_out/r04_strip; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "strip_mined or long_and_ragged or percent_positives or bytes_outside" > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee $OUT/ab.txt; tail -15 $OUT/pytest.log | tee -a $OUT/ab.txt
timeout -k 10 500 python3 tools/long_gene_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
