#!/bin/bash
set -u
OUT=gpurun_out/r05_col
mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "every_pocp_af_kernel or sparse64_chunked or full_size_set_metrics or real_collection" > $OUT/tests.txt 2>&1; tail -5 $OUT/tests.txt
bash tools/r05_col2.sh base
