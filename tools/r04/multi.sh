#!/bin/bash
set -u
OUT=gpurun_out/r04_multi; mkdir -p $OUT
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py tests/test_pipeline.py -x -q -m gpu -k "multi_context or gpus_in_one_process or falls_back or two_ranks_rehearsal" --durations=5 > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee $OUT/ab.txt; tail -15 $OUT/pytest.log | tee -a $OUT/ab.txt
