#!/bin/bash
set -u
OUT=gpurun_out/r04_suite; mkdir -p $OUT
timeout -k 10 1100 python3 -u -m pytest tests -q -m gpu --durations=15 > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee $OUT/ab.txt; tail -40 $OUT/pytest.log | tee -a $OUT/ab.txt
