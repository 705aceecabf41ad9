#!/bin/bash
set -u
OUT=gpurun_out/r04_final3; mkdir -p $OUT
timeout -k 10 1000 python3 -u -m pytest tests -q -m gpu --durations=10 > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee $OUT/ab.txt; tail -16 $OUT/pytest.log | tee -a $OUT/ab.txt
timeout -k 10 400 python3 -u tools/big_fill.py 2>&1 | grep --line-buffered -v amdgpu | tee $OUT/big_fill.txt
