#!/bin/bash
set -u
OUT=gpurun_out/r04_strip3; mkdir -p $OUT
timeout -k 10 400 python3 -u tools/long_gene_bench.py --lens 2000,4000,4500,9000,20000 --variants 0,32,48 --check 4 2>&1 | grep --line-buffered -v amdgpu | tee $OUT/long.txt
timeout -k 10 300 python3 -u tools/real_shape.py -n 5000 --out $OUT/real_shape.json 2>&1 | grep --line-buffered -v amdgpu | tee $OUT/real.txt
timeout -k 10 1000 python3 -u -m pytest tests -x -q -m gpu --durations=12 > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee $OUT/ab.txt; tail -25 $OUT/pytest.log | tee -a $OUT/ab.txt
