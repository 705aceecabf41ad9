#!/bin/bash
set -u
OUT=gpurun_out/r04_final_x; mkdir -p $OUT
timeout -k 10 200 python3 -u tools/long_gene_bench.py --lens 20000 --pairs 20000 --variants 0 --check 3 2>&1 | grep --line-buffered -v amdgpu | tee $OUT/long_gene_20000.txt
timeout -k 10 200 python3 -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "launch_policy" 2>&1 | tail -3
timeout -k 10 900 python3 -u tools/stress_random.py 20261006 2000 > $OUT/stress2000.txt 2>&1; tail -3 $OUT/stress2000.txt
