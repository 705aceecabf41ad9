#!/bin/bash
set -u
OUT=gpurun_out/r04_final_x; mkdir -p $OUT
timeout -k 10 300 python3 -u tools/long_gene_bench.py --lens 2000,4000,4500,5000,9000,20000 --pairs 6000 --variants 0,32,48 --check 4 2>&1 | grep --line-buffered -v amdgpu > $OUT/long_gene_bench.txt; cat $OUT/long_gene_bench.txt
python3 bench.py > $OUT/bench_check.json 2> $OUT/bench_check.err; tail -c 600 $OUT/bench_check.json
