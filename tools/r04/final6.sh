#!/bin/bash
# r04, second session: the first session's remaining records re-taken on the final device code (-> gpurun_out/r04_final_z/)
set -u
OUT=gpurun_out/r04_final_z; mkdir -p $OUT
t() { timeout -k 10 "$@"; }
t 400 python3 -u tools/pipeline_time.py 10000 peq > $OUT/pipeline_10000.txt 2>&1; tail -3 $OUT/pipeline_10000.txt | cut -c1-300
t 300 python3 -u tools/launch_cost.py -n 5000 --ranks 1,2,4 --out $OUT/launch_cost.txt 2>&1 | tail -3 | cut -c1-200
bash tools/rehearse_ranks.sh r04_final_z/k 4 2000 > $OUT/rehearse.txt 2>&1; tail -3 $OUT/rehearse.txt | cut -c1-300
t 200 python3 -u tools/long_gene_bench.py --lens 20000 --pairs 20000 --variants 0 --check 3 2>&1 | grep --line-buffered -v amdgpu | tee $OUT/long_gene_20000.txt
t 500 python3 -u tools/big_fill.py 2>&1 | grep --line-buffered -v amdgpu | tee $OUT/big_fill.txt | cut -c1-400
