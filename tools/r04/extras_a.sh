#!/bin/bash
# r04 final records beside tools/profile_round.sh: the tools' outputs under gpurun_out/r04_final_x/
set -u
OUT=gpurun_out/r04_final_x; mkdir -p $OUT
t() { timeout -k 10 "$@"; }
t 300 python3 -u tools/bucket_size_bench.py --lens 100,207,420,800 2>&1 | grep --line-buffered -v amdgpu > $OUT/bucket_size_bench.txt; tail -5 $OUT/bucket_size_bench.txt
t 300 python3 -u tools/long_gene_bench.py --lens 2000,4000,4500,5000,9000,20000 --pairs 2048 --variants 0,32,48,-1 --check 4 2>&1 | grep --line-buffered -v amdgpu > $OUT/long_gene_bench.txt; tail -7 $OUT/long_gene_bench.txt
t 400 python3 -u tools/real_shape.py -n 5000 --also 1000,2000 --out $OUT/real_shape.json > $OUT/real_shape.txt 2>&1; tail -3 $OUT/real_shape.txt | cut -c1-400
t 200 python3 -u tools/shard_balance.py 5000 8 peq balanced 2>&1 | grep --line-buffered -v amdgpu > $OUT/shard_rehearsal_8ranks.txt; tail -2 $OUT/shard_rehearsal_8ranks.txt
t 200 python3 -u tools/slice_scaling.py 5000 2>&1 | grep --line-buffered -v amdgpu > $OUT/slice_scaling.txt; tail -2 $OUT/slice_scaling.txt
for n in 2000 10000 20000; do t 300 python3 -u tools/quick_bench.py -n $n --steps 3 --check 3000 2>&1 | grep --line-buffered -E "step 2|oracle" >> $OUT/sizes.txt; done; cat $OUT/sizes.txt | cut -c1-200
