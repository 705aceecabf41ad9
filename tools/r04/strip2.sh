#!/bin/bash
set -u
OUT=gpurun_out/r04_strip2; mkdir -p $OUT
timeout -k 10 900 python3 -u -m pytest tests/test_gpu_parity.py tests/test_pipeline.py -x -q -m gpu -k "strip_mined or real_collection or sparse64_chunked or falls_back or two_ranks_rehearsal or borrowed or out_of_memory" --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee $OUT/ab.txt; tail -25 $OUT/pytest.log | tee -a $OUT/ab.txt
timeout -k 10 400 python3 -u tools/long_gene_bench.py --variants 0,32,48,64 2>&1 | tee -a $OUT/ab.txt
timeout -k 10 300 python3 -u tools/launch_cost.py -n 5000 --ranks 1,2,4 --out $OUT/launch_cost.txt 2>&1 | tee -a $OUT/ab.txt
