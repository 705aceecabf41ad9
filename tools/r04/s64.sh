#!/bin/bash
set -u
OUT=gpurun_out/r04_s64; mkdir -p $OUT
python3 -u tools/set_metric_bench.py --sizes 8000,20000 --steps 7 --out $OUT/sweep.json 2>&1 | grep --line-buffered '^{' | python3 -u -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['metric'], r['n_genomes'], round(r['device_ms'], 4), r['oracle_sample_equal'], flush=True)" | tee $OUT/ab.txt
PC_SET_KERNEL=sparse64 python3 -u tools/set_metric_bench.py --sizes 5000 --steps 7 --metrics jc,gcs 2>&1 | grep --line-buffered '^{' | python3 -u -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('forced sparse64', r['metric'], r['n_genomes'], round(r['device_ms'], 4), r['oracle_sample_equal'], flush=True)" | tee -a $OUT/ab.txt
