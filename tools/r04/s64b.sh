#!/bin/bash
set -u
OUT=gpurun_out/r04_s64; mkdir -p $OUT
for w in 16 8; do
echo "== PC_S64_WAVES=$w" | tee -a $OUT/ab2.txt
PC_S64_WAVES=$w python3 -u tools/set_metric_bench.py --sizes 5000,8000,20000 --steps 7 --metrics pocp,af 2>&1 | grep --line-buffered '^{' | python3 -u -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['metric'], r['n_genomes'], round(r['device_ms'], 4), r['oracle_sample_equal'], flush=True)" | tee -a $OUT/ab2.txt
done
