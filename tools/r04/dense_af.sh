#!/bin/bash
# r04 (second session): af's broadcast entries through LDS (pc_s6_dense) -- parity of the set metrics, then device times
set -u
OUT=gpurun_out/r04_dense_af; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "set_metric or pocp or af or sparse or golden_distance or shard" > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc $rc" | tee $OUT/ab.txt; tail -3 $OUT/pytest.log | tee -a $OUT/ab.txt
[ $rc -ne 0 ] && exit $rc
python3 -u tools/set_metric_bench.py --sizes 1000,2000,3000,3900 --steps 7 --metrics af,pocp 2>&1 | grep --line-buffered '^{' | python3 -u -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['metric'], r['n_genomes'], round(r['device_ms'], 4), r.get('bit_exact'), flush=True)" | tee -a $OUT/ab.txt
