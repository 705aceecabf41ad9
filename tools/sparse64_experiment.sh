#!/bin/bash
# A/B of the pocp / af kernels (PC_SET_KERNEL): default choice against the 64-tile sparse kernel, device ms + oracle check
out=gpurun_out/sparse64_experiment.txt
: > $out
for K in default sparse64 walker; do
  echo "== PC_SET_KERNEL=$K" >> $out
  if [ $K = default ]; then unset PC_SET_KERNEL; else export PC_SET_KERNEL=$K; fi
  timeout -k 10 500 python tools/set_metric_bench.py --metrics pocp,af --sizes ${SIZES:-2000,5000,20000} --steps 7 --check 5000 --out gpurun_out/sp64_$K.json 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['metric'], r['n_genomes'], 'device_ms %.4f'%r['device_ms'], 'oracle', r['oracle_sample_equal'])" >> $out
done
cat $out
