#!/bin/bash
for v in base dm4 dm8 dm24 base; do
  echo "== $v"
  PHAMCLUST_NATIVE_VARIANT=${v#base} python3 -u tools/set_metric_bench.py --sizes 1000,2000,3000 --steps 9 --metrics af --check 0 2>&1 | grep --line-buffered '^{' | python3 -u -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['metric'], r['n_genomes'], round(r['device_ms'], 4), flush=True)"
done
