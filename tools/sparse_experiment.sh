#!/bin/bash
for xf in 0 1 2 3 4 5 12 13; do
  echo "== PC_SP_XF=$xf"
  PC_SP_XF=$xf python3 tools/set_metric_bench.py --sizes 20000 --metrics pocp --steps 5 --check 0 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['metric'], r['n_genomes'], 'device_ms %.4f'%r['device_ms'])"
done
