#!/bin/bash
set -u
bash tools/profile_round.sh r04_final 2>&1 | tail -3
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for k in popc sparse64; do PC_SET_KERNEL=$k python3 -u tools/set_metric_bench.py --sizes 4000,5000,6000,7000 --steps 7 --metrics jc --check 0 2>&1 | grep --line-buffered '^{' | python3 -u -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('$k', r['metric'], r['n_genomes'], round(r['device_ms'], 4), flush=True)"; done | tee gpurun_out/r04_final_x/jc_crossover.txt
