#!/bin/bash
# A/B of the popcount kernels on the small set-metric fills (device time of jc, HIP events inside the library):
#   PC_POPC_TILE=64  64x64 tiles, a workgroup's four waves each own 16 rows of the tile
#   PC_POPC_TILE=32  32x32 tiles, the four waves split the bitmap words (k_set_popc_ksplit)
for cfg in "PC_POPC_TILE=64" "PC_POPC_TILE=32"; do
  echo "== $cfg"
  env $cfg python3 tools/set_metric_bench.py --sizes ${1:-1000,2000,3000,5000} --metrics jc --steps 9 --check 2000 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['n_genomes'], 'device_ms %.4f'%r['device_ms'], 'wall_dev %.4f'%r['wall_ms']['pc_fill_dev'], 'wall_cold %.3f'%r['wall_upload_sets_plus_fill_to_pinned_host_ms'], 'oracle', r['oracle_sample_equal'])"
done
