#!/bin/bash
# The r05 set-metric sweeps behind profiles/r05/experiments/sparse_col.txt sections 5-7, in one GPU-box call:
#   gpurun --timeout 900 -- 'bash tools/r05/sweeps.sh'
# (a) every kernel family forced, jc / pocp / af, N = 2,000 ... 20,000; (b) source tiles per unit of the column kernel (PC_COL_SEG);
# (c) small matrices: where the column kernel takes over from the popcount tiles (jc, pocp) and from the 64 x 64 sparse tiles (af).
set -u
row() { python3 -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('$1', r.get('metric'), r.get('n'), r.get('kernel'), r.get('ms_min'), r.get('ms_median'), r.get('oracle_sample_equal'), r.get('failed'))"; }
for K in popc sparse64 sparsecol; do timeout -k 10 300 python3 tools/set_time.py --sizes 2000,3000,5000,20000 --metrics jc,pocp,af --check 20000 --variants=base --env PC_SET_KERNEL=$K 2>/dev/null | row "forced $K"; done
timeout -k 10 300 python3 tools/set_time.py --sizes 2000,3000,5000,20000 --metrics jc,pocp,af --check 20000 --variants=base 2>/dev/null | row "default"
for SEG in 8 12 16 24; do timeout -k 10 300 python3 tools/set_time.py --sizes 8000,20000 --metrics jc,pocp,af --check 0 --variants=base --env PC_COL_SEG=$SEG 2>/dev/null | row "seg $SEG"; done
for K in sparsecol popc; do timeout -k 10 200 python3 tools/set_time.py --sizes 1000,1500,1800,2000,2200,2500 --metrics jc,pocp,af --check 2000 --variants=base --env PC_SET_KERNEL=$K 2>/dev/null | row "small $K"; done
for K in sparse sparse64 sparsecol; do timeout -k 10 200 python3 tools/set_time.py --sizes 200,400,600,800,1000,1300,1500,1800 --metrics af --check 2000 --variants=base --env PC_SET_KERNEL=$K 2>/dev/null | row "af $K"; done
