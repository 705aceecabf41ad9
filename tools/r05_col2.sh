#!/bin/bash
set -u
OUT=gpurun_out/r05_col
mkdir -p $OUT
V=${1:-base}
timeout -k 10 500 python3 tools/set_time.py --sizes 2000,5000,20000 --metrics jc,gcs --check 20000 --variants=$V --env PC_SET_KERNEL=sparsecol > $OUT/col2.jsonl 2> $OUT/col2.err
python3 -c "
import sys,json
for l in open('$OUT/col2.jsonl'):
    r=json.loads(l); print(r.get('variant'), r.get('metric'), r.get('n'), r.get('kernel'), r.get('ms_min'), r.get('ms_median'), r.get('oracle_sample_equal'), r.get('failed'))"
tail -3 $OUT/col2.err
