set -u
OUT=gpurun_out/r05_col; mkdir -p $OUT
for SEG in 8 12 16 24; do
PC_COL_SEG=$SEG timeout -k 10 300 python3 tools/set_time.py --sizes 8000,20000 --metrics jc,pocp,af --check 0 --variants=base --env PC_COL_SEG=$SEG 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('seg $SEG', r.get('metric'), r.get('n'), r.get('kernel'), r.get('ms_min'), r.get('ms_median'))"
done
