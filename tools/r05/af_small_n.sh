for K in sparse sparse64 sparsecol; do timeout -k 10 200 python3 tools/set_time.py --sizes 200,400,600,800,1000,1300 --metrics af --check 2000 --variants=base --env PC_SET_KERNEL=$K 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('$K', r.get('metric'), r.get('n'), r.get('kernel'), r.get('ms_min'), r.get('oracle_sample_equal'))"; done
