timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "column_kernel" 2>&1 | tail -5
for K in sparsecol popc; do timeout -k 10 200 python3 tools/set_time.py --sizes 1000,1500,1800,2000,2200,2500 --metrics jc,pocp,af --check 2000 --variants=base --env PC_SET_KERNEL=$K 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('$K', r.get('metric'), r.get('n'), r.get('kernel'), r.get('ms_min'), r.get('oracle_sample_equal'))"; done
timeout -k 10 200 python3 tools/set_time.py --sizes 1000,1500,1800,2000,2200,2500 --metrics af --check 2000 --variants=base --env PC_SET_KERNEL=sparse64 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('sparse64', r.get('metric'), r.get('n'), r.get('kernel'), r.get('ms_min'), r.get('oracle_sample_equal'))"
