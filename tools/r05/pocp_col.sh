set -u
OUT=gpurun_out/r05_col; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "column_kernel_value or every_pocp_af_kernel or sparse64_chunked or full_size_set_metrics or real_collection" --durations=5 > $OUT/tests.txt 2>&1; tail -12 $OUT/tests.txt
: > $OUT/pocp.jsonl
for K in popc sparse64 sparsecol; do
timeout -k 10 300 python3 tools/set_time.py --sizes 2000,3000,5000,20000 --metrics pocp,af --check 20000 --variants=base --env PC_SET_KERNEL=$K >> $OUT/pocp.jsonl 2>> $OUT/pocp.err
done
timeout -k 10 300 python3 tools/set_time.py --sizes 2000,3000,5000,20000 --metrics jc,pocp,af --check 20000 --variants=base >> $OUT/pocp.jsonl 2>> $OUT/pocp.err
python3 -c "
import json
for l in open('$OUT/pocp.jsonl'):
    r = json.loads(l); print(r.get('env'), r.get('metric'), r.get('n'), r.get('kernel'), r.get('ms_min'), r.get('ms_median'), r.get('oracle_sample_equal'), r.get('failed'))"
