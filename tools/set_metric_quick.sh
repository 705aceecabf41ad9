python tools/set_metric_bench.py --metrics jc,pocp,af --steps 7 --check 5000 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['metric'], r['n_genomes'], 'device_ms %.4f'%r['device_ms'], 'oracle', r['oracle_sample_equal'])"
