#!/bin/bash
# Column kernel after a change: parity of the forced kernel families, device times, and the HBM traffic per launch at N = 20,000
# (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled on gfx950).
set -u
OUT=gpurun_out/r05_col; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "column_kernel_value or every_pocp_af_kernel or sparse64_chunked or full_size_set_metrics or real_collection" > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt
timeout -k 10 300 python3 tools/set_time.py --sizes 2000,3000,5000,20000 --metrics jc,pocp,af --check 20000 --variants=base > $OUT/t.jsonl 2> $OUT/t.err
python3 -c "
import json
for l in open('$OUT/t.jsonl'):
    r = json.loads(l); print(r.get('metric'), r.get('n'), r.get('kernel'), r.get('ms_min'), r.get('ms_median'), r.get('oracle_sample_equal'), r.get('failed'))"
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT}"
rm -rf $OUT/pmc_f $OUT/pmc_w
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_f -o f -- python3 tools/set_time.py --sizes 20000 --metrics jc,pocp,af --steps 2 --check 0 > $OUT/pmc_f.log 2>&1 || echo "fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_w -o w -- python3 tools/set_time.py --sizes 20000 --metrics jc,pocp,af --steps 2 --check 0 > $OUT/pmc_w.log 2>&1 || echo "write failed"
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(dict)
for sub, name in (("pmc_f", "FETCH_SIZE"), ("pmc_w", "WRITE_SIZE")):
    for path in glob.glob(f"gpurun_out/r05_col/{sub}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(float); n = collections.defaultdict(set)
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == name and "sparse" in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                agg[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
        for k in agg: tot[k][name] = agg[k] * 1024 / len(n[k])
algo = 20000 * 79 * 8 + 16 * 20000 + 8 * (20000 * 19999 // 2)
for k, v in sorted(tot.items()):
    f, w = 2 * v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
    print(f"{k:24s} fetch(x2) {f / 1e9:6.3f} GB  write {w / 1e9:6.3f} GB  traffic {(f + w) / 1e9:6.3f} GB = {(f + w) / algo:5.2f} x algorithmic ({algo / 1e9:.3f} GB)")
PY
