#!/bin/bash
set -u
OUT=gpurun_out/r05_col
mkdir -p $OUT
timeout -k 10 500 python3 tools/set_time.py --sizes 20000 --metrics jc,af --check 0 --variants=base,y_epi1,y_epi2,y_noprobe,y_noprobe_epi2,y_noload,y_nothing --env PC_SET_KERNEL=sparsecol > $OUT/col3.jsonl 2> $OUT/col3.err
python3 -c "
import sys,json
for l in open('$OUT/col3.jsonl'):
    r=json.loads(l); print(r.get('variant'), r.get('metric'), r.get('n'), r.get('kernel'), r.get('ms_min'), r.get('ms_median'), r.get('failed'))"
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT}"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o p -- python3 tools/set_time.py --sizes 20000 --metrics jc,af --steps 2 --check 0 > $OUT/pmc_sq.log 2>&1 || echo "pmc sq failed"
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc_lds -o p -- python3 tools/set_time.py --sizes 20000 --metrics jc,af --steps 2 --check 0 > $OUT/pmc_lds.log 2>&1 || echo "pmc lds failed"
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INSTS_BRANCH --output-format csv -d $OUT/pmc_x -o p -- python3 tools/set_time.py --sizes 20000 --metrics jc,af --steps 2 --check 0 > $OUT/pmc_x.log 2>&1 || echo "pmc x failed"
python3 - <<'PY'
import csv, glob, collections
for sub in ("pmc_sq", "pmc_lds", "pmc_x"):
    for path in glob.glob(f"gpurun_out/r05_col/{sub}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
        for k in agg:
            if "sparse_col" in k:
                print(sub, k, len(n[k]), {c: round(v / len(n[k])) for c, v in agg[k].items()})
PY
