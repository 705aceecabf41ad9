#!/bin/bash
# r04 (second session): where the K4 kernels' empty VALU issue slots are -- instruction cache, second VALU pipe, waits
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
OUT=gpurun_out/r04_pmc_issue; mkdir -p $OUT
Q="--steps 1 --warmup 0 --cpu-seconds 0 --verify-pairs 0"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_VALU --output-format csv -d $OUT/a -o a -- python3 bench.py $Q > $OUT/a.log 2>&1 || echo "pass a failed"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/b -o b -- python3 bench.py $Q > $OUT/b.log 2>&1 || echo "pass b failed"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_ADD_F64 SQ_IFETCH_LEVEL SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/c -o c -- python3 bench.py $Q > $OUT/c.log 2>&1 || echo "pass c failed"
find $OUT -name "*counter_collection.csv" | head
