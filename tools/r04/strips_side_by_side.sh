#!/bin/bash
# Strip-mined launches with a slab region and a stream each (default) against lined up on the caller's stream (PC_STRIP_STREAMS=0):
# the real-collection-shaped peq fill, its kernel timeline, and the strip parity tests
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
N=${1:-5000}
echo "== side by side" && python3 -u tools/real_trace.py -n $N 2>&1 | tail -2 &&
echo "== in line (PC_STRIP_STREAMS=0)" && PC_STRIP_STREAMS=0 python3 -u tools/real_trace.py -n $N 2>&1 | tail -2 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/real_trace2 -o t -- python3 tools/real_trace.py -n $N > gpurun_out/real_trace2.log 2>&1 &&
python3 tools/real_trace.py --summarise gpurun_out/real_trace2 > gpurun_out/real_trace2_summary.txt 2>&1 &&
python3 -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "strip or real_collection or long" 2>&1 | tail -3
