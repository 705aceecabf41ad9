#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
N=${1:-1000}
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/real_trace_$N -o t -- python3 tools/real_trace.py -n $N > gpurun_out/real_trace_$N.log 2>&1 &&
python3 tools/real_trace.py --summarise gpurun_out/real_trace_$N > gpurun_out/real_trace_${N}_summary.txt 2>&1
