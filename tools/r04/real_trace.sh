#!/bin/bash
# Kernel trace of the real-collection-shaped peq fill: launches on their streams, then serialised (PC_ALIGN_STREAMS=1)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
N=${1:-5000}
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/real_trace -o t -- python3 tools/real_trace.py -n $N > gpurun_out/real_trace.log 2>&1 &&
python3 tools/real_trace.py --summarise gpurun_out/real_trace > gpurun_out/real_trace_summary.txt 2>&1 &&
PC_ALIGN_STREAMS=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/real_trace_serial -o t -- python3 tools/real_trace.py -n $N > gpurun_out/real_trace_serial.log 2>&1 &&
python3 tools/real_trace.py --summarise gpurun_out/real_trace_serial > gpurun_out/real_trace_serial_summary.txt 2>&1
tail -3 gpurun_out/real_trace.log gpurun_out/real_trace_serial.log
