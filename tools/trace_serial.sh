#!/bin/bash
# Per-launch durations of one fill with the alignment launches serialised (PC_ALIGN_STREAMS=1): which launch classes gain
# or lose between two builds.   gpurun -- 'bash tools/trace_serial.sh TAG'   ->  gpurun_out/TAG_serial/t_kernel_trace.csv
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
export PC_ALIGN_STREAMS=1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${TAG}_serial -o t -- python3 bench.py --steps 1 --warmup 1 --cpu-seconds 0 --verify-pairs 0 > gpurun_out/${TAG}_serial.log 2>&1
tail -2 gpurun_out/${TAG}_serial.log
