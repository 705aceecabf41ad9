#!/bin/bash
# Kernel timeline of the benchmark's collection at N = 2,000 (where the fixed cost of a fill shows)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl2000b -o t -- python3 tools/real_trace.py -n 2000 --synth 5000 > gpurun_out/tl2000b.log 2>&1 &&
python3 tools/real_trace.py --summarise gpurun_out/tl2000b > gpurun_out/tl2000b_summary.txt 2>&1 &&
PC_ALIGN_STREAMS=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl2000b_serial -o t -- python3 tools/real_trace.py -n 2000 --synth 5000 > gpurun_out/tl2000b_serial.log 2>&1 &&
python3 tools/real_trace.py --summarise gpurun_out/tl2000b_serial > gpurun_out/tl2000b_serial_summary.txt 2>&1
