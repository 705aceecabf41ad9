#!/bin/bash
# Hardware queues (GPU_MAX_HW_QUEUES, read by the HIP runtime when it starts; default 4) x alignment streams on the
# real-collection-shaped fill (three strip-mined launches hold a queue for ~100 ms each) and on the benchmark's
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
N=${1:-5000}
for q in 4 8 16; do for s in 8 16; do
  echo "== GPU_MAX_HW_QUEUES=$q PC_ALIGN_STREAMS=$s synth_real($N)"
  GPU_MAX_HW_QUEUES=$q PC_ALIGN_STREAMS=$s python3 -u tools/real_trace.py -n $N 2>&1 | tail -1 || exit 1
done; done
for q in 4 8 16; do
  echo "== GPU_MAX_HW_QUEUES=$q bench.py"
  GPU_MAX_HW_QUEUES=$q python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --verify-pairs 0 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.readlines()[-1]); print(r['ms_per_step'], r['stage_ms'])" || exit 1
done
GPU_MAX_HW_QUEUES=8 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/real_trace3 -o t -- python3 tools/real_trace.py -n $N > gpurun_out/real_trace3.log 2>&1 &&
python3 tools/real_trace.py --summarise gpurun_out/real_trace3 > gpurun_out/real_trace3_summary.txt 2>&1
